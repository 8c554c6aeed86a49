#!/bin/bash
# Round-3 artefacts on one MI355X box: bench lines of every workload, rocprofv3 summaries of the headline command, PMC traffic of the
# dominant kernel.  Everything lands in gpurun_out/r03_final_*; the builder copies it to profiles/.
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out
python bench.py --steps 20 --warmup 3 > $O/r03_final_bench_qvh.json 2> $O/r03_final_bench_qvh.err
python bench.py --workload charades --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/r03_final_bench_charades.json 2>> $O/r03_final_bench_qvh.err
python bench.py --workload anet --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/r03_final_bench_anet.json 2>> $O/r03_final_bench_qvh.err
python bench.py --batch-per-gpu 4 --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-kernels > $O/r03_final_bench_qvh_b4.json 2>> $O/r03_final_bench_qvh.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels --no-lookahead > $O/r03_final_bench_qvh_nolookahead.json 2>> $O/r03_final_bench_qvh.err
python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-hbm-kernels --vary-text > $O/r03_final_bench_qvh_varytext.json 2>> $O/r03_final_bench_qvh.err
python tools/phase_times2.py > $O/r03_final_phase_times.txt 2>> $O/r03_final_bench_qvh.err
ATTN_ONLY=t5enc,t5enc_masked,vit python tools/attn_bench.py > $O/r03_final_attention.txt 2>> $O/r03_final_bench_qvh.err
python tools/attn_vit_bench.py 2>/dev/null | grep "S=" >> $O/r03_final_attention.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/r03_final_prof_bench.log 2>&1
cd $R
DB=$(find gpurun_out/prof -name "*.db" | head -1)
python tools/prof_summary.py $DB $O/r03_final_kernel_stats.txt 0.0 > /dev/null
python tools/prof_summary.py $DB $O/r03_final_kernel_stats_by_grid.txt 0.0 grid > /dev/null
python tools/prof_gaps.py $DB 0.5 > $O/r03_final_gpu_busy.txt
python tools/prof_streams.py $DB > $O/r03_final_streams.txt 2>&1
grep '"metric"' $O/r03_final_prof_bench.log > $O/r03_final_bench_under_rocprof.json
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r03_final_rocprofv3_kernel_stats.csv
rm -rf gpurun_out/prof
bash tools/pmc_fc1.sh > $O/r03_final_pmc_fc1.log 2>&1
cp $O/pmc_fc1.json $O/r03_final_pmc_fc1.json
for f in qvh charades anet qvh_b4 qvh_nolookahead qvh_varytext; do cut -c1-420 $O/r03_final_bench_$f.json; done
head -12 $O/r03_final_kernel_stats_by_grid.txt | cut -c1-180
tail -5 $O/r03_final_pmc_fc1.log | cut -c1-600
