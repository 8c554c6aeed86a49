#!/bin/bash
# Round-5 artefacts on one MI355X box: bench lines of every workload, rocprofv3 summaries of the headline command, decoder timelines,
# PMC traffic of the dominant kernel, the full GPU test run (tests/ -m gpu) first.  Everything lands in gpurun_out/r06_final_*; the builder copies it to profiles/.
ulimit -c 0
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export GRAFT_REPO_ROOT=$R
E=$O/r06_final_err.log; : > $E
timeout 2400 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -E "^E |passed|failed|^FAILED|^tests/.*(Error|assert)" | head -60 | tee $O/r06_final_tests.log
python bench.py > $O/r06_final_bench_qvh.json 2>> $E
python bench.py --workload charades --steps 30 --warmup 8 --no-cpu-baseline --no-hbm-kernels > $O/r06_final_bench_charades.json 2>> $E
python bench.py --workload anet --steps 12 --warmup 4 --no-cpu-baseline --no-hbm-kernels > $O/r06_final_bench_anet.json 2>> $E
python bench.py --batch-per-gpu 4 --steps 8 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/r06_final_bench_qvh_b4.json 2>> $E
python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-hbm-kernels --no-lookahead > $O/r06_final_bench_qvh_nolookahead.json 2>> $E
python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-hbm-kernels --vary-text > $O/r06_final_bench_qvh_varytext.json 2>> $E
python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-hbm-kernels --vary-video > $O/r06_final_bench_qvh_varyvideo.json 2>> $E
for f in qvh charades anet qvh_b4 qvh_nolookahead qvh_varytext qvh_varyvideo; do python tools/bench_brief.py < $O/r06_final_bench_$f.json; done | tee $O/r06_final_brief.txt
python tools/phase_times2.py > $O/r06_final_phase_times.txt 2>> $E
python tools/phase_times2.py --no-lookahead >> $O/r06_final_phase_times.txt 2>> $E
python tools/phase_times2.py --workload=charades > $O/r06_final_phase_times_charades.txt 2>> $E
cat $O/r06_final_phase_times.txt
python tools/dec_proj_bench.py 2>>$E | tee $O/r06_final_dec_proj_bench.txt
python tools/dec_cross_attn_bench.py 2>>$E | tee $O/r06_final_dec_cross_attn.txt
ATTN_ONLY=t5enc,t5enc_masked,t5enc_nodrop,vit,qf_cross,dec_cross python tools/attn_bench.py > $O/r06_final_attention.txt 2>> $E
python tools/determinism_check.py 60 6 2>>$E | tail -4 | tee $O/r06_final_determinism.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/r06_final_prof_bench.log 2>&1
cd $R
DB=$(find gpurun_out/prof -name "*.db" | head -1)
python tools/prof_summary.py $DB $O/r06_final_kernel_stats.txt 0.0 > /dev/null
python tools/prof_summary.py $DB $O/r06_final_kernel_stats_by_grid.txt 0.0 grid > /dev/null
python tools/prof_gaps.py $DB 0.5 > $O/r06_final_gpu_busy.txt
python tools/prof_streams.py $DB > $O/r06_final_streams.txt 2>&1
python tools/prof_layer.py $DB 12 3 > $O/r06_final_layer_timeline.txt
grep '"metric"' $O/r06_final_prof_bench.log > $O/r06_final_bench_under_rocprof.json
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r06_final_rocprofv3_kernel_stats.csv
rm -rf gpurun_out/prof
cd /tmp
rm -rf $O/profd
timeout 600 rocprofv3 --kernel-trace -d $O/profd -- python $R/tools/dec_prof.py 6 > $O/profd_run.log 2>&1
cd $R
DB=$(find gpurun_out/profd -name "*.db" | head -1)
python tools/prof_dec_layer.py $DB 12 > $O/r06_final_dec_layer_alone.txt
rm -rf gpurun_out/profd
bash tools/pmc_fc1.sh > $O/r06_final_pmc_fc1.log 2>&1
cp $O/pmc_fc1.json $O/r06_final_pmc_fc1.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
head -14 $O/r06_final_kernel_stats_by_grid.txt | cut -c1-180
tail -3 $O/r06_final_pmc_fc1.log | cut -c1-500
