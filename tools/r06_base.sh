#!/bin/bash
# Round-6 base line on one MI355X box: the GPU tests, the headline bench line, Charades-STA, the exclusive phase times and the rocprofv3
# summaries / layer timeline of the headline command.  Everything lands in gpurun_out/r06_base_*.   (arg 1 = "notests" skips pytest)
ulimit -c 0
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export GRAFT_REPO_ROOT=$R
T=${TAG:-r06_base}
E=$O/${T}_err.log; : > $E
if [ "$1" != "notests" ]; then
timeout 2400 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -E "^E |passed|failed|^FAILED|^tests/.*(Error|assert)" | head -60 | tee $O/${T}_tests.log
cp profiles/r06_parity_errors.json $O/${T}_parity_errors.json 2>/dev/null
fi
python bench.py > $O/${T}_bench_qvh.json 2>> $E
python bench.py --workload charades --steps 30 --warmup 8 --no-cpu-baseline --no-hbm-kernels > $O/${T}_bench_charades.json 2>> $E
for f in qvh charades; do python tools/bench_brief.py < $O/${T}_bench_$f.json; done | tee $O/${T}_brief.txt
python tools/phase_times2.py > $O/${T}_phase_times.txt 2>> $E
python tools/phase_times2.py --no-lookahead >> $O/${T}_phase_times.txt 2>> $E
cat $O/${T}_phase_times.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/${T}_prof_bench.log 2>&1
cd $R
DB=$(find gpurun_out/prof -name "*.db" | head -1)
python tools/prof_summary.py $DB $O/${T}_kernel_stats.txt 0.0 > /dev/null
python tools/prof_summary.py $DB $O/${T}_kernel_stats_by_grid.txt 0.0 grid > /dev/null
python tools/prof_gaps.py $DB 0.5 > $O/${T}_gpu_busy.txt
python tools/prof_layer.py $DB 12 3 > $O/${T}_layer_timeline.txt
grep '"metric"' $O/${T}_prof_bench.log > $O/${T}_bench_under_rocprof.json
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${T}_rocprofv3_kernel_stats.csv
rm -rf gpurun_out/prof
head -30 $O/${T}_kernel_stats_by_grid.txt | cut -c1-180
tail -5 $E
