#!/bin/bash
# round-4 baseline on one box: GPU tests, headline bench, phase times with / without look-ahead, in-step layer timelines, decoder alone
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$R}
[ "$1" = "notests" ] || timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $O/r04a_tests.log
timeout 600 python bench.py > $O/r04a_bench_qvh.json 2> $O/r04a_bench_qvh.err; cut -c1-500 $O/r04a_bench_qvh.json
timeout 300 python tools/phase_times2.py > $O/r04a_phase_times.txt 2>> $O/r04a_bench_qvh.err
timeout 300 python tools/phase_times2.py --no-lookahead >> $O/r04a_phase_times.txt 2>> $O/r04a_bench_qvh.err
cat $O/r04a_phase_times.txt
bash tools/prof_layer.sh; cp $O/prof_layer.txt $O/r04a_prof_layer.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/profd
timeout 600 rocprofv3 --kernel-trace -d $O/profd -- python $R/tools/dec_prof.py 6 > $O/profd_run.log 2>&1
tail -2 $O/profd_run.log
cd $R
DB=$(find gpurun_out/profd -name "*.db" | head -1)
python tools/prof_dec_layer.py $DB 12 > $O/r04a_dec_layer_alone.txt
rm -rf gpurun_out/profd
head -70 $O/r04a_dec_layer_alone.txt
