#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$R}
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "cross_block or attention_bias_mask" 2>&1 | tail -5 | tee $O/r04d_xs_tests.log
timeout 600 python -m pytest tests/test_generate_gpu.py tests/test_model_gpu.py -q -x 2>&1 | tail -4 | tee -a $O/r04d_xs_tests.log
for spec in "xs_on:" "xs_off:MRB_ATTN_XS=0"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "== $label"; env $envs timeout 300 python tools/dec_prof.py 20 2>&1 | tail -1
done | tee $O/r04d_dec_loop.txt
B="timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels"
for rep in 1 2; do for spec in "base(v1,noxs):MRB_DEC_PROJ_V2=0;MRB_ATTN_XS=0" "v2_g0+xs:" "v2_g64+xs:MRB_DEC_GRID=64" "v1+xs:MRB_DEC_PROJ_V2=0" "v2_g64,noxs:MRB_DEC_GRID=64;MRB_ATTN_XS=0"; do
  label=${spec%%:*}; envs=${spec#*:}
  for wl in "" "--workload charades"; do
    line=$(env $(echo $envs | tr ';' ' ') $B --steps 24 --warmup 8 $wl 2>$O/ab_err.log | python tools/bench_brief.py)
    [ -z "$line" ] && line="FAILED: $(tail -2 $O/ab_err.log | tr '\n' ' ' | cut -c1-300)"
    echo "$label | $line" | cut -c1-200
  done
done; done | tee $O/r04d_ab.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/profd
timeout 600 rocprofv3 --kernel-trace -d $O/profd -- python $R/tools/dec_prof.py 6 > $O/profd_run.log 2>&1
cd $R
DB=$(find gpurun_out/profd -name "*.db" | head -1)
python tools/prof_dec_layer.py $DB 12 > $O/r04d_dec_layer_alone.txt
rm -rf gpurun_out/profd
head -50 $O/r04d_dec_layer_alone.txt | cut -c1-170
