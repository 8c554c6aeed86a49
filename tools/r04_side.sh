#!/bin/bash
# where does the decoder window go: phase times with the side-stream work toggled (look-ahead on)
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
for spec in "default:" "dec_fwd_side_off:MRB_DEC_FWD_SIDE=0" "dec_bwd_side_off:MRB_DEC_SIDE=0" "both_off:MRB_DEC_FWD_SIDE=0;MRB_DEC_SIDE=0" "all_side_off:MRB_GRAD_SIDE=0" "v2_g64:MRB_DEC_GRID=64" "v1:MRB_DEC_PROJ_V2=0"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "== $label"
  env $(echo $envs | tr ';' ' ') timeout 300 python tools/phase_times2.py 2>/dev/null | grep -E "decoder|encoder|sum of"
done | tee $O/r04c_side_toggles.txt
