#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
rm -f $O/r02_phases.log
for args in "" "--no-lookahead"; do for ks in 0 1; do
  MRB_KSPLIT=$ks timeout 300 python tools/phase_times2.py $args 2>&1 | grep -v amdgpu >> $O/r02_phases.log
done; done
cat $O/r02_phases.log
