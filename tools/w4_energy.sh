#!/bin/bash
# Energy ledger of the ViT fc1 launch -> gpurun_out/w4_energy.txt.  Ablation builds: tools/exp_build.sh <tag> gemm.hip -DEXP_...
cd "$(dirname "$0")/.."
O=gpurun_out/w4_energy.txt; : > $O
run() { tag=$1; shift; line=$(bash tools/pwr_probe.sh $tag "$@" 2>/dev/null | tr '\n' ' '); echo "$tag | $line" | tee -a $O; }
run product        python tools/w4_energy.py 1
run bias_only      python tools/w4_energy.py 0
run no_epilogue    env MRBLIP_LIB=exp_libs/lib_w4noepi.so python tools/w4_energy.py 1
run no_lds_dma     env MRBLIP_LIB=exp_libs/lib_w4nodma.so python tools/w4_energy.py 1
run mfma_only_1w   exp_libs/mfma_power 33 1 7
run idle           sleep 7
