#!/bin/bash
# x1: HBM-side bytes of the bandwidth-bound side kernels and MFMA-busy of the four ViT GEMMs from rocprofv3 PMC passes —
# one counter per pass, only --kernel-trace beside --pmc (MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots").  -> gpurun_out/pmc_side.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmcs_$C
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcs_$C -- python $R/tools/side_kernels.py side > $R/gpurun_out/pmcs_$C.log 2>&1
done
for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do
  rm -rf $R/gpurun_out/pmcs_$C
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcs_$C -- python $R/tools/side_kernels.py gemm 8 > $R/gpurun_out/pmcs_$C.log 2>&1
done
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmcg_$C
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcg_$C -- python $R/tools/side_kernels.py gemm 8 > $R/gpurun_out/pmcg_$C.log 2>&1
done
cd $R
python tools/pmc_side_summary.py > gpurun_out/pmc_side_summary.log 2>&1
cat gpurun_out/pmc_side_summary.log | tail -40
for d in gpurun_out/pmcs_* gpurun_out/pmcg_*; do [ -d $d ] && rm -rf $d; done
