#!/bin/bash
# Build an experiment variant of one source into exp_libs/lib_<tag>.so:  tools/exp_build.sh <tag> <src.hip> [-DFLAG ...]
# (other objects are reused from the in-tree build).  Select it at run time with MRBLIP_LIB=exp_libs/lib_<tag>.so
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
mkdir -p exp_libs
C=mr-blip_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast "$@" -c $C/$src -o exp_libs/${tag}_${src%.hip}.o
objs=""
for s in errors gemm norm attention elementwise lora decproj; do
  if [ "$s.hip" == "$src" ]; then objs="$objs exp_libs/${tag}_${s}.o"; else objs="$objs $C/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o exp_libs/lib_${tag}.so
echo exp_libs/lib_${tag}.so
