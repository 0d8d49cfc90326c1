#!/bin/bash
# Builds of libipdm_hip.so that differ in ONE translation unit's -D flags, for tools/ab_lib.py (interleaved timing in one
# process):   tools/build_variants.sh conv_wino2.hip ko1:-DIPDM_WINO2_KO=1 ko2:-DIPDM_WINO2_KO=2 ...
# -> ipdm-pytorch_amd/libipdm_hip_<tag>.so   (git-ignored; they travel to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/../ipdm-pytorch_amd/csrc"
make -j8 >/dev/null
src=$1; shift
base=${src%.hip}
for spec in "$@"; do
    tag=${spec%%:*}; flags=${spec#*:}
    hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-function ${flags//,/ } -c $src -o /tmp/${base}_${tag}.o
    objs=$(ls *.o | grep -v -E "^(${base}|conv_nm|conv_sx|attn_sx)\.o$")      # (the product library's objects: the opt-in kernels are a second .so)
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../libipdm_hip_${tag}.so $objs /tmp/${base}_${tag}.o -ldl
    echo "built ../libipdm_hip_${tag}.so ($flags)"
done
