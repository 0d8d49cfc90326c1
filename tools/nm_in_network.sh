#!/bin/bash
# The conv_nm paradox: the 16-cout MFMA kernel wins 1.15-1.25x in the micro-benchmark and nothing in the
# network.  Kernel traces of two proj+img forwards with the narrow layers on conv_direct (default) and on conv_nm (option
# conv_nm = 1), summarised per (kernel, grid): which launches move over, and what each costs IN the network.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd $R
for nm in 0 1; do
  IPDM_CONV_NM=$nm rocprofv3 --kernel-trace -d $OUT/nm_t$nm -o c -- python3 tools/time_forward.py 8 2 > $OUT/r04_nm_in_network_$nm.log 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/nm_t$nm -name "*.db" | head -1) $OUT/r04_nm_in_network_$nm
  rm -rf $OUT/nm_t$nm
  echo "== conv_nm=$nm"; cat $OUT/r04_nm_in_network_$nm.log | grep -v amdgpu.ids
  grep -E "conv_direct|conv3x3_direct|conv_nm" $OUT/r04_nm_in_network_${nm}_by_grid.csv | cut -c1-200
done
