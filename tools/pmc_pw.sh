#!/bin/bash
# Counter passes of the pointwise kernel (conv_pw.hip) on two of its layers: matrix-pipe busy cycles and the effective clock
# (GRBM_GUI_ACTIVE / 8 / wall time).  Run through gpurun from the repo root:   tools/pmc_pw.sh <tag>   -> gpurun_out/<tag>_pmc_pw_*
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
OUT=gpurun_out
for shape in "8 128 128 228 500 128 1 1 0 0" "8 256 0 57 125 768 1 1 1 0"; do
  name=${TAG}_pmc_pw_$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM --kernel-trace -d $OUT/${name}_sq -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/${name}_sq -name "*.db" | head -1) $OUT/${name}_sq
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM --kernel-trace -d $OUT/${name}_grbm -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/${name}_grbm -name "*.db" | head -1) $OUT/${name}_grbm
  rm -rf $OUT/${name}_sq $OUT/${name}_grbm
  grep -h conv_pw $OUT/${name}_sq_counters.csv $OUT/${name}_grbm_counters.csv $OUT/${name}_sq_kernel_stats.csv $OUT/${name}_grbm_kernel_stats.csv
done
