#!/bin/bash
# Matrix-pipe counters of the opt-in split-bf16 kernels (own --pmc passes, kernel-trace only).
#   tools/profile_split_pmc.sh <tag>  -> gpurun_out/<tag>_pmc_*  (summarise: tools/rocpd_summary.py)
TAG=${1:-r01l}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export IPDM_CONV_SPLIT=3 IPDM_ATTN_SPLIT=3
CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS"
rocprofv3 --pmc $CNT --kernel-trace -d $OUT/${TAG}_pmc_sq_convsx -o c -- python3 tools/one_conv.py 8 128 0 512 512 128 3 1 2 1 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/${TAG}_pmc_grbm_convsx -o c -- python3 tools/one_conv.py 8 128 0 512 512 128 3 1 2 1 > /dev/null 2>&1
rocprofv3 --pmc $CNT --kernel-trace -d $OUT/${TAG}_pmc_sq_attnsx -o c -- python3 tools/one_attn.py 8 4 4096 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/${TAG}_pmc_grbm_attnsx -o c -- python3 tools/one_attn.py 8 4 4096 > /dev/null 2>&1
for k in sq_convsx grbm_convsx sq_attnsx grbm_attnsx; do
  db=$(find $OUT/${TAG}_pmc_$k -name "*.db" | head -1)
  python3 tools/rocpd_summary.py $db $OUT/${TAG}_pmc_$k > /dev/null 2>&1
done
cat $OUT/${TAG}_pmc_*_counters.csv
