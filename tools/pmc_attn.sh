#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
OUT=gpurun_out
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT/attn_clock -o c -- python3 tools/one_attn.py 8 4 7125 > /dev/null 2>&1
python3 tools/clock_summary.py $(find $OUT/attn_clock -name "*.db" | head -1) $OUT/r04d_pmc_attn_8_4_7125_clock.csv
rm -rf $OUT/attn_clock
