#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
OUT=gpurun_out; TAG=${1:-r04d}
RED="--steps 1 --warmup 0 --t_start_proj 3 --t_start_img 2 --no-ultra --no-cpu-baseline --no-roofline --no-alt --no-extra-legs"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT/${TAG}_pmc_step_clock -o s -- python3 bench.py $RED > /dev/null 2>&1
python3 tools/clock_summary.py $(find $OUT/${TAG}_pmc_step_clock -name "*.db" | head -1) $OUT/${TAG}_clock_by_kernel.csv
rm -rf $OUT/${TAG}_pmc_step_clock
