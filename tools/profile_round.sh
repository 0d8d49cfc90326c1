#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*  (copy the summaries you want judged into profiles/)
# Counters are collected in their own passes (--pmc with --kernel-trace only), never with --stats.
TAG=${1:-r01}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err
for shape in "8 128 0 512 512 128 3 1 2 1" "8 64 0 512 512 64 3 1 2 1"; do
  name=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_pmc_fetch_$name -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_pmc_write_$name -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace -d $OUT/${TAG}_pmc_sq_$name -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/${TAG}_pmc_grbm_$name -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
done
find $OUT -name "*.csv" -path "*${TAG}*" | head -40
# whole-step HBM traffic of every kernel (per-launch averages): two separate counter passes over one bench step
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_pmc_step_fetch -o s -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_pmc_step_write -o s -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
