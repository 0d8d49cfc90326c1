#!/bin/bash
# rocprofv3 kernel-trace summaries of one bench step in the default (exact-f32) mode and in the opt-in split-bf16 mode.
#   tools/profile_split.sh <tag>   -> gpurun_out/<tag>_{f32,x6}_*   (summarise with tools/rocpd_summary.py)
TAG=${1:-r01j}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_f32_trace -o bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt > $OUT/${TAG}_f32_bench.json 2> $OUT/${TAG}_f32.err
IPDM_CONV_SPLIT=3 IPDM_ATTN_SPLIT=3 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_x6_trace -o bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt > $OUT/${TAG}_x6_bench.json 2> $OUT/${TAG}_x6.err
for m in f32 x6; do
  db=$(find $OUT/${TAG}_${m}_trace -name "*.db" | head -1)
  python3 tools/rocpd_summary.py $db $OUT/${TAG}_${m}_bench_b8 > /dev/null 2>&1
done
ls $OUT | grep ${TAG}
