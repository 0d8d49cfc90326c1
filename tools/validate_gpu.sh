#!/bin/bash
# The whole GPU validation of a round in one gpurun call (from the repo root):
#   gpurun --timeout 1800 -- tools/validate_gpu.sh
# 1. the parity suite in the default (exact-f32) mode; 2. the same suite with the opt-in split-bf16 kernels forced on
# (same tolerances); 3. bench.py (headline + alt_modes + cpu_baseline) -> gpurun_out/bench_validate.json
set -o pipefail
mkdir -p gpurun_out
echo "== default mode"; timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3
echo "== IPDM_CONV_SPLIT=3 IPDM_ATTN_SPLIT=3"; IPDM_CONV_SPLIT=3 IPDM_ATTN_SPLIT=3 timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3
echo "== bench"; python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_validate.json | cut -c1-400
