#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*  (copy the summaries you want judged into profiles/)
# Counters are collected in their own passes (--pmc with --kernel-trace only), never with --stats.
# (the traced bench runs ONE warm-up and three timed steps: with two or more warm-up steps bench.py samples the shader clock under the last of
#  them with one extra wave on a stream of its own -- that wave keeps one SIMD from taking two of conv_wino2's 256-register waves, so the kernel's
#  256 workgroups run on 255 CUs for that untimed step (x1.4 per launch) and the trace's average over ALL steps reads 0.465 ms where the timed
#  steps' events read 0.428: profiles/r06p_*.  The kernel statistics average over five steps: 1 + 3 + the untimed profiling step.)
TAG=${1:-r02}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-extra-legs > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err
python3 tools/rocpd_summary.py $(find $OUT/${TAG}_trace -name "*.db" | head -1) $OUT/${TAG}_bench_b8
# HBM traffic of every kernel (per-launch averages): two separate counter passes.  A counter pass serialises every
# dispatch (a full step of 23k launches takes > 25 minutes), so the passes run a step with the same 3 : 2 mix of proj and
# img UNet forwards as the headline (45 : 30) but one fifteenth of them: t_start_proj=[3], t_start_img=[2], no ultra
if [ -z "$SKIP_TRAFFIC" ]; then
RED="--steps 1 --warmup 0 --t_start_proj 3 --t_start_img 2 --no-ultra --no-cpu-baseline --no-roofline --no-alt --no-extra-legs"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_pmc_step_fetch -o s -- python3 bench.py $RED > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_pmc_step_write -o s -- python3 bench.py $RED > /dev/null 2>&1
python3 tools/traffic_summary.py $(find $OUT/${TAG}_pmc_step_fetch -name "*.db" | head -1) $(find $OUT/${TAG}_pmc_step_write -name "*.db" | head -1) ${TAG}
cp profiles/${TAG}_traffic.json profiles/${TAG}_hbm_by_kernel.csv $OUT/
fi
# matrix-pipe utilisation of two common shapes of the dominant kernel (conv_wino2: 128->128 @512x512, 256->256 @128x128)
# ... and of the F(2x2,2x2) Upsample kernel (conv_wup2: 128 channels, source 228x500; act 512 = the micro-benchmark's Upsample mode)
[ -n "$SKIP_PMC" ] && exit 0
for shape in "8 128 0 512 512 128 3 1 2 1" "8 256 0 128 128 256 3 1 2 1" "8 128 0 228 500 128 3 1 512 0"; do
  name=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace -d $OUT/${TAG}_pmc_sq_$name -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/${TAG}_pmc_sq_$name -name "*.db" | head -1) $OUT/${TAG}_pmc_sq_$name
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/${TAG}_pmc_grbm_$name -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/${TAG}_pmc_grbm_$name -name "*.db" | head -1) $OUT/${TAG}_pmc_grbm_$name
done
ls $OUT | grep ${TAG}_ | head -40
# the databases are large (the merge back is capped at 64 MiB): keep the summaries only
rm -rf $OUT/${TAG}_trace 2>/dev/null
find $OUT -maxdepth 1 -type d -name "${TAG}_pmc_*" -exec rm -rf {} + 2>/dev/null
ls -la $OUT | grep ${TAG}_ | head -40
