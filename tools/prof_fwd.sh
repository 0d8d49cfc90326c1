cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_fwd
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace -d $OUT/t -o c -- python3 tools/time_forward.py 8 2 > $OUT/log.txt 2>&1
python3 tools/rocpd_summary.py $(find $OUT/t -name "*.db" | head -1) $OUT/fwd
rm -rf $OUT/t
head -12 $OUT/fwd_kernel_stats.csv | cut -c1-150
