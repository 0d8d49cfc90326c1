# kernel trace of whole UNet forwards at B = 1 (and B = 8): durations and the gaps between launches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r05a}
OUT=$R/gpurun_out
cd $R
for B in 1 8; do
  rocprofv3 --kernel-trace -d $OUT/${TAG}_gaps_b$B -o c -- python3 tools/time_forward.py $B 3 > $OUT/${TAG}_gaps_b$B.log 2>&1
  python3 tools/gap_summary.py $(find $OUT/${TAG}_gaps_b$B -name "*.db" | head -1) $OUT/${TAG}_gaps_b$B.csv
  rm -rf $OUT/${TAG}_gaps_b$B
done
head -30 $OUT/${TAG}_gaps_b1.csv | cut -c1-170
