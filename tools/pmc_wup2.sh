# Counters of the F(2x2,2x2) Upsample kernel (conv_wup2) on its two largest layers: HBM traffic (FETCH_SIZE / WRITE_SIZE in separate
# passes, as MI355X_MICROARCH.md prescribes), instruction mix and matrix-pipe time.
#   tools/pmc_wup2.sh <tag>  ->  gpurun_out/<tag>_pmc_wup2{fetch,write,mem,alu}_<shape>_counters.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r05}
OUT=$R/gpurun_out
cd $R
for shape in "8 128 0 228 500 128 3 1 512 0" "8 128 0 256 256 128 3 1 512 0"; do
  name=$(echo $shape | tr ' ' '_')
  for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "grbm:GRBM_GUI_ACTIVE" \
              "mem:SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU" \
              "alu:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    key=${pass%%:*}; ctrs=${pass#*:}
    rocprofv3 --pmc $ctrs --kernel-trace -d $OUT/pm_w -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
    python3 tools/rocpd_summary.py $(find $OUT/pm_w -name "*.db" | head -1) $OUT/${TAG}_pmc_wup2${key}_${name}
    rm -rf $OUT/pm_w
  done
done
rm -f $OUT/${TAG}_pmc_wup2*_by_grid.csv $OUT/${TAG}_pmc_wup2*_kernel_stats.csv
grep -h wup2 $OUT/${TAG}_pmc_wup2*_counters.csv | cut -c1-140
