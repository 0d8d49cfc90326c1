# Counters of the narrow direct convolutions (conv_direct.hip), one shape per kernel instantiation that matters in a step
# (VERDICT r04 item 4b):   tools/pmc_direct.sh <tag>   ->  gpurun_out/<tag>_pmc_direct_<shape>_{a,b,c}_counters.csv
#   8 -> 8 @2000x912 + residual      conv_direct_kernel<8,3,4,false,1,0>   the bandwidth-bound family's heaviest layer
#   128+16 -> 16 @1000x456, x1 parity-planar   conv_direct_kernel<16,3,8,true,1,0>   the VALU-bound reader (act code 258 = GN+SiLU | planar x1)
#   16 -> 16 @1000x456 + residual    conv_direct_kernel<16,3,8,false,1,0>
# Counter passes are separate runs with --kernel-trace only (no --stats, no other trace domain).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r05}
OUT=$R/gpurun_out
cd $R
for shape in "8 8 0 2000 912 8 3 1 2 1" "8 128 16 1000 456 16 3 1 258 0" "8 16 0 1000 456 16 3 1 2 1"; do
  name=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace -d $OUT/pd_a -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/pd_a -name "*.db" | head -1) $OUT/${TAG}_pmc_direct_${name}_a
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace -d $OUT/pd_b -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/pd_b -name "*.db" | head -1) $OUT/${TAG}_pmc_direct_${name}_b
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace -d $OUT/pd_c -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/pd_c -name "*.db" | head -1) $OUT/${TAG}_pmc_direct_${name}_c
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pd_d -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/pd_d -name "*.db" | head -1) $OUT/${TAG}_pmc_direct_${name}_fetch
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pd_e -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/pd_e -name "*.db" | head -1) $OUT/${TAG}_pmc_direct_${name}_write
  rm -rf $OUT/pd_a $OUT/pd_b $OUT/pd_c $OUT/pd_d $OUT/pd_e
done
rm -f $OUT/${TAG}_pmc_direct_*_by_grid.csv
ls $OUT | grep ${TAG}_pmc_direct | head -40
