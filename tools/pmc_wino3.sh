# Counters of conv_wino3 (option conv_bf16x3) on the gate shape: tools/pmc_wino3.sh <tag>  ->  gpurun_out/<tag>_pmc_wino3{alu,mem}_<shape>_counters.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r06}
OUT=$R/gpurun_out
cd $R
export IPDM_CONV_BF16X3=1
for shape in "8 128 0 512 512 128 3 1 2 1"; do
  name=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU --kernel-trace -d $OUT/pm_a -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/pm_a -name "*.db" | head -1) $OUT/${TAG}_pmc_wino3mem_${name}
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT/pm_b -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/pm_b -name "*.db" | head -1) $OUT/${TAG}_pmc_wino3alu_${name}
  rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pm_c -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/pm_c -name "*.db" | head -1) $OUT/${TAG}_pmc_wino3lds_${name}
  rm -rf $OUT/pm_a $OUT/pm_b $OUT/pm_c
done
grep -h wino3 $OUT/${TAG}_pmc_wino3*_counters.csv | cut -c60-200
