cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcd
mkdir -p $OUT
i=0
for shape in "8 8 0 2000 912 8 3 1 2 1" "8 16 0 1000 456 16 3 1 2 1" "8 16 0 2000 912 16 3 1 0 0"; do
  i=$((i+1))
  cd $R
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace -d $OUT/a$i -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/a$i -name "*.db" | head -1) $OUT/a$i
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace -d $OUT/b$i -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/b$i -name "*.db" | head -1) $OUT/b$i
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAIT_INST_ANY --kernel-trace -d $OUT/c$i -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/c$i -name "*.db" | head -1) $OUT/c$i
  true
done
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + 2>/dev/null
true
