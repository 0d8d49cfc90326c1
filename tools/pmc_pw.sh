#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
OUT=gpurun_out
for shape in "8 128 128 228 500 128 1 1 0 0" "8 256 0 57 125 768 1 1 1 0"; do
  name=pw_$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM --kernel-trace -d $OUT/$name -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/$name -name "*.db" | head -1) $OUT/$name
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC --kernel-trace -d $OUT/${name}_g -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/${name}_g -name "*.db" | head -1) $OUT/${name}_g
  rm -rf $OUT/$name $OUT/${name}_g
  grep -h conv_pw $OUT/${name}_counters.csv $OUT/${name}_g_counters.csv $OUT/${name}_kernel_stats.csv $OUT/${name}_g_kernel_stats.csv
done
