# Memory-pipeline counters of the dominant kernel (conv_wino2) on its gate shapes: how many vector-memory / LDS / scalar instructions
# a launch issues and how long the memory instructions keep their pipe (NOTEBOOK.md round 5: a vector-memory instruction costs its
# slot whatever it fetches).    tools/pmc_wino2_mem.sh <tag>  ->  gpurun_out/<tag>_pmc_wino2mem_<shape>_counters.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r05}
OUT=$R/gpurun_out
cd $R
for shape in "8 128 0 512 512 128 3 1 2 1" "8 128 0 512 512 128 3 1 2 0" "8 256 0 128 128 256 3 1 2 1"; do
  name=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU --kernel-trace -d $OUT/pm_a -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/pm_a -name "*.db" | head -1) $OUT/${TAG}_pmc_wino2mem_${name}
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT/pm_b -o c -- python3 tools/one_conv.py $shape > /dev/null 2>&1
  python3 tools/rocpd_summary.py $(find $OUT/pm_b -name "*.db" | head -1) $OUT/${TAG}_pmc_wino2alu_${name}
  rm -rf $OUT/pm_a $OUT/pm_b
done
rm -f $OUT/${TAG}_pmc_wino2*_by_grid.csv $OUT/${TAG}_pmc_wino2*_kernel_stats.csv
grep -h wino2 $OUT/${TAG}_pmc_wino2mem_*_counters.csv $OUT/${TAG}_pmc_wino2alu_*_counters.csv | cut -c1-160
