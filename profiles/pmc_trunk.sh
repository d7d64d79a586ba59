set -u
# SQ counters of the one-launch trunk forward and the per-block kernels it replaces (profiles/trunk_fused_probe.py, four tiles per launch)
export T2H_PROBE_DEFAULT_ONLY=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_trunk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -o p -- python3 $R/profiles/trunk_fused_probe.py 4 > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/p2 -o p -- python3 $R/profiles/trunk_fused_probe.py 4 > /dev/null 2> $OUT/p2.err
cd $R
python3 profiles/pmc_table.py $OUT/p1 $OUT/p2 --match trunk_ --last 2 > $OUT/table.txt 2>&1
rm -rf $OUT/p1 $OUT/p2
cat $OUT/table.txt | cut -c1-400
tail -n 3 $OUT/p1.err $OUT/p2.err
