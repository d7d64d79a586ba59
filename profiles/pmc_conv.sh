set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -o p -- python3 $R/profiles/layer_probe.py --only conv --reps 2 > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/p2 -o p -- python3 $R/profiles/layer_probe.py --only conv --reps 2 > /dev/null 2> $OUT/p2.err
cd $R
python3 profiles/pmc_table.py $OUT/p1 $OUT/p2 --match bx3_ --last 2 > $OUT/table.txt 2>&1
rm -rf $OUT/p1 $OUT/p2
cat $OUT/table.txt | cut -c1-400
tail -3 $OUT/p1.err $OUT/p2.err
