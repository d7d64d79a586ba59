#!/bin/bash
# rocprofv3 kernel trace of the DEFAULT step (tile pipeline + side streams on) and its stream / overlap analysis:
#   bash profiles/timeline_probe.sh <tag>     (on the GPU box, from the repo root)
set -u
TAG=${1:-rXX}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/timeline -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 8 --skip-cpu-baseline --profile-steps 0 --sustain-s 0 --exact-split-steps 0 > $OUT/timeline_bench.json 2> $OUT/timeline.err
cd $GRAFT_REPO_ROOT
python3 profiles/timeline_analyze.py $(ls $OUT/timeline/*kernel_trace.csv $OUT/timeline/*/*kernel_trace.csv 2>/dev/null | head -1) > $OUT/timeline_summary.txt 2>&1
rm -f $OUT/timeline/*kernel_trace.csv $OUT/timeline/*/*kernel_trace.csv
cat $OUT/timeline_summary.txt
