#!/bin/bash
# One gpurun call that produces every file profiles/README.md lists for a round:  bash profiles/run_profiles.sh r02e
# (run from the repo root on the GPU box; outputs under gpurun_out/<tag>/, copied into profiles/ afterwards)
set -u
TAG=${1:-rXX}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --steps 20 --warmup 5 --kernel-table $OUT/bench_kernels.json > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --skip-cpu-baseline --profile-steps 0 --sustain-s 0 > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p --output-format csv -- python3 $R/bench.py --steps 6 --warmup 3 --skip-cpu-baseline --profile-steps 0 --sustain-s 0 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o p --output-format csv -- python3 $R/bench.py --steps 6 --warmup 3 --skip-cpu-baseline --profile-steps 0 --sustain-s 0 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/probe_fetch -o p --output-format csv -- python3 $R/profiles/pmc_probe.py > /dev/null 2> $OUT/probe_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/probe_write -o p --output-format csv -- python3 $R/profiles/pmc_probe.py > /dev/null 2> $OUT/probe_write.err
rocprofv3 --kernel-trace --stats -d $OUT/trace_infer -o t --output-format csv -- python3 $R/bench.py --mode infer --steps 12 --warmup 4 > $OUT/infer_bench.json 2> $OUT/trace_infer.err
# secondary lines (same box): tiles from the device tile producer; BASELINE configs[2] (cloud+image, bf16-operand MFMA) and its
# fp32 sibling; the other tile sizes of SURVEY 8d
python3 $R/bench.py --steps 20 --warmup 5 --from-producer --skip-cpu-baseline --kernel-table $OUT/bench_producer_kernels.json > $OUT/bench_producer.json 2> $OUT/bench_producer.err
python3 $R/bench.py --steps 20 --warmup 5 --use-image --mlp-precision bf16 --skip-cpu-baseline --profile-steps 0 > $OUT/bench_image_bf16.json 2> $OUT/bench_image_bf16.err
python3 $R/bench.py --steps 20 --warmup 5 --use-image --skip-cpu-baseline --profile-steps 0 > $OUT/bench_image_fp32.json 2> $OUT/bench_image_fp32.err
python3 $R/bench.py --steps 20 --warmup 5 --points 65536 --skip-cpu-baseline --profile-steps 0 --sustain-s 2 > $OUT/bench_n65536.json 2> /dev/null
python3 $R/bench.py --steps 20 --warmup 5 --points 262144 --skip-cpu-baseline --profile-steps 0 --sustain-s 2 > $OUT/bench_n262144.json 2> /dev/null
cd $R
python3 profiles/collect_pmc.py --bench $OUT/pmc_fetch $OUT/pmc_write --probe $OUT/probe_fetch $OUT/probe_write --tag $TAG > $OUT/pmc_summary.txt 2>&1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python3 profiles/summarize_trace.py $(ls $OUT/trace/*kernel_trace.csv $OUT/trace/*/*kernel_trace.csv 2>/dev/null | head -1) --steps 6 --top 70 > $OUT/kernel_trace_steady_state.txt 2>&1
python3 profiles/summarize_trace.py $(ls $OUT/trace_infer/*kernel_trace.csv $OUT/trace_infer/*/*kernel_trace.csv 2>/dev/null | head -1) --steps 4 --top 40 > $OUT/infer_kernel_trace_steady_state.txt 2>&1
cp $(ls $OUT/trace/*kernel_stats.csv $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/rocprofv3_kernel_stats.csv
cp $(ls $OUT/trace_infer/*kernel_stats.csv $OUT/trace_infer/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/infer_rocprofv3_kernel_stats.csv
# the raw traces are large: keep only what the collectors need
rm -f $OUT/trace/*kernel_trace.csv $OUT/trace/*/*kernel_trace.csv $OUT/trace_infer/*kernel_trace.csv $OUT/trace_infer/*/*kernel_trace.csv
rm -f $OUT/pmc_*/*kernel_trace.csv $OUT/pmc_*/*/*kernel_trace.csv $OUT/probe_*/*kernel_trace.csv $OUT/probe_*/*/*kernel_trace.csv
du -sh $OUT; tail -c 600 $OUT/bench.json; head -12 $OUT/kernel_trace_steady_state.txt; head -8 $OUT/infer_kernel_trace_steady_state.txt
