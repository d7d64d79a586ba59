#!/bin/bash
# One gpurun call that produces every file profiles/README.md lists for a round:  bash profiles/run_profiles.sh r04a
# (run from the repo root on the GPU box; outputs under gpurun_out/<tag>/, copied into profiles/ afterwards)
# Order (r04): trace and PMC passes first, then the collectors refresh profiles/rocprof_kernels.json and profiles/pmc_traffic.json IN
# THE BOX'S COPY of the repo, then the driver-style bench.py leg -- so that the committed bench line carries THIS tag's profiler
# durations (roofline.avg_us_rocprof) and PMC traffic, not the previous profile's.
set -u
TAG=${1:-rXX}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# the profiler passes time kernels ALONE (side streams and the tile pipeline off), like the per-kernel leg inside bench.py whose durations they must agree
# with; the bench legs further down run the product's default (weight gradients on side streams)
export T2H_OVERLAP_WGRAD=0 T2H_OVERLAP_CONV_WGRAD=0 T2H_PIPELINE_TILES=0
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 36 --warmup 9 --skip-cpu-baseline --profile-steps 0 --sustain-s 0 --exact-split-steps 0 --micro-batch-steps 0 --strict-b1-steps 0 > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p --output-format csv -- python3 $R/bench.py --steps 16 --warmup 5 --skip-cpu-baseline --profile-steps 0 --sustain-s 0 --exact-split-steps 0 --micro-batch-steps 0 --strict-b1-steps 0 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o p --output-format csv -- python3 $R/bench.py --steps 16 --warmup 5 --skip-cpu-baseline --profile-steps 0 --sustain-s 0 --exact-split-steps 0 --micro-batch-steps 0 --strict-b1-steps 0 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/probe_fetch -o p --output-format csv -- python3 $R/profiles/pmc_probe.py > /dev/null 2> $OUT/probe_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/probe_write -o p --output-format csv -- python3 $R/profiles/pmc_probe.py > /dev/null 2> $OUT/probe_write.err
rocprofv3 --kernel-trace --stats -d $OUT/trace_infer -o t --output-format csv -- python3 $R/bench.py --mode infer --batch 4 --hip-graph 1 --steps 12 --warmup 4 > $OUT/infer_bench.json 2> $OUT/trace_infer.err
cd $R
unset T2H_OVERLAP_WGRAD T2H_OVERLAP_CONV_WGRAD T2H_PIPELINE_TILES
python3 profiles/collect_pmc.py --bench $OUT/pmc_fetch $OUT/pmc_write --probe $OUT/probe_fetch $OUT/probe_write --tag $TAG > $OUT/pmc_summary.txt 2>&1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python3 profiles/summarize_trace.py $(ls $OUT/trace/*kernel_trace.csv $OUT/trace/*/*kernel_trace.csv 2>/dev/null | head -1) --steps 6 --tiles-per-step 4 --top 70 --tag $TAG --json profiles/rocprof_kernels.json > $OUT/kernel_trace_steady_state.txt 2>&1
cp profiles/rocprof_kernels.json $OUT/rocprof_kernels.json
python3 profiles/summarize_trace.py $(ls $OUT/trace_infer/*kernel_trace.csv $OUT/trace_infer/*/*kernel_trace.csv 2>/dev/null | head -1) --steps 4 --top 40 > $OUT/infer_kernel_trace_steady_state.txt 2>&1
cp $(ls $OUT/trace/*kernel_stats.csv $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/rocprofv3_kernel_stats.csv
cp $(ls $OUT/trace_infer/*kernel_stats.csv $OUT/trace_infer/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/infer_rocprofv3_kernel_stats.csv
# the raw traces are large: keep only what the collectors need
rm -f $OUT/trace/*kernel_trace.csv $OUT/trace/*/*kernel_trace.csv $OUT/trace_infer/*kernel_trace.csv $OUT/trace_infer/*/*kernel_trace.csv
rm -f $OUT/pmc_*/*kernel_trace.csv $OUT/pmc_*/*/*kernel_trace.csv $OUT/probe_*/*kernel_trace.csv $OUT/probe_*/*/*kernel_trace.csv
# the driver's command (the headline line; reads the two json files refreshed above)
python3 bench.py --steps 20 --warmup 5 --kernel-table $OUT/bench_kernels.json > $OUT/bench.json 2> $OUT/bench.err
# secondary lines (same box), each named after the oracle test that checked its configuration (tests/test_full_size_vs_oracle.py,
# tests/test_hip_ragged.py): micro-batched accumulation window; tiles from the device tile producer; BASELINE configs[2]
# (cloud+image, bf16 mode) and its fp32 sibling; the other tile sizes of SURVEY 8d; the no-skew control
python3 bench.py --steps 20 --warmup 5 --coalesce 1 --skip-cpu-baseline --kernel-table $OUT/bench_b1_kernels.json > $OUT/bench_b1.json 2> $OUT/bench_b1.err
python3 bench.py --steps 64 --warmup 24 --coalesce 8 --skip-cpu-baseline --profile-steps 0 --sustain-s 2 > $OUT/bench_b8.json 2> /dev/null      # (warm-up: both tile streams' pools)
python3 bench.py --steps 20 --warmup 5 --from-producer --skip-cpu-baseline --kernel-table $OUT/bench_producer_kernels.json > $OUT/bench_producer.json 2> $OUT/bench_producer.err
python3 bench.py --steps 20 --warmup 5 --use-image --mlp-precision bf16 --skip-cpu-baseline --profile-steps 0 > $OUT/bench_image_bf16.json 2> $OUT/bench_image_bf16.err
python3 bench.py --steps 20 --warmup 5 --use-image --skip-cpu-baseline --profile-steps 0 > $OUT/bench_image_fp32.json 2> $OUT/bench_image_fp32.err
python3 bench.py --steps 20 --warmup 5 --points 65536 --skip-cpu-baseline --profile-steps 0 --sustain-s 2 > $OUT/bench_n65536.json 2> /dev/null
python3 bench.py --steps 20 --warmup 5 --points 262144 --skip-cpu-baseline --profile-steps 0 --sustain-s 2 > $OUT/bench_n262144.json 2> /dev/null
python3 bench.py --steps 20 --warmup 5 --uniform-xy --skip-cpu-baseline --profile-steps 0 --sustain-s 2 > $OUT/bench_uniform.json 2> /dev/null
python3 bench.py --mode infer --batch 1 --steps 12 --warmup 4 > $OUT/infer_bench_b1.json 2> /dev/null
du -sh $OUT; tail -c 700 $OUT/bench.json; echo; head -12 $OUT/kernel_trace_steady_state.txt; head -8 $OUT/infer_kernel_trace_steady_state.txt; tail -3 $OUT/pmc_summary.txt
