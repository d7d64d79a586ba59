#!/bin/bash
# After `bash profiles/run_profiles.sh TAG` on the GPU box: copy what gpurun merged back into gpurun_out/TAG/ to the tracked names
# profiles/TAG_* (and the two collector files bench.py reads).      bash profiles/install_profiles.sh r06
set -eu
TAG=${1:?tag}
S=gpurun_out/$TAG
for f in bench.json bench_kernels.json bench_b1.json bench_b1_kernels.json bench_b8.json bench_producer.json bench_image_bf16.json \
         bench_image_fp32.json bench_n65536.json bench_n262144.json bench_uniform.json infer_bench.json infer_bench_b1.json \
         kernel_trace_steady_state.txt infer_kernel_trace_steady_state.txt rocprofv3_kernel_stats.csv infer_rocprofv3_kernel_stats.csv \
         pmc_summary.txt; do
    [ -s $S/$f ] && cp $S/$f profiles/${TAG}_$f || echo "missing: $S/$f"
done
cp $S/rocprof_kernels.json profiles/rocprof_kernels.json
cp $S/pmc_traffic.json profiles/pmc_traffic.json
