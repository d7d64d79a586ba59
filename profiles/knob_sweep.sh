#!/bin/bash
# r06: one-box A/B of the launch-geometry knobs at the Trainer's default coalescing (4 tiles per forward / backward): each line is
# `bench.py` with one environment variable changed; sustained tiles/s over ~4 s (same box, back to back).  Output: gpurun_out/knobs.txt
mkdir -p gpurun_out
OUT=gpurun_out/knobs.txt
: > $OUT
run() {
  label="$1"; shift
  line=$(env "$@" python bench.py --gpus 1 --steps 20 --warmup 5 --skip-cpu-baseline --exact-split-steps 0 --strict-b1-steps 0 --profile-steps 0 --sustain-s 4 2>/dev/null | tail -1)
  python - "$label" "$line" >> $OUT <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2])
    print(f"{sys.argv[1]:44s} value {d['value']:8.2f}  sustained {d['sustained']['tiles_per_s']:8.2f} tiles/s  (median {d['sustained']['median_ms']:.3f} ms)")
except Exception as e:
    print(f"{sys.argv[1]:44s} FAILED {e}")
PY
}
run "default" T2H_NOP=1
for kv in "$@"; do run "$kv" $kv; done
run "default (again)" T2H_NOP=1
cat $OUT
