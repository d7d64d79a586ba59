#!/usr/bin/env python3
"""Steady-state per-step kernel summary from a rocprofv3 --kernel-trace CSV of bench.py.

rocprofv3's own --stats aggregates the whole process, which for this workload is dominated by MIOpen's
one-time naive_conv warm-up; steps are delimited here by t2h::tile_keys_kernel (one launch per tile) and only
the last `--steps` tiles are summarised.

    python profiles/summarize_trace.py <kernel_trace.csv> [--steps 6] [--top 45]
"""
import argparse
import collections
import csv
import json
import re


def short_symbol(k: str) -> str:
    """A rocprofv3 kernel name as bench.py's kernel tables spell the symbol (t2h_last_kernel_name): no return type, namespaces,
    argument list or blanks -- 'void t2h::(anonymous namespace)::bx3_rows_kernel<4, 128, 2, 2, 32, 3, 9>(t2h::...)' ->
    'bx3_rows_kernel<4,128,2,2,32,3,9>'."""
    k = re.sub(r"^void ", "", k)
    depth, out = 0, []
    for ch in k:                      # cut at the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0 and out and "".join(out).rstrip().endswith(("kernel", ">")):
            break
        out.append(ch)
    k = "".join(out).replace("(anonymous namespace)::", "").replace("t2h::", "").replace(" ", "")
    return k


def category(k):
    if "t2h::" in k:
        return "t2h hand-written HIP"
    if k.startswith("Cijk"):
        return "rocBLAS/hipBLASLt GEMM"
    if any(s in k for s in ("conv", "igemm", "Sp3Asm", "batched_transpose", "gemm_xdlops_bwd_weight", "Conv")):
        return "MIOpen conv"
    if "at::native" in k or "elementwise" in k:
        return "torch elementwise/reduce/cat"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--json", default="", help="also write {tag, steps, kernels: {symbol: {avg_us, launches_per_step}}} here")
    ap.add_argument("--tag", default="")
    ap.add_argument("--tiles-per-step", type=int, default=1,
                    help="tiles per delimited step (r06: the Trainer coalesces 4 tiles per forward / backward = one tile_keys launch); "
                         "the ms/step and launches/step columns are per such step, the headline also per tile")
    a = ap.parse_args()
    rows = sorted(csv.DictReader(open(a.trace)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "tile_keys_kernel" in r["Kernel_Name"]]
    lo, hi = starts[-(a.steps + 1)], starts[-1]
    sel = rows[lo:hi]
    wall = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e6 / a.steps
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in sel:
        agg[r["Kernel_Name"]][0] += 1
        agg[r["Kernel_Name"]][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy = sum(v[1] for v in agg.values()) / 1e3 / a.steps
    tps = max(1, a.tiles_per_step)
    print(f"steady state over the last {a.steps} steps of {tps} tile(s): wall {wall:.2f} ms/step, kernel-busy {busy:.2f} ms/step, "
          f"{len(sel) / a.steps:.0f} launches/step  =  {wall / tps:.2f} ms, {busy / tps:.2f} ms kernel-busy, {len(sel) / a.steps / tps:.0f} launches per TILE")
    cat = collections.defaultdict(float)
    for k, (c, t) in agg.items():
        cat[category(k)] += t
    for k, v in sorted(cat.items(), key=lambda kv: -kv[1]):
        print(f"  {k:<34s} {v / 1e3 / a.steps:9.3f} ms/step")
    print()
    print(f"{'kernel':<100s} {'launches/step':>13s} {'avg us':>10s} {'ms/step':>9s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f"{k[:100]:<100s} {c / a.steps:13.1f} {t / c:10.1f} {t / 1e3 / a.steps:9.3f}")
    if a.json:
        kernels = {}
        for k, (c, t) in agg.items():
            e = kernels.setdefault(short_symbol(k), {"us": 0.0, "launches": 0})
            e["us"] += t
            e["launches"] += c
        out = {"tag": a.tag, "steps": a.steps, "tiles_per_step": tps, "wall_ms_per_step": round(wall, 3), "busy_ms_per_step": round(busy, 3),
               "launches_per_step": round(len(sel) / a.steps, 1),
               "kernels": {k: {"avg_us": round(v["us"] / v["launches"], 2), "launches_per_step": round(v["launches"] / a.steps, 2)}
                           for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["us"])}}
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
