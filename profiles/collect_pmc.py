#!/usr/bin/env python3
"""HBM traffic per launch from rocprofv3 PMC passes -> profiles/pmc_traffic.json (read by bench.py for `roofline.traffic`).

Two sources, each collected as two separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass):

  bench passes  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d F -- python3 bench.py --steps 6 --warmup 3 \
                    --skip-cpu-baseline --profile-steps 0        (and the same with WRITE_SIZE -> W)
                -> per DEVICE KERNEL SYMBOL, mean over the launches of the last tile-steps (steady state, delimited by
                   tile_keys_kernel): the key bench.py's `roofline` uses (t2h_last_kernel_name), e.g. gemm_dma_kernel
  probe passes  the same two passes over profiles/pmc_probe.py -> per ENTRY-POINT TAG for ops that share a kernel
                symbol across shapes (the largest scatter_mean, the coarse sample kernels)

    python profiles/collect_pmc.py --bench F W [--probe PF PW]

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly half the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-byte streaming
stores: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import argparse
import collections
import csv
import glob
import json
import os
import re

REPS = 3        # profiles/pmc_probe.py launches every group this many times in a row
# the probe's launches in order: groups of (bench.py tag, kernel-name substrings of one launch of the op); ops that share a
# kernel symbol (the pixel gather of both sample adjoints) are told apart by their position in the dispatch sequence
PROBE_GROUPS = [
    [("t2h_segmean_fwd[C=512,r=32]", ["segmean_cells_kernel", "segmean_finalize_kernel"])],
    [("t2h_sample_fwd[C=512,r=32]", ["sample_fwd_kernel"]),
     ("t2h_sample_bwd[C=512,r=32]", ["sample_bwd_cells_", "sample_bwd_gather9_kernel"])],
    # r03, deferred point update: per-cell sums of the widest hidden activations at the finest resolution
    [("t2h_segsum_fwd[C=1024,r=256]", ["segmean_fwd_kernel<4, false>"])],
    [("t2h_segsum_bwd_multi[C=1024,n=4]", ["segsum_bwd_multi_kernel"])],
    # r03, hidden activations on chip: sample + ReLU + per-cell sums + sign bits; the backward walk + its pixel gather
    [("t2h_sample_relu_cellsums[C=1024,r=32]", ["sample_relu_cellsums_kernel"])],
    [("t2h_sample_bwd_from_sums[C=1024,r=32]", ["sample_bwd_walk_kernel", "sample_bwd_gather9_kernel"])],
]


def probe_sequence(rows):
    """Walk the probe's dispatches in order -> {tag: [mean counter value per part]} (None where the sequence does not match)."""
    out, i = {}, 0
    for group in PROBE_GROUPS:
        acc = {tag: [0.0] * len(parts) for tag, parts in group}
        ok = True
        for _ in range(REPS):
            for tag, parts in group:
                for j, sub in enumerate(parts):
                    while i < len(rows) and sub not in rows[i]["Kernel_Name"]:
                        i += 1
                    if i >= len(rows):
                        ok = False
                        break
                    acc[tag][j] += float(rows[i]["Counter_Value"]) / REPS
                    i += 1
        for tag, _ in group:
            out[tag] = acc[tag] if ok else None
        if not ok:
            break
    return out


# entry points of point_grid.hip do not note a kernel symbol: bench.py keys them by entry-point name
ENTRY_OF = {"sample_fwd_kernel": "t2h_sample_fwd", "segmean_bwd_kernel": "t2h_segmean_bwd"}


def short(name):
    """rocprofv3's demangled name -> the symbol t2h_last_kernel_name() reports."""
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*$", "", n)                            # argument list
    return n.replace("t2h::", "").replace(" ", "")


def rows_of(folder, counter):
    rows = []
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def steady(rows, steps):
    starts = [i for i, r in enumerate(rows) if "tile_keys_kernel" in r["Kernel_Name"]]
    if len(starts) <= steps:
        return rows
    return rows[starts[-(steps + 1)]:starts[-1]]


def mean_by(rows, key):
    s, c = collections.defaultdict(float), collections.defaultdict(int)
    for r in rows:
        k = key(r["Kernel_Name"])
        s[k] += float(r["Counter_Value"])
        c[k] += 1
    return {k: s[k] / c[k] for k in s}, c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bench", nargs=2, metavar=("FETCH_DIR", "WRITE_DIR"))
    ap.add_argument("--probe", nargs=2, metavar=("FETCH_DIR", "WRITE_DIR"))
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--tag", default="?", help="profile tag the passes belong to (recorded; bench.py prints it as the source)")
    a = ap.parse_args()
    out, detail = {}, {}
    if a.bench:
        f, nf = mean_by(steady(rows_of(a.bench[0], "FETCH_SIZE"), a.steps), short)
        w, _ = mean_by(steady(rows_of(a.bench[1], "WRITE_SIZE"), a.steps), short)
        for k in f:
            if k in w:
                out[k] = int((2.0 * f[k] + w[k]) * 1024.0)
                detail[k] = {"FETCH_SIZE_KiB": round(f[k], 1), "WRITE_SIZE_KiB": round(w[k], 1),
                             "launches_per_step": round(nf[k] / a.steps, 2), "source": "bench.py steady state"}
                if k.split("<")[0] in ENTRY_OF:
                    out[ENTRY_OF[k.split("<")[0]]] = out[k]
    if a.probe:
        f = probe_sequence(rows_of(a.probe[0], "FETCH_SIZE"))
        w = probe_sequence(rows_of(a.probe[1], "WRITE_SIZE"))
        for group in PROBE_GROUPS:
            for tag, parts in group:
                if f.get(tag) is None or w.get(tag) is None:
                    continue
                rows = [{"kernel": sub, "FETCH_SIZE_KiB": round(f[tag][j], 1), "WRITE_SIZE_KiB": round(w[tag][j], 1)}
                        for j, sub in enumerate(parts)]
                out[tag] = int(sum((2.0 * f[tag][j] + w[tag][j]) * 1024.0 for j in range(len(parts))))
                detail[tag] = {"parts": rows, "source": "profiles/pmc_probe.py"}
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "pmc_traffic.json"), "w") as fjs:
        json.dump({"workload": "BASELINE.json configs[1], N=131072", "tag": a.tag,
                   "formula": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts half of wide reads)",
                   "bytes_per_launch": out, "detail": detail}, fjs, indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1])[:40]:
        print(f"{k:<60s} {v / 1e6:10.1f} MB / launch")


if __name__ == "__main__":
    main()
