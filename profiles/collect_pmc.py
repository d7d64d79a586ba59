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

PROBE_OPS = {   # bench.py tag -> kernel-name substrings of one launch of the op, in the probe's order
    "t2h_segmean_fwd[C=512,r=32]": ["segmean_cells_kernel", "segmean_finalize_kernel"],
    "t2h_sample_fwd[C=512,r=32]": ["sample_fwd_kernel"],
    "t2h_sample_bwd[C=512,r=32]": ["sample_bwd_cells_", "sample_bwd_gather9_kernel"],
    # r03, deferred point update: per-cell sums of the widest hidden activations at the finest resolution
    "t2h_segsum_fwd[C=1024,r=256]": ["segmean_fwd_kernel<4, false>"],
    "t2h_segsum_bwd_multi[C=1024,n=4]": ["segsum_bwd_multi_kernel"],
}
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
        fr, wr = rows_of(a.probe[0], "FETCH_SIZE"), rows_of(a.probe[1], "WRITE_SIZE")
        f, _ = mean_by(fr, lambda n: n)
        w, _ = mean_by(wr, lambda n: n)
        for tag, parts in PROBE_OPS.items():
            total, rows = 0.0, []
            for sub in parts:
                kf, kw = [k for k in f if sub in k], [k for k in w if sub in k]
                if not kf or not kw:
                    total = None
                    break
                rows.append({"kernel": sub, "FETCH_SIZE_KiB": round(f[kf[0]], 1), "WRITE_SIZE_KiB": round(w[kw[0]], 1)})
                total += (2.0 * f[kf[0]] + w[kw[0]]) * 1024.0
            if total is not None:
                out[tag] = int(total)
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
