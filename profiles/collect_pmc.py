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
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from probe_manifest import PROBE_GROUPS, REPS            # noqa: E402  (shared with profiles/pmc_probe.py: one definition)


def probe_sequence(rows):
    """The probe's dispatches in order -> {tag: [mean counter value per part]}.  STRICT: the dispatches that belong to the probe's
    ops (any kernel-name substring of the manifest) must be exactly the manifest's sequence, REPS times per group -- a reordered
    or added launch in pmc_probe.py, or a kernel-selection change in the library, fails here instead of attributing counters to
    the wrong tag."""
    subs = sorted({sub for group in PROBE_GROUPS for _, parts in group for sub in parts}, key=len, reverse=True)
    mine = [r for r in rows if any(sub in r["Kernel_Name"] for sub in subs)]
    expected = [(tag, j, sub) for group in PROBE_GROUPS for _ in range(REPS) for tag, parts in group for j, sub in enumerate(parts)]
    if len(mine) != len(expected):
        raise SystemExit(f"collect_pmc: the probe issued {len(mine)} dispatches of the manifest's kernels, the manifest "
                         f"(profiles/probe_manifest.py) expects {len(expected)}: update the manifest with the probe")
    out = {tag: [0.0] * len(parts) for group in PROBE_GROUPS for tag, parts in group}
    for r, (tag, j, sub) in zip(mine, expected):
        if sub not in r["Kernel_Name"]:
            raise SystemExit(f"collect_pmc: dispatch {r['Dispatch_Id']} is {r['Kernel_Name'][:80]!r}, the manifest expects "
                             f"{sub!r} for {tag}: launch order and manifest disagree")
        out[tag][j] += float(r["Counter_Value"]) / REPS
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
