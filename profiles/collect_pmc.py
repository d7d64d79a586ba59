#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes over profiles/pmc_probe.py (FETCH_SIZE, WRITE_SIZE) into profiles/pmc_traffic.json.

Units/corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-byte
streaming stores.  Per launch = mean over the probe's repetitions; ops made of two kernels are summed."""
import collections
import csv
import glob
import json
import os
import sys

OPS = {   # bench.py kernel tag -> kernel-name substrings that make up one launch of the op ("#k": k-th probe op using it)
    "t2h_linear_fwd[K=512,N=1024]": ["gemm_dma_kernel"],
    "t2h_linear_fwd[K=1024,N=512]": ["gemm_dma_kernel#2"],
    "t2h_linear_dgrad[N=1024,K=512]": ["gemm_dma_nn_kernel"],
    "t2h_linear_dgrad[N=512,K=1024]": ["gemm_dma_nn_kernel#2"],
    "t2h_linear_wgrad[N=1024,K=512]": ["gemm_kernel<128, 128, 2, 2, false, false", "reduce_slabs_kernel"],
    "t2h_linear_wgrad[N=512,K=1024]": ["gemm_kernel<128, 128, 2, 2, false, false#2", "reduce_slabs_kernel#2"],
    "t2h_segmean_fwd[C=512,r=32]": ["segmean_cells_kernel", "segmean_finalize_kernel"],
    "t2h_pool_max_fwd": ["pool_max_fwd_kernel"],
    "t2h_sample_fwd[C=512,r=32]": ["sample_fwd_kernel"],
    "t2h_sample_bwd[C=512,r=32]": ["sample_bwd_cells_kernel", "sample_bwd_gather9_kernel"],
    "t2h_conv3x3_fwd[64->128,512x512]": ["conv_rows_kernel<128, 2, 2, 0"],
    "t2h_conv3x3_dgrad[128->64,512x512]": ["conv_rows_kernel<64, 2, 2, 1"],
    "t2h_conv3x3_wgrad[64->128,512x512]": ["conv_wgrad_kernel<128, 128, 2, 2, 4, false", "reduce_slabs_kernel#3"],
}


REPS = 3      # profiles/pmc_probe.py runs every op this many times, one launch of each kernel per run


def read_counter(folder, counter):
    """mean counter value per kernel name; a kernel that serves several probe ops (REPS dispatches each) is split by
    dispatch order into name, name#2, name#3 ..."""
    rows = []
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    seen, sums, counts = collections.defaultdict(int), collections.defaultdict(float), collections.defaultdict(int)
    for r in rows:
        name = r["Kernel_Name"]
        group = seen[name] // REPS
        seen[name] += 1
        key = name if group == 0 else f"{name}#{group + 1}"
        sums[key] += float(r["Counter_Value"])
        counts[key] += 1
    return {k: sums[k] / counts[k] for k in sums}, counts


def _suffix(key):
    return int(key.rsplit("#", 1)[1]) if "#" in key and key.rsplit("#", 1)[1].isdigit() else 1


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    fetch, nf = read_counter(fetch_dir, "FETCH_SIZE")
    write, nw = read_counter(write_dir, "WRITE_SIZE")
    out, detail = {}, {}
    for tag, parts in OPS.items():
        total, rows = 0.0, []
        for sub in parts:
            want = _suffix(sub)
            stem = sub.rsplit("#", 1)[0] if want > 1 else sub
            kf = [k for k in fetch if stem in k and _suffix(k) == want]
            kw = [k for k in write if stem in k and _suffix(k) == want]
            if not kf or not kw:
                total = None
                break
            f_kib = max(fetch[k] for k in kf)
            w_kib = max(write[k] for k in kw)
            rows.append({"kernel": sub, "FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib})
            total += (2.0 * f_kib + w_kib) * 1024.0
        if total is not None:
            out[tag] = int(total)
            detail[tag] = rows
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "pmc_traffic.json"), "w") as f:
        json.dump({"workload": "BASELINE.json configs[1], N=131072, profiles/pmc_probe.py",
                   "formula": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts half of wide reads)",
                   "bytes_per_launch": out, "detail": detail}, f, indent=1)
    for k, v in out.items():
        print(f"{k:<36s} {v / 1e6:10.1f} MB / launch")


if __name__ == "__main__":
    main()
