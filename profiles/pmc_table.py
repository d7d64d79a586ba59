#!/usr/bin/env python3
"""Pivot rocprofv3 --pmc output (one or more output folders, one counter pass each) into a per-kernel table: mean counter values per
dispatch of each kernel symbol, plus the ratios that answer "what is this kernel waiting for".

    python profiles/pmc_table.py DIR [DIR ...] [--match bx3_rows] [--last N]
"""
import argparse
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_trace import short_symbol  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--match", default="")
    ap.add_argument("--last", type=int, default=0, help="only the last N dispatches of each kernel (steady state)")
    a = ap.parse_args()
    vals = collections.defaultdict(lambda: collections.defaultdict(list))       # kernel -> counter -> [values in dispatch order]
    grid = {}
    for d in a.dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
            for r in rows:
                k = short_symbol(r["Kernel_Name"])
                if a.match and a.match not in k:
                    continue
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                grid[k] = r.get("Grid_Size", "?")
    counters = sorted({c for k in vals for c in vals[k]})
    print(f"{'kernel':<46s} {'grid':>9s} " + " ".join(f"{c[-18:]:>18s}" for c in counters))
    for k in sorted(vals):
        row = []
        for c in counters:
            v = vals[k].get(c, [])
            v = v[-a.last:] if a.last else v
            row.append(sum(v) / len(v) if v else float("nan"))
        print(f"{k[:46]:<46s} {grid[k]:>9s} " + " ".join(f"{x:18.4g}" for x in row))


if __name__ == "__main__":
    main()
