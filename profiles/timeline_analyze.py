#!/usr/bin/env python3
"""Overlap analysis of a rocprofv3 --kernel-trace CSV of the pipelined training step: per hardware queue (stream) busy time,
how many kernels run at once, and the time during which ONLY small launches (< 25 % of the chip's workgroup slots) are running.
    python3 profiles/timeline_analyze.py <kernel_trace.csv> [--steps 8]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        try:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        except (KeyError, ValueError):
            continue
        wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", "256")) or 256)
        grid = int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0)
        rows.append((s, e, r.get("Queue_Id", "?"), r.get("Kernel_Name", "?"), grid // max(wg, 1)))
rows.sort()
# steady state: the last third of the trace
t_lo = rows[len(rows) * 2 // 3][0]
sel = [r for r in rows if r[0] >= t_lo]
t0, t1 = sel[0][0], max(r[1] for r in sel)
wall = (t1 - t0) / 1e6
print(f"{len(sel)} launches in {wall:.2f} ms of steady state")
busy = defaultdict(float)
for s, e, q, _, _ in sel:
    busy[q] += (e - s) / 1e6
for q, b in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"  queue {q}: busy {b:8.2f} ms = {b / wall:5.1%} of the wall time")
# concurrency histogram by sweeping the events
ev = []
for s, e, q, n, wgs in sel:
    ev.append((s, 1, wgs)); ev.append((e, -1, wgs))
ev.sort()
hist = defaultdict(float)
small_only = 0.0
live, live_wgs, last = 0, 0, ev[0][0]
for t, d, wgs in ev:
    hist[live] += (t - last) / 1e6
    if live > 0 and live_wgs < 512:
        small_only += (t - last) / 1e6
    last = t
    live += d
    live_wgs += d * wgs
print("kernels running at once -> share of the wall time")
for k in sorted(hist):
    print(f"  {k}: {hist[k] / wall:6.1%}")
print(f"time with fewer than 512 workgroups in flight in total (chip half-empty): {small_only / wall:.1%}")
print(f"sum of kernel durations / wall = {sum(busy.values()) / wall:.2f}")
