"""Summarise a rocprofv3 --kernel-trace CSV: kernels per queue, and how much of the busy time has >= 2 kernels in flight.
usage: python scripts/probes/trace_overlap.py <kernel_trace.csv>"""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
q = Counter((r.get("Queue_Id"), r.get("Stream_Id")) for r in rows)
print("kernels per (queue, stream):", dict(q))
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth, last, busy, over = 0, None, 0, 0
for t, d in ev:
    if last is not None:
        if depth >= 1: busy += t - last
        if depth >= 2: over += t - last
    depth += d; last = t
print(f"busy {busy/1e6:.3f} ms, of which >= 2 kernels in flight {over/1e6:.3f} ms ({100*over/max(busy,1):.1f} %)")
