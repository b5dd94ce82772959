#!/usr/bin/env python3
"""Concurrency in a rocprofv3 kernel trace: per queue the busy time, the union of all kernel intervals, and for the first
kernels after a given kernel name the (queue, start, end, name) rows -- do kernels of different queues overlap?

    python3 tools/timeline_overlap.py /tmp/tl [first_kernel_substring] [rows]"""
import csv
import glob
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:44]


d = sys.argv[1]
rows = []
for p in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
rows.sort()
busy = {}
for s, e, q, n in rows:
    busy[q] = busy.get(q, 0) + e - s
union, cur_s, cur_e = 0, None, None
for s, e, q, n in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print("queues busy ms:", {q: round(v / 1e6, 1) for q, v in busy.items()}, "sum", round(sum(busy.values()) / 1e6, 1), "union", round(union / 1e6, 1))
key = sys.argv[2] if len(sys.argv) > 2 else "conv3_px"
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 80
hits = [i for i, r in enumerate(rows) if key in r[3]]
i0 = hits[len(hits) // 2]
t0 = rows[i0][0]
for s, e, q, n in rows[i0:i0 + nrows]:
    print(f"q{q:>3s} {(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} us  {n}")
