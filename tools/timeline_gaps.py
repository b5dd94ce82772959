#!/usr/bin/env python3
"""Where a step's wall time goes that no kernel accounts for: reads rocprofv3's kernel trace (csv) of one command and
prints, per kernel name, calls / total / average duration, then the device-idle time between consecutive kernels
grouped by the kernel that FOLLOWS the gap (the launch the host was late with).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python3 $ROOT/bench.py ...
    python3 tools/timeline_gaps.py /tmp/tl [--from-kernel NAME] > gpurun_out/timeline.txt
"""
import csv
import glob
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:60]


def main():
    d = sys.argv[1]
    paths = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    if not paths:
        raise SystemExit(f"no kernel_trace.csv under {d}")
    rows = []
    for p in paths:
        for r in csv.DictReader(open(p)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    t_first, t_last = rows[0][0], max(r[1] for r in rows)
    busy = {}
    gaps = {}
    end = rows[0][0]
    for s, e, n in rows:
        b = busy.setdefault(n, [0, 0])
        b[0] += 1
        b[1] += e - s
        if s > end:
            g = gaps.setdefault(n, [0, 0, 0])
            g[0] += 1
            g[1] += s - end
            g[2] = max(g[2], s - end)
        end = max(end, e)
    tot_busy = sum(b[1] for b in busy.values())
    tot_gap = sum(g[1] for g in gaps.values())
    print(f"span {(t_last - t_first) / 1e6:.1f} ms, kernel time {tot_busy / 1e6:.1f} ms, idle between kernels {tot_gap / 1e6:.1f} ms "
          f"({len(rows)} launches)")
    print("-- kernels by time")
    for n, (c, t) in sorted(busy.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{n:62s} x{c:5d} {t / 1e6:9.2f} ms  avg {t / c / 1e3:9.1f} us")
    print("-- idle time by the kernel that follows the gap")
    for n, (c, t, m) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{n:62s} x{c:5d} {t / 1e6:9.2f} ms  avg {t / c / 1e3:9.1f} us  max {m / 1e3:9.1f} us")


if __name__ == "__main__":
    main()
