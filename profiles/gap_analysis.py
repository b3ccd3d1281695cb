#!/usr/bin/env python3
"""Where the device idles: reads a rocprofv3 --kernel-trace CSV (one row per dispatch with Start_Timestamp / End_Timestamp in ns), orders the
dispatches by start time and reports busy time, idle time, and the idle time by gap class and by the kernel that ran BEFORE the gap (a host
read-back shows up as a long gap after the kernel whose result the host waited for).
usage: gap_analysis.py <kernel_trace.csv> [--skip-first-frac 0.3] [--top 25]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    skip = float(sys.argv[sys.argv.index("--skip-first-frac") + 1]) if "--skip-first-frac" in sys.argv else 0.0
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
    rows.sort()
    rows = rows[int(len(rows) * skip):]
    span = rows[-1][1] - rows[0][0]
    busy = 0
    gaps = []
    cur_end = rows[0][0]
    prev = None
    for s, e, name in rows:
        if s > cur_end:
            gaps.append((s - cur_end, prev, name))
        busy += max(0, e - max(s, cur_end))
        if e > cur_end:
            cur_end, prev = e, name
    idle = span - busy
    print(f"dispatches {len(rows)}  span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {idle / 1e6:.3f} ms ({100.0 * idle / span:.1f} %)")
    classes = [(0, 5e3, "< 5 us"), (5e3, 15e3, "5-15 us"), (15e3, 40e3, "15-40 us"), (40e3, 100e3, "40-100 us"), (100e3, 1e18, ">= 100 us")]
    for lo, hi, label in classes:
        g = [x[0] for x in gaps if lo <= x[0] < hi]
        print(f"  gaps {label:>10}: {len(g):6d}  total {sum(g) / 1e6:8.3f} ms")
    by_prev = defaultdict(lambda: [0, 0])
    for g, p, n in gaps:
        by_prev[(p, n)][0] += 1; by_prev[(p, n)][1] += g
    print("idle by (kernel before the gap -> kernel after), largest first:")
    for (p, n), (c, t) in sorted(by_prev.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"  {t / 1e6:8.3f} ms  x{c:5d}  avg {t / c / 1e3:7.1f} us   {p}  ->  {n}")


if __name__ == "__main__":
    main()
