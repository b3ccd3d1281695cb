#!/usr/bin/env python3
"""One steady-state Q3 step out of a rocprofv3 --kernel-trace CSV: the dispatches between two consecutive k_dict_predicate launches (the customer filter opens every step),
each with the idle gap before it and its duration; first the per-step totals of the steady-state steps.  usage: step_timeline.py <kernel_trace.csv> [first-kernel-substring]"""
import csv
import sys


def main():
    path = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "k_dict_predicate"
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:64]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if first in r[2]]
    steps = []
    for a, b in zip(starts, starts[1:]):
        seg = rows[a:b]
        steps.append((rows[b][0] - rows[a][0], sum(e - s for s, e, _ in seg), len(seg), a, b))
    steady = sorted(steps)[: max(1, len(steps) * 2 // 3)]           # the faster two thirds: steps of one plan shape without the profiled / general-path passes
    n = len(steady)
    print(f"steps {len(steps)}, steady {n}: wall {sum(s[0] for s in steady) / n / 1e3:.1f} us, busy {sum(s[1] for s in steady) / n / 1e3:.1f} us, dispatches {sum(s[2] for s in steady) / n:.1f}")
    _, _, _, a, b = sorted(steady)[n // 2]
    prev_end = None
    for s, e, name in rows[a:b]:
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print(f"{gap:8.1f} us gap | {(e - s) / 1e3:8.1f} us  {name}")
        prev_end = e


if __name__ == "__main__":
    main()
