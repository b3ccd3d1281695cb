#!/usr/bin/env python3
"""HBM bytes per step of single-workload runs out of the passes profiles/collect_r04.sh writes (<tag>_fetch.csv, <tag>_write.csv: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
one counter per run, nothing else traced with it; <tag>_stats.csv: --kernel-trace --stats of the same command).  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024
(gfx950 tallies a 128-B read request at 64 B: /opt/skills/guides/MI355X_MICROARCH.md, HBM section).  Every run makes `steps` steps (warm-up + the profiled step + timed).
usage: traffic_workloads.py <dir> <out.json> <steps> tag=workload:algorithmic_bytes_per_step ..."""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot = defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"dfgpu::(?:pq::)?(k_[A-Za-z0-9_]+)", r["Kernel_Name"])
        if m:
            tot[m.group(1)] += float(r["Counter_Value"])
    return tot


def main():
    d, out_path, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out = {"_note": "HBM bytes per step of workloads run alone in a process (bench_workloads.py --only <name> --sf 100), from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; "
                    "HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024; per step = all dfgpu launches of the run / its %d steps; algorithmic_bytes = the workload's own figure "
                    "(input rows x bytes of the columns the plan reads)" % steps}
    for spec in sys.argv[4:]:
        tag, rest = spec.split("=", 1); name, algo = rest.rsplit(":", 1); algo = int(algo)
        fetch, write = per_kernel(f"{d}/{tag}_fetch.csv", "FETCH_SIZE"), per_kernel(f"{d}/{tag}_write.csv", "WRITE_SIZE")
        per = {k: (2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024 / steps for k in set(fetch) | set(write)}
        total = sum(per.values())
        ms = None
        try:
            rows = list(csv.DictReader(open(f"{d}/{tag}_stats.csv")))
            ms = round(sum(float(r["TotalDurationNs"]) for r in rows if "dfgpu::" in r["Name"]) / steps / 1e6, 3)
        except OSError:
            pass
        out[name] = {"hbm_bytes_per_step": int(total), "algorithmic_bytes": algo, "traffic_ratio": round(total / algo, 2), "dfgpu_kernel_ms_per_step_under_rocprof_stats": ms,
                     "largest_kernels_bytes_per_step": {k: int(v) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:8]}}
    json.dump(out, open(out_path, "w"), indent=1)
    print({k: (v["hbm_bytes_per_step"], v["traffic_ratio"]) for k, v in out.items() if isinstance(v, dict)})


if __name__ == "__main__":
    main()
