"""Per-kernel HBM bytes from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter unit KB) over a bench_workloads.py run.
HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (gfx950 tallies a 128-B read request at 64 B: /opt/skills/guides/MI355X_MICROARCH.md, HBM section).
Reported per kernel as bytes per launch and bytes per STEP (all launches / the steps the run made), because one step launches a kernel on
inputs of different sizes (build side 15 M rows, probe side 150 M rows).
usage: python profiles/pmc_workloads.py <fetch counter_collection.csv> <write counter_collection.csv> <steps in the run> <out.json> [note]"""
import csv, json, re, sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"dfgpu::(?:pq::)?(k_[A-Za-z0-9_]+)", r["Kernel_Name"])
        if not m:
            continue
        tot[m.group(1)] += float(r["Counter_Value"]); cnt[m.group(1)] += 1
    return tot, cnt


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
out = {"_note": (sys.argv[5] + "  " if len(sys.argv) > 5 else "") + "HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024 from separate rocprofv3 --pmc passes; per_step = all launches of the run / its %d steps" % steps}
total = 0
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
    b = (2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024
    n = max(nf.get(k, 0), nw.get(k, 0))
    out[k] = {"bytes_per_launch": int(b / n), "launches": n, "bytes_per_step": int(b / steps), "read_bytes_per_step": int(2 * fetch.get(k, 0.0) * 1024 / steps), "write_bytes_per_step": int(write.get(k, 0.0) * 1024 / steps)}
    total += b / steps
out["_total_bytes_per_step"] = int(total)
json.dump(out, open(sys.argv[4], "w"), indent=1)
print("total GB per step", round(total / 1e9, 3)); print({k: round(v["bytes_per_step"] / 1e9, 3) for k, v in list(out.items()) if isinstance(v, dict)})
