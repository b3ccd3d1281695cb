"""Per-kernel HBM bytes per launch from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter unit KB).
HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE: gfx950 counts a 128-B read request as 64 B (/opt/skills/guides/MI355X_MICROARCH.md, HBM section);
checked against k_compare_scalar_fast, which reads exactly 4 B per row.  Only dfgpu kernels of the bench's steps are kept.
usage: python profiles/pmc_to_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"dfgpu::(k_[a-z0-9_]+)", r["Kernel_Name"])
        if not m:
            continue
        tot[m.group(1)] += float(r["Counter_Value"]); cnt[m.group(1)] += 1
    return {k: tot[k] / cnt[k] for k in tot}, dict(cnt)


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_note": "HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KB*1024), averaged over the launches of bench.py steps (SF100, 1 GPU); separate rocprofv3 --pmc passes; "
                "factor 2 on FETCH_SIZE per /opt/skills/guides/MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B), consistent with k_compare_scalar_fast: 4 B x rows read"}
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
    out[k] = int((2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print({k: v for k, v in list(out.items())[1:8]})
