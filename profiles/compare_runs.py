#!/usr/bin/env python3
"""Markdown table of every workload of two default bench runs (bench_detail.json objects): before / now, median step, result check, host read-backs.
usage: compare_runs.py <before.json> <now.json>"""
import json
import sys

a, b = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))


def syncs(d):
    h = (d.get("roofline") or {}).get("host_syncs_per_step")
    return sum(h.values()) if isinstance(h, dict) else h


print("| workload | before (ms) | now (ms) | median of the timed steps | result check | host read-backs per step (before -> now) |")
print("|---|---|---|---|---|---|")
print(f"| TPC-H Q3 (headline) | {a['ms_per_step']} | {b['ms_per_step']} | | {b['result_check']['ok']} | {syncs(a)} -> {syncs(b)} |")
for key, label in (("q3_general_paths", "TPC-H Q3, clustered-key shortcuts off"), ("q3_shuffled_inputs", "TPC-H Q3 over row-wise permuted tables")):
    if a.get(key) and b.get(key):
        print(f"| {label} | {a[key]['ms_per_step']} | {b[key]['ms_per_step']} | | same checksums | {a[key].get('host_syncs_per_step', '')} -> {b[key].get('host_syncs_per_step', '')} |")
for k, w in b["workloads"].items():
    o = a["workloads"].get(k, {})
    print(f"| `{k}` | {o.get('ms_per_step', '--')} | {w['ms_per_step']} | {w.get('ms_per_step_median', '')} | {w.get('result_check', {}).get('ok')} | {o.get('host_syncs_per_step', '--')} -> {w.get('host_syncs_per_step')} |")
