"""Markdown table of the nested workloads of two default-run lines (python profiles/workload_table.py <old.json> <new.json>)."""
import json
import sys

old, new = (json.load(open(p)) for p in sys.argv[1:3])
ow, nw = old.get("workloads", {}), new.get("workloads", {})
print("| workload | before (ms) | now (ms) | median of the timed steps | result check | host syncs per step |")
print("|---|---|---|---|---|---|")
print(f"| TPC-H Q3 (headline) | {old['ms_per_step']} | {new['ms_per_step']} | | {new.get('result_check', {}).get('ok')} | {sum(new['roofline']['host_syncs_per_step'].values())} |")
print(f"| TPC-H Q3, clustered-key shortcuts off | {old['q3_general_paths']['ms_per_step']} | {new['q3_general_paths']['ms_per_step']} | | same checksums | |")
for k, v in nw.items():
    o = ow.get(k, {})
    print(f"| `{k}` | {o.get('ms_per_step', '--')} | {v['ms_per_step']} | {v.get('ms_per_step_median', '')} | {v.get('result_check', {}).get('ok')} | {v.get('host_syncs_per_step', '')} |")
