#!/usr/bin/env python3
"""TPC-H Q3 over row-wise permuted tables, alone in a process (what bench.py reports as q3_shuffled_inputs), for the rocprofv3 passes of collect_r04.sh: no key column is
sorted, no join or group key arrives clustered -- the regime behind a hash repartition.  usage: q3_shuffled_run.py [--sf 100] [--steps 5] [--option k=v ...]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=100.0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--option", action="append", default=[])
    args = ap.parse_args()
    import torch
    import dfgpu
    from dfgpu import physical_plan as ops, tpch
    torch.cuda.set_device(0)
    ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    for kv in args.option:
        k, v = kv.split("=", 1); ctx.set_option(k, int(v))
    tc = ops.TaskContext(ctx, batch_size=8192)
    tensors = tpch.gen_device_tensors(args.sf)
    want = tpch.q3_checksum_torch(tensors)
    g = torch.Generator(device="cuda"); g.manual_seed(20260311)
    t2 = {}
    for prefix in ("c_", "o_", "l_"):
        cols = [k for k in tensors if k.startswith(prefix)]
        perm = torch.randperm(tensors[cols[0]].shape[0], generator=g, device="cuda")
        for k in cols:
            t2[k] = tensors[k][perm] if tensors[k].dim() == 1 else torch.stack([tensors[k][:, h][perm] for h in range(tensors[k].shape[1])], dim=1).contiguous()
        del perm
    del tensors
    tables = tpch.tables_from_torch(ctx, t2)
    template = tpch.q3_plan(tables, batch_size=8192)
    out = None

    def step():
        nonlocal out
        out = [b for b in ops.with_fresh_state(template).execute(0, tc)]
        ctx.synchronize()
    for _ in range(args.warmup):
        step()
    ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read(); step(); prof = ctx.profile_read(); ctx.profile_enable(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / args.steps * 1e3
    got = tpch.q3_checksum_result(out or [])
    assert got == want, (got, want)
    kern = {k: round(v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1]) if not k.startswith("sync:")}
    syncs = {k[5:]: v[0] for k, v in prof.items() if k.startswith("sync:")}
    print(json.dumps({"workload": "q3_shuffled_inputs", "sf": args.sf, "ms_per_step": round(ms, 3), "steps": args.steps, "result_check": "equals the torch recomputation over the unshuffled tensors",
                      "kernel_ms_per_step": dict(list(kern.items())[:14]), "host_syncs_by_cause": syncs, "options": args.option}))


if __name__ == "__main__":
    main()
