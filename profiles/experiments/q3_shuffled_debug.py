"""Q3 over row-wise permuted tables at a small scale factor, option by option: which path raises / differs.  python profiles/experiments/q3_shuffled_debug.py [sf]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dfgpu
from dfgpu import tpch, physical_plan as ops

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
tensors = tpch.gen_device_tensors(sf, tpch.SEED, 0, 1, "cuda")
want = tpch.q3_checksum_torch(tensors)
g = torch.Generator(device="cuda"); g.manual_seed(1)
t2 = {}
for prefix in ("c_", "o_", "l_"):
    cols = [k for k in tensors if k.startswith(prefix)]
    perm = torch.randperm(tensors[cols[0]].shape[0], generator=g, device="cuda")
    for k in cols:
        t2[k] = tensors[k][perm] if tensors[k].dim() == 1 else torch.stack([tensors[k][:, h][perm] for h in range(tensors[k].shape[1])], dim=1).contiguous()
tc = ops.TaskContext(ctx, 8192)
for name, opts in [("default", {}), ("no preaggregate", {"agg_partitioned": 0}), ("no lazy build rows", {"join_lazy_build_rows": 0}), ("no unsorted rank", {"join_rank_index_unsorted": 0}),
                   ("no fused aggregate", {"fused_aggregate_min_rows": -1}), ("no partitioned join", {"join_partitioned": 0})]:
    saved = {k: ctx.get_option(k) for k in opts}
    for k, v in opts.items():
        ctx.set_option(k, v)
    try:
        plan = tpch.q3_plan(tpch.tables_from_torch(ctx, t2), batch_size=8192)
        out = [b for b in plan.execute(0, tc)]
        ctx.synchronize()
        got = tpch.q3_checksum_result(out)
        print(name, "ok" if got == want else f"DIFFERS {got} vs {want}", flush=True)
    except Exception as e:  # noqa
        print(name, "RAISED", str(e)[:300], flush=True)
    finally:
        for k, v in saved.items():
            ctx.set_option(k, v)
