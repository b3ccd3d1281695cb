"""Diagnostic: host-side cost of one N>1 bench step (plan construction, execution, gather) with world_size 1 on RCCL."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
import dfgpu
from dfgpu import exchange, tpch, physical_plan as ops
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
tc = ops.TaskContext(ctx, batch_size=8192)
tables = tpch.gen_device(ctx, sf)
torch.cuda.synchronize()
for name in ("q3_colocated_plan", "q3_broadcast_plan", "q3_distributed_plan"):
    for rep in range(5):
        ctx.synchronize(); t0 = time.perf_counter()
        plan = getattr(tpch, name)(tables, batch_size=8192)
        t1 = time.perf_counter()
        local = [b for b in plan.execute(0, tc)]
        t2 = time.perf_counter()
        mine = ops.concat_batches(local[0].schema, local)
        g = exchange.gather_batches(ctx, None, mine, 0, names=["l_orderkey", "revenue", "o_orderdate", "o_shippriority"])
        ctx.synchronize(); t3 = time.perf_counter()
    print(f"{name} sf={sf}: build {1e6*(t1-t0):.0f} us | execute {1e6*(t2-t1):.0f} us | gather {1e6*(t3-t2):.0f} us | total {1e6*(t3-t0):.0f} us")
import cProfile, pstats
plan = tpch.q3_colocated_plan(tables, batch_size=8192)
local = [b for b in plan.execute(0, tc)]
mine = ops.concat_batches(local[0].schema, local)
ctx.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    g = exchange.gather_batches(ctx, None, mine, 0, names=["l_orderkey", "revenue", "o_orderdate", "o_shippriority"])
ctx.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
dist.destroy_process_group()
