"""Diagnostic: cProfile of N>1 bench steps (colocated plan) with world_size 1 on RCCL at the per-rank scale of an 8-GPU SF100 run."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
import dfgpu
from dfgpu import exchange, tpch, physical_plan as ops
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 12.5
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29656")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
tc = ops.TaskContext(ctx, batch_size=8192)
tables = tpch.gen_device(ctx, sf)
torch.cuda.synchronize()
NAMES = ["l_orderkey", "revenue", "o_orderdate", "o_shippriority"]
staged = tpch.Q3ColocatedStaged(tables, batch_size=8192)
def step():
    plan = staged
    local = [b for b in plan.execute(0, tc)]
    mine = ops.concat_batches(local[0].schema, local)
    g = exchange.gather_batches(ctx, None, mine, 0, names=NAMES)
    C = ops.Column
    final = ops.SortExec([ops.PhysicalSortExpr(C("revenue", 1), True, True), ops.PhysicalSortExpr(C("o_orderdate", 2), False, False)], ops.MemoryExec([[g]], g.schema))
    out = [b for b in final.execute(0, tc)]
    ctx.synchronize()
for _ in range(3): step()
t0 = time.perf_counter()
for _ in range(10): step()
print(f"colocated step at sf={sf}: {(time.perf_counter() - t0) * 100:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
dist.destroy_process_group()
