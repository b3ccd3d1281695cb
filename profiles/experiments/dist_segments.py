"""Diagnostic: where one N>1 bench step spends its time at the per-rank scale of an 8-GPU SF100 run (world_size 1 on RCCL):
segments separated by stream syncs (so their sum exceeds the unsegmented step), next to the single-process plan at the same scale."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
import dfgpu
from dfgpu import exchange, tpch, physical_plan as ops
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 12.5
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29657")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
tc = ops.TaskContext(ctx, batch_size=8192)
tables = tpch.gen_device(ctx, sf)
torch.cuda.synchronize()
NAMES = ["l_orderkey", "revenue", "o_orderdate", "o_shippriority"]
staged = tpch.Q3ColocatedStaged(tables, batch_size=8192)
C = ops.Column
keys = [ops.PhysicalSortExpr(C("revenue", 1), True, True), ops.PhysicalSortExpr(C("o_orderdate", 2), False, False)]
template = tpch.q3_plan(tables, batch_size=8192)
def single():
    out = [b for b in ops.with_fresh_state(template).execute(0, tc)]
    ctx.synchronize()
def step(seg=None):
    t = [time.perf_counter()]
    def mark():
        if seg is not None:
            ctx.synchronize(); t.append(time.perf_counter())
    staged.stage2.handle(tc)
    got = [b for b in staged.bcast.execute(0, tc)]
    mark()
    staged.slot.replace([got])
    local = [b for b in ops.with_fresh_state(staged.stage2).execute(0, tc)]
    mark()
    with ctx.deferred_flags():
        mine = ops.concat_batches(local[0].schema, local)
        g = exchange.gather_batches(ctx, None, mine, 0, names=NAMES)
    mark()
    final = ops.SortExec(keys, ops.MemoryExec([[g]], g.schema))
    out = [b for b in final.execute(0, tc)]
    ctx.synchronize(); t.append(time.perf_counter())
    if seg is not None:
        for i in range(len(t) - 1): seg[i] += t[i + 1] - t[i]
for _ in range(3): step(); single()
t0 = time.perf_counter()
for _ in range(20): single()
print(f"single-process plan at sf={sf}: {(time.perf_counter() - t0) * 50:.3f} ms")
t0 = time.perf_counter()
for _ in range(20): step()
print(f"N>1 step at sf={sf}: {(time.perf_counter() - t0) * 50:.3f} ms")
seg = [0.0] * 4
for _ in range(20): step(seg)
print("segments (ms, with a sync after each): stage1+broadcast %.3f | stage2 %.3f | concat+gather %.3f | final merge %.3f" % tuple(s * 50 for s in seg))
ctx.profile_enable(True); ctx.profile_read(); step(); p = ctx.profile_read(); ctx.profile_enable(False)
print("kernels:", {k: round(v[1], 3) for k, v in sorted(p.items(), key=lambda kv: -kv[1][1])[:14]})
print("launch count:", sum(v[0] for k, v in p.items() if not k.startswith("sync:")), "syncs:", {k: v[0] for k, v in p.items() if k.startswith("sync:")})
dist.destroy_process_group()
