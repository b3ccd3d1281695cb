"""Experiment: who keeps 35 GB of device arrays alive after the partition workload?  Reads the ctx's live_bytes around each stage.
  A  RepartitionExec outputs dropped right away             -> product-side leak if live stays up
  B  the same, outputs first touched through dev_tensor()   -> the bench's zero-copy torch view if live stays up only here
Run on a GPU box: python profiles/experiments/leak_partition.py [rows]"""
import gc, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import dfgpu
from dfgpu import capi, physical_plan as ops
import bench_workloads as bw

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
torch.cuda.set_device(0)
ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream); tc = ops.TaskContext(ctx, batch_size=8192)
live = lambda: round(ctx.get_option("live_bytes") / 1e9, 3)
k = torch.arange(n, dtype=torch.int64, device="cuda") * 7 + 1
v = torch.zeros((n, 2), dtype=torch.int64, device="cuda"); v[:, 0] = k
batch = ops.RecordBatch.from_arrays(ctx, ["k", "v"], [ctx.wrap_tensor(k, capi.INT64), ctx.wrap_tensor(v, capi.DECIMAL128, 15, 2)])
plan = ops.RepartitionExec(ops.MemoryExec([[batch]], batch.schema), ops.Partitioning.Hash([ops.Column("k", 0)], 8))
print("start live", live())

def run(touch):
    p2 = ops.with_fresh_state(plan); keep = []
    for d in range(8):
        for b in p2.execute(d, tc):
            b.columns; keep.append((d, b))
    ctx.synchronize()
    inside = live()
    if touch:
        s = 0
        for d, b in keep:
            if b.num_rows:
                t = bw.dev_tensor(torch, b.columns[0], "<i8"); s += int(t.sum().item()); del t
    del keep, p2
    return inside

for name, touch in (("A plain", False), ("B dev_tensor", True), ("A plain again", False)):
    inside = run(touch); after = live(); gc.collect(); after_gc = live()
    print(f"{name:16s} live inside step {inside}  after dropping refs {after}  after gc.collect {after_gc}")
del plan, batch; gc.collect(); print("after del plan/batch", live())
