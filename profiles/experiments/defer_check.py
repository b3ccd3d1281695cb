"""Diagnostic: is the deferred flag region active inside dfgpu_stream_next?  Times tiny takes with / without a region."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pyarrow as pa, torch
import dfgpu
from dfgpu import physical_plan as ops
ctx = dfgpu.Context(0)
a = ctx.from_arrow(pa.array(np.arange(1000, dtype=np.int64)))
idx = ctx.from_arrow(pa.array(np.arange(0, 1000, 3, dtype=np.uint32)))
def bench(tag):
    ctx.synchronize(); t = time.perf_counter()
    for _ in range(2000): ctx.take(a, idx)
    ctx.synchronize(); print(tag, (time.perf_counter() - t) / 2000 * 1e6, "us per take")
bench("no region")
ctx.set_option("defer_flag_checks", 1); bench("region"); ctx.set_option("defer_flag_checks", 0)
# through a plan: projection of 40 expressions over one batch
t = pa.table({"x": pa.array(np.arange(100000, dtype=np.int64))})
b = ops.batch_from_arrow(ctx, t)
C = ops.Column
exprs = [(ops.BinaryExpr(C("x", 0), "/", ops.Literal(3 + i, pa.int64())), f"e{i}") for i in range(40)]
plan = ops.ProjectionExec(exprs, ops.MemoryExec([[b]], b.schema))
tc = ops.TaskContext(ctx, 8192)
for rep in range(3):
    ctx.synchronize(); t0 = time.perf_counter()
    out = list(plan.execute(0, tc)); ctx.synchronize()
    print("plan with 40 checked divisions:", (time.perf_counter() - t0) * 1e6, "us")
