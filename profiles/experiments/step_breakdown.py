"""Diagnostic: host-side cost split of one bench step (plan construction vs execution) at a small scale factor."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dfgpu
from dfgpu import tpch, physical_plan as ops
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
tc = ops.TaskContext(ctx, batch_size=8192)
tables = tpch.gen_device(ctx, sf)
torch.cuda.synchronize()
for rep in range(6):
    ctx.synchronize(); t0 = time.perf_counter()
    plan = tpch.q3_plan(tables, batch_size=8192)
    h = plan.handle(tc)
    t1 = time.perf_counter()
    out = [b for b in plan.execute(0, tc)]
    t2 = time.perf_counter()
    ctx.synchronize(); t3 = time.perf_counter()
    print(f"sf={sf} build {1e6*(t1-t0):.0f} us | execute (host returns) {1e6*(t2-t1):.0f} us | final sync {1e6*(t3-t2):.0f} us | total {1e6*(t3-t0):.0f} us")
