"""Diagnostic: device memory before / after many executions of the Q3 plan and of the fused Q1 aggregate (free bytes must level off)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dfgpu
from dfgpu import tpch, physical_plan as ops
torch.cuda.set_device(0)
ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
tc = ops.TaskContext(ctx, batch_size=8192)
tables = tpch.gen_device(ctx, 10.0)
template = tpch.q3_plan(tables, batch_size=8192)
def step():
    out = [b for b in ops.with_fresh_state(template).execute(0, tc)]
    ctx.synchronize()
    return sum(b.num_rows for b in out)
for phase in range(4):
    for _ in range(100):
        rows = step()
    free, total = torch.cuda.mem_get_info()
    print(f"after {100 * (phase + 1)} Q3 steps: free {free / 2**20:.0f} MiB of {total / 2**20:.0f}, result rows {rows}", flush=True)
