"""Diagnostic: GroupValues::intern rate for unclustered Int64 keys (hash table path), 100 M rows."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dfgpu
from dfgpu import capi
torch.cuda.set_device(0)
ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
n = 100_000_000
for total in (1000, 1_000_000, 20_000_000):
    k = torch.randint(0, total, (n,), device="cuda", dtype=torch.int64) * 7919
    kd = ctx.wrap_tensor(k, capi.INT64)
    def run():
        gv = dfgpu.GroupValues(ctx, 1)
        ids = gv.intern([kd])
        return len(gv)
    run(); ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(3): g = run()
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
    ctx.profile_enable(True); ctx.profile_read(); run(); p = ctx.profile_read(); ctx.profile_enable(False)
    print(f"{total} groups: intern {dt:.2f} ms ({n / dt / 1e6:.1f} G rows/s), {g} groups; kernels {({k2: round(v[1], 2) for k2, v in p.items() if not k2.startswith('sync:')})}", flush=True)
