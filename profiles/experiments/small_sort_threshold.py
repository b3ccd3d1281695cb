"""Q3 step at a per-rank scale factor with the packed-key sort taking smaller inputs: python profiles/experiments/small_sort_threshold.py [sf]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dfgpu
from dfgpu import tpch, physical_plan as ops
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 12.5
ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
tables = tpch.tables_from_torch(ctx, tpch.gen_device_tensors(sf, tpch.SEED, 0, 1, "cuda"))
template = tpch.q3_plan(tables, batch_size=8192); tc = ops.TaskContext(ctx, 8192)
def step():
    out = [b for b in ops.with_fresh_state(template).execute(0, tc)]; ctx.synchronize(); return out
for thr in (1 << 20, 1 << 17, 1 << 15, 1 << 12):
    ctx.set_option("sort_packed_min_rows", thr)
    for _ in range(3): step()
    ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read(); step(); pr = ctx.profile_read(); ctx.profile_enable(False)
    t0 = time.perf_counter()
    for _ in range(20): out = step()
    dt = (time.perf_counter() - t0) / 20 * 1e3
    srt = {k: round(v[1], 3) for k, v in pr.items() if "sort" in k or "radix" in k}
    print(f"sort_packed_min_rows {thr}: {dt:.3f} ms per step, result rows {sum(b.num_rows for b in out)}, sort kernels {srt}, syncs {sum(v[0] for k, v in pr.items() if k.startswith('sync:'))}", flush=True)
