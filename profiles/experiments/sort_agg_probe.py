"""Diagnostic: what a sort-based accumulate of a high-cardinality group-by would cost with today's kernels (100 M rows):
sort_to_indices over the u32 group ids, gather of ids and values through the order, accumulate over the now clustered ids --
against the direct atomics path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dfgpu
from dfgpu import capi
torch.cuda.set_device(0)
ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
n = 100_000_000
for total in (1_000_000, 20_000_000):
    g = torch.randint(0, total, (n,), device="cuda", dtype=torch.int32)
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    gd, xd = ctx.wrap_tensor(g, capi.UINT32), ctx.wrap_tensor(x, capi.FLOAT64)
    def timed(f, reps=3):
        f(); ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): r = f()
        ctx.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
    t_sort, order = timed(lambda: ctx.sort_to_indices([gd], [False], [False]))
    t_tg, gs = timed(lambda: ctx.take(gd, order))
    t_tx, xs = timed(lambda: ctx.take(xd, order))
    def acc(ids, vals):
        a = dfgpu.GroupsAccumulator(ctx, capi.AGG_AVG, capi.FLOAT64)
        a.update_batch(vals, ids, None, total)
        return a
    t_direct, _ = timed(lambda: acc(gd, xd))
    t_sorted, _ = timed(lambda: acc(gs, xs))
    print(f"{total} groups: sort_to_indices {t_sort:.2f} ms | take ids {t_tg:.2f} | take values {t_tx:.2f} | AVG on clustered ids {t_sorted:.2f} | AVG direct {t_direct:.2f}", flush=True)
