"""Experiment: why does AggregateExec leave the pre-aggregation for the 3-key Decimal128 workload?  Prints the verdict-only and the real call's status text."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch, dfgpu
from dfgpu import capi
n, total = 8_000_000, 80_000
torch.cuda.set_device(0); ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda"); g.manual_seed(1)
gid = torch.randint(0, total, (n,), generator=g, device="cuda", dtype=torch.int64)
k0 = gid * 7919; k1 = (8035 + gid % 2400).to(torch.int32); k2 = (gid % 3).to(torch.int32)
val = torch.zeros((n, 2), dtype=torch.int64, device="cuda"); val[:, 0] = torch.randint(90000, 10494951, (n,), generator=g, device="cuda")
keys = [ctx.wrap_tensor(k0, capi.INT64), ctx.wrap_tensor(k1, capi.DATE32), ctx.wrap_tensor(k2, capi.INT32)]
v = ctx.wrap_tensor(val, capi.DECIMAL128, 15, 2)
for label, kinds, vals in (("SUM decimal + COUNT", [0, 2], [v, None]), ("COUNT only", [2], [None]), ("SUM decimal only", [0], [v])):
    try:
        pk, st = dfgpu.agg_preaggregate(ctx, keys, kinds, vals)
        print(label, "-> taken:", len(pk[0]), "partial rows")
    except capi.DfgpuError as e:
        print(label, "-> declined:", e)
