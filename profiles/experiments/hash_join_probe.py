"""Diagnostic: join build / probe rates when the build keys are NOT sorted (no rank index): shuffled dense keys (bitmap prefilter +
hash table) and sparse random 64-bit keys (hash table only), 15 M build rows, 150 M probe rows, ~20 % of probes match."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dfgpu
from dfgpu import capi
torch.cuda.set_device(0)
ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
nb, npr = 15_000_000, 150_000_000
def timed(f, reps=3):
    f(); ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    ctx.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
for name in ("shuffled dense keys", "sparse random keys", "two int64 key columns"):
    if name == "shuffled dense keys":
        bk = torch.randperm(nb, device="cuda", dtype=torch.int64) * 5
        pk = torch.randint(0, nb * 25, (npr,), device="cuda", dtype=torch.int64)
        build, probe = [ctx.wrap_tensor(bk, capi.INT64)], [ctx.wrap_tensor(pk, capi.INT64)]
    elif name == "sparse random keys":
        bk = torch.randint(0, 2**62, (nb,), device="cuda", dtype=torch.int64)
        pk = torch.cat([bk[torch.randint(0, nb, (npr // 5,), device="cuda")], torch.randint(0, 2**62, (npr - npr // 5,), device="cuda", dtype=torch.int64)])
        pk = pk[torch.randperm(npr, device="cuda")]
        build, probe = [ctx.wrap_tensor(bk, capi.INT64)], [ctx.wrap_tensor(pk, capi.INT64)]
    else:
        b1 = torch.randint(0, 2**40, (nb,), device="cuda", dtype=torch.int64); b2 = torch.randint(0, 2**40, (nb,), device="cuda", dtype=torch.int64)
        sel = torch.randint(0, nb, (npr,), device="cuda")
        hit = torch.rand(npr, device="cuda") < 0.2
        p1 = torch.where(hit, b1[sel], torch.randint(0, 2**40, (npr,), device="cuda", dtype=torch.int64)); p2 = b2[sel]
        build, probe = [ctx.wrap_tensor(b1, capi.INT64), ctx.wrap_tensor(b2, capi.INT64)], [ctx.wrap_tensor(p1, capi.INT64), ctx.wrap_tensor(p2, capi.INT64)]
        del sel, hit
    torch.cuda.synchronize()
    t_build, table = timed(lambda: dfgpu.JoinTable(ctx, build))
    t_probe, (bi, pi) = timed(lambda: table.probe(probe))
    print(f"{name}: build {t_build:.2f} ms ({nb / t_build / 1e6:.1f} G rows/s) | probe {t_probe:.2f} ms ({npr / t_probe / 1e6:.1f} G rows/s), {len(pi)} matches", flush=True)
    del table, bi, pi, build, probe
    torch.cuda.empty_cache()
