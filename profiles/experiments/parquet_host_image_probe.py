"""Where the time of a Parquet read from a host image goes: open (page-locking the mapping), read with the image in host memory, read with the image in HBM.
Run on the GPU box: python profiles/experiments/parquet_host_image_probe.py"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pyarrow as pa, pyarrow.parquet as pq, torch
import dfgpu
from dfgpu.parquet import ParquetFile

ctx = dfgpu.Context(0)
n = 12_000_000
rng = np.random.default_rng(1)
table = pa.table({"k": pa.array(rng.integers(0, 1 << 40, n)), "a": pa.array(rng.integers(0, 1 << 40, n)), "b": pa.array(rng.integers(0, 1 << 40, n)), "c": pa.array(rng.random(n)),
                  "d": pa.array(rng.random(n)), "e": pa.array(rng.integers(0, 1 << 30, n).astype(np.int32))})
path = os.path.join(tempfile.gettempdir(), "probe.parquet")
pq.write_table(table, path, row_group_size=1 << 20, compression="NONE", use_dictionary=False)
fb = os.path.getsize(path)
print("file MB", fb / 1e6)
t0 = time.perf_counter(); fh = ParquetFile(ctx, path=path, stage_on_device=False); print("open (host image) ms", (time.perf_counter() - t0) * 1e3)
for i in range(4):
    t0 = time.perf_counter(); cols = fh.read(); t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
    print(f"host image read {i}: enqueue {(t1 - t0) * 1e3:.2f} ms, total {(t2 - t0) * 1e3:.2f} ms -> {fb / (t2 - t0) / 1e9:.1f} GB/s of file bytes")
    del cols
fh.close()
t0 = time.perf_counter(); fd = ParquetFile(ctx, path=path, stage_on_device=True); print("open (device image) ms", (time.perf_counter() - t0) * 1e3)
for i in range(3):
    t0 = time.perf_counter(); cols = fd.read(); ctx.synchronize(); t2 = time.perf_counter()
    print(f"device image read {i}: {(t2 - t0) * 1e3:.2f} ms")
    del cols
fd.close()
# plain pinned H2D rate of this box
h = torch.empty(fb, dtype=torch.uint8).pin_memory(); d = torch.empty(fb, dtype=torch.uint8, device="cuda")
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"torch pinned H2D: {fb / dt / 1e9:.1f} GB/s")
os.unlink(path)
