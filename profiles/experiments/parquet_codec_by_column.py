"""Which column's pages bound the page decompression kernel (codec = argv[2], default zstd): one read per column of the bench file, kernel times from the ctx profiler."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pyarrow as pa, pyarrow.parquet as pq, torch
import dfgpu
from dfgpu.parquet import ParquetFile

nr = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
rng = np.random.default_rng(11)
def dec(lo, hi):
    v = rng.integers(lo, hi, nr).astype(np.int64); buf = np.empty((nr, 2), dtype=np.int64); buf[:, 0] = v; buf[:, 1] = v >> 63
    return pa.Array.from_buffers(pa.decimal128(15, 2), nr, [None, pa.py_buffer(buf.tobytes())])
pick = lambda words: pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, len(words), nr).astype(np.int32)), pa.array(words)).cast(pa.string())
table = pa.table({"l_orderkey": pa.array(np.sort(rng.integers(0, nr // 4 * 32, nr)).astype(np.int64)), "l_quantity": dec(100, 5001), "l_extendedprice": dec(90000, 10494951),
                  "l_discount": dec(0, 11), "l_shipdate": pa.array(rng.integers(8035, 10560, nr).astype(np.int32), type=pa.date32()),
                  "l_returnflag": pick(["A", "N", "R"]), "l_shipmode": pick(["AIR", "FOB", "MAIL", "RAIL", "REG AIR", "SHIP", "TRUCK"])})
path = os.path.join(tempfile.gettempdir(), "codec_by_col.parquet")
pq.write_table(table, path, row_group_size=1 << 20, compression=(sys.argv[2] if len(sys.argv) > 2 else "zstd"))
md = pq.ParquetFile(path).metadata
ctx = dfgpu.Context(0)
f = ParquetFile(ctx, path=path, stage_on_device=True)
for c, name in enumerate(f.column_names()):
    f.read(columns=[c]); ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_read(); f.read(columns=[c]); ctx.synchronize(); p = ctx.profile_read(); ctx.profile_enable(False)
    cc = md.row_group(0).column(c)
    print(name, {k: round(v[1], 3) for k, v in p.items() if not k.startswith("sync")}, "rg0: comp", cc.total_compressed_size, "uncomp", cc.total_uncompressed_size, cc.encodings, flush=True)
