"""Diagnostic (not part of the product): time the SF100 lineitem probe launch in isolation, back to back, with the real tables."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dfgpu
from dfgpu import tpch, physical_plan as ops

ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
tables = tpch.gen_device(ctx, float(sys.argv[1]) if len(sys.argv) > 1 else 100.0)
torch.cuda.synchronize()
li, od = tables["lineitem"], tables["orders"]
names = [f.name for f in li.schema.fields]
lk = li.columns[names.index("l_orderkey")]
ls = li.columns[names.index("l_shipdate")]
onames = [f.name for f in od.schema.fields]
ok = od.columns[onames.index("o_orderkey")]
odate = od.columns[onames.index("o_orderdate")]
import pyarrow as pa
omask = ctx.binary(12, odate, ctx.from_arrow(pa.array([tpch.Q3_DATE], type=pa.date32())), False, True)
lmask = ctx.binary(14, ls, ctx.from_arrow(pa.array([tpch.Q3_DATE], type=pa.date32())), False, True)
table = dfgpu.JoinTable(ctx, [ok], mask=omask)
ctx.profile_select("k_probe_match_bitmap")
ctx.profile_enable(True)
for rep in range(3):
    ctx.profile_read()
    for _ in range(5):
        bi, pi = table.probe([lk], mask=lmask)
    print("masked probe x5:", ctx.profile_read(), "matches", len(pi))
    for _ in range(5):
        bi, pi = table.probe([lk])
    print("unmasked probe x5:", ctx.profile_read(), "matches", len(pi))
