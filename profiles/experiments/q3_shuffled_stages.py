"""Q3 over row-wise permuted tables, stage by stage: which operator raises / returns the wrong number of rows.  python profiles/experiments/q3_shuffled_stages.py [sf]"""
import decimal, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pyarrow as pa
import torch
import dfgpu
from dfgpu import capi, tpch, physical_plan as ops
from dfgpu.tpch import CUSTOMER_SCHEMA, ORDERS_SCHEMA, LINEITEM_SCHEMA, Q3_SEGMENT, Q3_DATE, _schema

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
shuffle = (sys.argv[2] if len(sys.argv) > 2 else "col").split(",")
ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
tensors = tpch.gen_device_tensors(sf, tpch.SEED, 0, 1, "cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
t2 = {}
for prefix in ("c_", "o_", "l_"):
    cols = [k for k in tensors if k.startswith(prefix)]
    if prefix[0] in shuffle:
        perm = torch.randperm(tensors[cols[0]].shape[0], generator=g, device="cuda")
        for k in cols:
            t2[k] = tensors[k][perm] if tensors[k].dim() == 1 else torch.stack([tensors[k][:, h][perm] for h in range(tensors[k].shape[1])], dim=1).contiguous() if tensors[k].dim() == 1 else torch.stack([tensors[k][:, h][perm] for h in range(tensors[k].shape[1])], dim=1).contiguous()
    else:
        for k in cols:
            t2[k] = tensors[k]
tables = tpch.tables_from_torch(ctx, t2)
tc = ops.TaskContext(ctx, 8192)
C, L, B = ops.Column, ops.Literal, ops.BinaryExpr
cust = ops.MemoryExec([[tables["customer"]]], _schema(CUSTOMER_SCHEMA)); orders = ops.MemoryExec([[tables["orders"]]], _schema(ORDERS_SCHEMA)); line = ops.MemoryExec([[tables["lineitem"]]], _schema(LINEITEM_SCHEMA))
cb = lambda p: ops.CoalesceBatchesExec(p, 8192)
f_c = cb(ops.FilterExec(B(C("c_mktsegment", 1), "=", L(Q3_SEGMENT, pa.utf8())), cust))
p_c = ops.ProjectionExec([(C("c_custkey", 0), "c_custkey")], f_c)
f_o = cb(ops.FilterExec(B(C("o_orderdate", 2), "<", L(Q3_DATE, pa.date32())), orders))
j1 = cb(ops.HashJoinExec(p_c, f_o, [(C("c_custkey", 0), C("o_custkey", 1))], None, "Inner", "Partitioned"))
p_j1 = ops.ProjectionExec([(C("o_orderkey", 1), "o_orderkey"), (C("o_orderdate", 3), "o_orderdate"), (C("o_shippriority", 4), "o_shippriority")], j1)
f_l = cb(ops.FilterExec(B(C("l_shipdate", 3), ">", L(Q3_DATE, pa.date32())), line))
p_l = ops.ProjectionExec([(C("l_orderkey", 0), "l_orderkey"), (C("l_extendedprice", 1), "l_extendedprice"), (C("l_discount", 2), "l_discount")], f_l)
j2 = cb(ops.HashJoinExec(p_j1, p_l, [(C("o_orderkey", 0), C("l_orderkey", 0))], None, "Inner", "Partitioned"))
p_j2 = ops.ProjectionExec([(C("o_orderdate", 1), "o_orderdate"), (C("o_shippriority", 2), "o_shippriority"), (C("l_orderkey", 3), "l_orderkey"), (C("l_extendedprice", 4), "l_extendedprice"), (C("l_discount", 5), "l_discount")], j2)
revenue = B(C("l_extendedprice", 3), "*", B(L(decimal.Decimal(1), pa.decimal128(20, 0)), "-", C("l_discount", 4)))
p_rev = ops.ProjectionExec([(C("l_orderkey", 2), "l_orderkey"), (revenue, "rev")], p_j2)
agg = ops.AggregateExec("Single", [(C("l_orderkey", 2), "l_orderkey"), (C("o_orderdate", 0), "o_orderdate"), (C("o_shippriority", 1), "o_shippriority")],
                        [ops.AggregateFunctionExpr("SUM", revenue, "s", input_field=ops.Field("rev", capi.DECIMAL128, 38, 4))], p_j2)
# expectations from torch over the unshuffled tensors
seg = tpch.SEGMENTS.index(Q3_SEGMENT); tt = tensors
cust_ok = tt["c_mktsegment"] == seg; o_sel = (tt["o_orderdate"] < Q3_DATE) & cust_ok[tt["o_custkey"] - 1]
k = tt["l_orderkey"] - 1; oidx = (k // 32) * 8 + (k % 32); l_sel = (tt["l_shipdate"] > Q3_DATE) & o_sel[oidx]
print("expected rows: j1", int(o_sel.sum()), "j2", int(l_sel.sum()), flush=True)
for name, plan in [("f_c", f_c), ("f_o", f_o), ("j1", j1), ("f_l", f_l), ("j2", j2), ("p_j2", p_j2), ("p_rev", p_rev), ("agg", agg)]:
    try:
        rows = 0; checks = 0
        for b in ops.with_fresh_state(plan).execute(0, tc):
            b.columns; rows += b.num_rows
        ctx.synchronize()
        print(name, "rows", rows, flush=True)
    except Exception as e:  # noqa
        print(name, "RAISED", str(e)[:200], flush=True)
import numpy as np
bs = list(ops.with_fresh_state(p_j2).execute(0, tc))
for ci, nm in ((2, "l_orderkey"), (3, "l_extendedprice"), (4, "l_discount"), (0, "o_orderdate")):
    a = pa.concat_arrays([b.columns[ci].to_arrow() for b in bs])
    if pa.types.is_decimal(a.type):
        v = np.frombuffer(a.buffers()[1], dtype=np.int64).reshape(-1, 2)
        print(nm, a.type, "lo min/max", v[:, 0].min(), v[:, 0].max(), "hi min/max", v[:, 1].min(), v[:, 1].max(), "sum lo", int(v[:, 0].sum()), flush=True)
    else:
        v = a.to_numpy(zero_copy_only=False); print(nm, a.type, v.min(), v.max(), int(v.astype(np.int64).sum()), flush=True)
print("torch: price sum", int(tt["l_extendedprice"][:, 0][l_sel].sum()), "disc sum", int(tt["l_discount"][:, 0][l_sel].sum()), "okey sum", int(tt["l_orderkey"][l_sel].sum()), "price max", int(tt["l_extendedprice"][:, 0].max()), "hi max", int(tt["l_extendedprice"][:, 1].max()))
print("shuffled tensor l_extendedprice contiguous", t2["l_extendedprice"].is_contiguous(), t2["l_extendedprice"].shape, t2["l_extendedprice"].stride(), "hi max", int(t2["l_extendedprice"][:, 1].max()))
