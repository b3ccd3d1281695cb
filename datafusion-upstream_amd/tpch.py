"""Synthetic TPC-H-shaped columns and the reference's Q3 physical plan.

dbgen is not available, so columns follow the TPC-H value distributions with fixed seeds and the dtypes of
benchmarks/src/tpch/mod.rs:44-140: keys Int64, money Decimal128(15,2), dates Date32, o_shippriority Int32,
c_mktsegment Dictionary(Int8, Utf8) over the 5 TPC-H segments.

  customer : c_custkey 1..150000*SF dense; c_mktsegment uniform over 5 values
  orders   : o_orderkey sparse TPC-H pattern (8 of every 32 integers); o_custkey uniform over custkeys not
             divisible by 3; o_orderdate uniform in [1992-01-01, 1998-08-02] = days 8035..10440; o_shippriority 0
  lineitem : 1..7 lines per order (mean 4); l_extendedprice uniform [900.00, 104949.50]; l_discount 0.00..0.10;
             l_shipdate = o_orderdate + U[1,121]

`gen_host` (numpy, parity tests + CPU baseline sample) and `gen_device` (torch on the GPU, SF100 without staging
30 GB through the host) draw from different generators but the same distributions; each is deterministic per seed.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

from . import capi
from . import physical_plan as ops
from .device import Array, Context

SEGMENTS = ["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"]
Q3_SEGMENT = "BUILDING"
Q3_DATE = 9204                      # DATE '1995-03-15' (tpch/q3.slt.part: Date32("9204"))
DATE_LO, DATE_HI = 8035, 10440
SEED = 20241024

CUSTOMER_PER_SF, ORDERS_PER_SF = 150_000, 1_500_000


def table_rows(sf: float):
    return int(CUSTOMER_PER_SF * sf), int(ORDERS_PER_SF * sf)


def orderkey_of(i):
    """sparse TPC-H o_orderkey: 8 keys out of every 32 integers"""
    return (i // 8) * 32 + (i % 8) + 1


def gen_host(sf: float, seed: int = SEED) -> Dict[str, np.ndarray]:
    """Host (numpy) tables; Decimal128 columns as (n, 2) uint64 little-endian [lo, hi] unscaled values."""
    nc, no = table_rows(sf)
    r = np.random.default_rng(seed)
    t: Dict[str, np.ndarray] = {}
    t["c_custkey"] = np.arange(1, nc + 1, dtype=np.int64)
    t["c_mktsegment"] = r.integers(0, 5, nc).astype(np.int8)
    i = np.arange(no, dtype=np.int64)
    t["o_orderkey"] = orderkey_of(i)
    valid_cust = nc - nc // 3
    j = r.integers(0, valid_cust, no).astype(np.int64)          # j-th custkey not divisible by 3
    t["o_custkey"] = j + j // 2 + 1
    t["o_orderdate"] = r.integers(DATE_LO, DATE_HI + 1, no).astype(np.int32)
    t["o_shippriority"] = np.zeros(no, dtype=np.int32)
    lines = r.integers(1, 8, no)
    t["l_orderkey"] = np.repeat(t["o_orderkey"], lines)
    nl = len(t["l_orderkey"])
    ext = r.integers(90000, 10494951, nl).astype(np.int64)
    disc = r.integers(0, 11, nl).astype(np.int64)
    t["l_extendedprice"] = np.stack([ext.view(np.uint64), np.zeros(nl, np.uint64)], axis=1)
    t["l_discount"] = np.stack([disc.view(np.uint64), np.zeros(nl, np.uint64)], axis=1)
    t["l_shipdate"] = (np.repeat(t["o_orderdate"], lines) + r.integers(1, 122, nl)).astype(np.int32)
    return t


def total_input_rows(t) -> int:
    return len(t["c_custkey"]) + len(t["o_orderkey"]) + len(t["l_orderkey"])


# ----------------------------------------------------------------------------- device tables
def _segment_dictionary(ctx: Context) -> Array:
    import pyarrow as pa
    return ctx.from_arrow(pa.array(SEGMENTS, type=pa.utf8()))


def _wrap_dictionary(ctx: Context, keys_tensor, dictionary: Array) -> Array:
    d = capi.ArrayDesc()
    dd = dictionary.describe()
    d.type, d.key_type, d.length, d.null_count = capi.DICTIONARY, capi.INT8, keys_tensor.numel(), 0
    d.values = keys_tensor.data_ptr()
    import ctypes as C
    d.dictionary = C.pointer(dd)
    a = ctx.wrap_device(d, keepalive=(keys_tensor, dictionary, dd))
    return a


def _schema(names_types) -> ops.Schema:
    return ops.Schema([ops.Field(n, t, p, s) for n, t, p, s in names_types])


CUSTOMER_SCHEMA = [("c_custkey", capi.INT64, 0, 0), ("c_mktsegment", capi.UTF8, 0, 0)]
ORDERS_SCHEMA = [("o_orderkey", capi.INT64, 0, 0), ("o_custkey", capi.INT64, 0, 0), ("o_orderdate", capi.DATE32, 0, 0), ("o_shippriority", capi.INT32, 0, 0)]
LINEITEM_SCHEMA = [("l_orderkey", capi.INT64, 0, 0), ("l_extendedprice", capi.DECIMAL128, 15, 2), ("l_discount", capi.DECIMAL128, 15, 2), ("l_shipdate", capi.DATE32, 0, 0)]


def tables_from_torch(ctx: Context, tt: dict) -> Dict[str, ops.RecordBatch]:
    """tt: torch CUDA tensors keyed by column name (decimals as (n,2) int64 [lo,hi])."""
    import torch
    torch.cuda.synchronize()       # the tensors were produced on torch's stream; the ctx may run on another one
    segd = _segment_dictionary(ctx)
    cust = [ctx.wrap_tensor(tt["c_custkey"], capi.INT64), _wrap_dictionary(ctx, tt["c_mktsegment"], segd)]
    orders = [ctx.wrap_tensor(tt["o_orderkey"], capi.INT64), ctx.wrap_tensor(tt["o_custkey"], capi.INT64),
              ctx.wrap_tensor(tt["o_orderdate"], capi.DATE32), ctx.wrap_tensor(tt["o_shippriority"], capi.INT32)]
    line = [ctx.wrap_tensor(tt["l_orderkey"], capi.INT64), ctx.wrap_tensor(tt["l_extendedprice"], capi.DECIMAL128, 15, 2),
            ctx.wrap_tensor(tt["l_discount"], capi.DECIMAL128, 15, 2), ctx.wrap_tensor(tt["l_shipdate"], capi.DATE32)]
    mk = lambda sch, cols: ops.RecordBatch.from_arrays(ctx, [n for n, _, _, _ in sch], cols)
    return {"customer": mk(CUSTOMER_SCHEMA, cust), "orders": mk(ORDERS_SCHEMA, orders), "lineitem": mk(LINEITEM_SCHEMA, line)}


def upload(ctx: Context, host: Dict[str, np.ndarray], device: str = "cuda") -> Dict[str, ops.RecordBatch]:
    """numpy tables -> HBM (torch owns the memory, dfgpu wraps it zero-copy)."""
    import torch
    tt = {k: torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v).to(device) for k, v in host.items()}
    return tables_from_torch(ctx, tt)


def gen_device(ctx: Context, sf: float, seed: int = SEED, rank: int = 0, world: int = 1, device: str = "cuda") -> Dict[str, ops.RecordBatch]:
    """Generate rank's 1/world shard of the tables directly in HBM (contiguous ranges of customers / orders;
    lineitems stay with their order, as TPC-H files are clustered)."""
    return tables_from_torch(ctx, gen_device_tensors(sf, seed, rank, world, device))


def gen_device_tensors(sf: float, seed: int = SEED, rank: int = 0, world: int = 1, device: str = "cuda") -> dict:
    """The torch tensors behind gen_device (the same generator calls in the same order: identical values for identical arguments)."""
    import torch
    nc, no = table_rows(sf)
    g = torch.Generator(device=device)
    g.manual_seed(seed + 7919 * rank)
    c_lo, c_hi = nc * rank // world, nc * (rank + 1) // world
    o_lo, o_hi = no * rank // world, no * (rank + 1) // world
    tt = {}
    tt["c_custkey"] = torch.arange(c_lo + 1, c_hi + 1, dtype=torch.int64, device=device)
    tt["c_mktsegment"] = torch.randint(0, 5, (c_hi - c_lo,), generator=g, device=device, dtype=torch.int8)
    i = torch.arange(o_lo, o_hi, dtype=torch.int64, device=device)
    tt["o_orderkey"] = (i // 8) * 32 + (i % 8) + 1
    valid_cust = nc - nc // 3
    j = torch.randint(0, valid_cust, (o_hi - o_lo,), generator=g, device=device, dtype=torch.int64)
    tt["o_custkey"] = j + j // 2 + 1
    tt["o_orderdate"] = torch.randint(DATE_LO, DATE_HI + 1, (o_hi - o_lo,), generator=g, device=device, dtype=torch.int32)
    tt["o_shippriority"] = torch.zeros(o_hi - o_lo, dtype=torch.int32, device=device)
    lines = torch.randint(1, 8, (o_hi - o_lo,), generator=g, device=device, dtype=torch.int64)
    tt["l_orderkey"] = torch.repeat_interleave(tt["o_orderkey"], lines)
    nl = tt["l_orderkey"].numel()
    tt["l_shipdate"] = (torch.repeat_interleave(tt["o_orderdate"], lines) + torch.randint(1, 122, (nl,), generator=g, device=device, dtype=torch.int32)).contiguous()
    del lines, i, j
    for name, lo, hi in (("l_extendedprice", 90000, 10494951), ("l_discount", 0, 11)):
        v = torch.zeros((nl, 2), dtype=torch.int64, device=device)          # [lo, hi] words; values are non-negative => hi = 0
        v[:, 0] = torch.randint(lo, hi, (nl,), generator=g, device=device, dtype=torch.int64)
        tt[name] = v
    return tt


def q3_checksum_torch(tt: dict) -> dict:
    """TPC-H Q3 over the tensors of gen_device_tensors(rank 0 of 1) recomputed with plain torch ops (no dfgpu code): number of result groups,
    wrapping Int64 sums of their l_orderkey and of their revenue (unscaled Decimal128(38,4) = price * (100 - discount), exact in Int64 here).
    Used by bench.py to check the SF100 result outside the timed region."""
    import torch
    seg = SEGMENTS.index(Q3_SEGMENT)
    cust_ok = tt["c_mktsegment"] == seg                                              # index = c_custkey - 1 (dense keys)
    o_sel = (tt["o_orderdate"] < Q3_DATE) & cust_ok[tt["o_custkey"] - 1]
    k = tt["l_orderkey"] - 1
    oidx = (k // 32) * 8 + (k % 32)                                                   # inverse of orderkey_of
    del k
    l_sel = (tt["l_shipdate"] > Q3_DATE) & o_sel[oidx]
    rev = tt["l_extendedprice"][:, 0] * (100 - tt["l_discount"][:, 0])
    sums = torch.zeros(tt["o_orderkey"].numel(), dtype=torch.int64, device=rev.device)
    sums.index_add_(0, oidx[l_sel], rev[l_sel])
    grp = sums > 0
    return {"groups": int(grp.sum().item()), "sum_orderkey": int(tt["o_orderkey"][grp].sum().item()), "sum_revenue": int(sums.sum().item())}


def q3_checksum_result(batches: List["ops.RecordBatch"]) -> dict:
    """The same three numbers from a Q3 result (columns l_orderkey, revenue Decimal128(38,4), o_orderdate, o_shippriority)."""
    r = q3_result_to_numpy(batches)
    rev = r["revenue"]
    lo = rev[:, 0].view(np.int64) if rev.ndim == 2 else rev.astype(np.int64)
    return {"groups": int(len(r["l_orderkey"])), "sum_orderkey": int(r["l_orderkey"].astype(np.int64).sum()), "sum_revenue": int(lo.sum())}


def tables_to_host(tables: Dict[str, "ops.RecordBatch"]) -> Dict[str, np.ndarray]:
    """Device tables -> the numpy layout of gen_host (Decimal128 as (n,2) uint64 [lo,hi], c_mktsegment as Int8 codes)."""
    import pyarrow as pa
    host = {}
    for batch in tables.values():
        for name, col in zip(batch.schema.names(), batch.columns):
            a = col.to_arrow()
            if pa.types.is_dictionary(a.type):
                host[name] = np.asarray(a.indices).astype(np.int8)
            elif pa.types.is_decimal(a.type):
                host[name] = np.frombuffer(a.buffers()[1], dtype=np.uint64, count=2 * len(a), offset=a.offset * 16).reshape(-1, 2).copy()
            elif pa.types.is_date32(a.type):
                host[name] = np.asarray(a.cast(pa.int32()))
            else:
                host[name] = np.asarray(a)
    return host


# ----------------------------------------------------------------------------- Q3 physical plan
def q3_plan(tables: Dict[str, ops.RecordBatch], batch_size: int = 8192) -> ops.ExecutionPlan:
    """The reference's physical plan for TPC-H Q3 (sqllogictest/test_files/tpch/q3.slt.part, benchmarks/queries/q3.sql)
    on ONE partition: with target_partitions = 1 EnforceDistribution adds no RepartitionExec and
    CombinePartialFinalAggregate folds Partial+Final into AggregateExec(mode=Single)."""
    import pyarrow as pa
    C, L, B = ops.Column, ops.Literal, ops.BinaryExpr
    cust = ops.MemoryExec([[tables["customer"]]], _schema(CUSTOMER_SCHEMA))
    orders = ops.MemoryExec([[tables["orders"]]], _schema(ORDERS_SCHEMA))
    line = ops.MemoryExec([[tables["lineitem"]]], _schema(LINEITEM_SCHEMA))

    cb = lambda p: ops.CoalesceBatchesExec(p, batch_size)
    f_c = cb(ops.FilterExec(B(C("c_mktsegment", 1), "=", L(Q3_SEGMENT, pa.utf8())), cust))
    p_c = ops.ProjectionExec([(C("c_custkey", 0), "c_custkey")], f_c)
    f_o = cb(ops.FilterExec(B(C("o_orderdate", 2), "<", L(Q3_DATE, pa.date32())), orders))
    j1 = cb(ops.HashJoinExec(p_c, f_o, [(C("c_custkey", 0), C("o_custkey", 1))], None, "Inner", "Partitioned"))
    # join schema: c_custkey, o_orderkey, o_custkey, o_orderdate, o_shippriority
    p_j1 = ops.ProjectionExec([(C("o_orderkey", 1), "o_orderkey"), (C("o_orderdate", 3), "o_orderdate"), (C("o_shippriority", 4), "o_shippriority")], j1)
    f_l = cb(ops.FilterExec(B(C("l_shipdate", 3), ">", L(Q3_DATE, pa.date32())), line))
    p_l = ops.ProjectionExec([(C("l_orderkey", 0), "l_orderkey"), (C("l_extendedprice", 1), "l_extendedprice"), (C("l_discount", 2), "l_discount")], f_l)
    j2 = cb(ops.HashJoinExec(p_j1, p_l, [(C("o_orderkey", 0), C("l_orderkey", 0))], None, "Inner", "Partitioned"))
    # join schema: o_orderkey, o_orderdate, o_shippriority, l_orderkey, l_extendedprice, l_discount
    p_j2 = ops.ProjectionExec([(C("o_orderdate", 1), "o_orderdate"), (C("o_shippriority", 2), "o_shippriority"), (C("l_orderkey", 3), "l_orderkey"),
                               (C("l_extendedprice", 4), "l_extendedprice"), (C("l_discount", 5), "l_discount")], j2)
    import decimal
    revenue = B(C("l_extendedprice", 3), "*", B(L(decimal.Decimal(1), pa.decimal128(20, 0)), "-", C("l_discount", 4)))
    agg = ops.AggregateExec("Single", [(C("l_orderkey", 2), "l_orderkey"), (C("o_orderdate", 0), "o_orderdate"), (C("o_shippriority", 1), "o_shippriority")],
                            [ops.AggregateFunctionExpr("SUM", revenue, "SUM(lineitem.l_extendedprice * Int64(1) - lineitem.l_discount)",
                                                       input_field=ops.Field("rev", capi.DECIMAL128, 38, 4))], p_j2)
    proj = ops.ProjectionExec([(C("l_orderkey", 0), "l_orderkey"), (C("revenue", 3), "revenue"), (C("o_orderdate", 1), "o_orderdate"), (C("o_shippriority", 2), "o_shippriority")], agg)
    return ops.SortExec([ops.PhysicalSortExpr(C("revenue", 1), descending=True, nulls_first=True),
                         ops.PhysicalSortExpr(C("o_orderdate", 2), descending=False, nulls_first=False)], proj)


def q3_result_to_numpy(batches: List[ops.RecordBatch]) -> Dict[str, np.ndarray]:
    """Device result -> numpy columns (revenue as (n,2) uint64 [lo,hi] unscaled Decimal128(38,4))."""
    import pyarrow as pa
    if not batches:
        return {"l_orderkey": np.zeros(0, np.int64), "revenue": np.zeros((0, 2), np.uint64), "o_orderdate": np.zeros(0, np.int32), "o_shippriority": np.zeros(0, np.int32)}
    cols = [pa.concat_arrays([b.columns[i].to_arrow() for b in batches]) for i in range(4)]
    rev = cols[1]
    assert rev.type == pa.decimal128(38, 4), rev.type
    raw = np.frombuffer(rev.buffers()[1], dtype=np.uint64, count=2 * len(rev), offset=rev.offset * 16).reshape(-1, 2).copy()
    return {"l_orderkey": np.asarray(cols[0]), "revenue": raw, "o_orderdate": np.asarray(cols[2].cast(pa.int32())), "o_shippriority": np.asarray(cols[3])}


def q3_broadcast_plan(tables: Dict[str, ops.RecordBatch], group=None, batch_size: int = 8192) -> ops.ExecutionPlan:
    """Q3 on N GPUs with both joins in PartitionMode::CollectLeft (JoinSelection picks it when the build side is under
    hash_join_single_partition_threshold, common/src/config.rs:562-566; with 288 GB of HBM per GPU that threshold is GBs):
    the filtered customers (SF100: 24 MB) and the customer-orders join result (234 MB) are all-gathered (BroadcastExec) and
    every rank probes its LOCAL orders / lineitem shard -- the 13 GB filtered lineitem never crosses xGMI.  Only the Partial
    aggregate state (36 MB) is shuffled.  Same rows as q3_distributed_plan / q3_plan."""
    import decimal
    import pyarrow as pa
    from .exchange import BroadcastExec, ShuffleExec
    C, L, B = ops.Column, ops.Literal, ops.BinaryExpr
    cust = ops.MemoryExec([[tables["customer"]]], _schema(CUSTOMER_SCHEMA))
    orders = ops.MemoryExec([[tables["orders"]]], _schema(ORDERS_SCHEMA))
    line = ops.MemoryExec([[tables["lineitem"]]], _schema(LINEITEM_SCHEMA))
    cb = lambda p: ops.CoalesceBatchesExec(p, batch_size)
    f_c = cb(ops.FilterExec(B(C("c_mktsegment", 1), "=", L(Q3_SEGMENT, pa.utf8())), cust))
    b_c = BroadcastExec(ops.ProjectionExec([(C("c_custkey", 0), "c_custkey")], f_c), group)
    f_o = cb(ops.FilterExec(B(C("o_orderdate", 2), "<", L(Q3_DATE, pa.date32())), orders))
    j1 = cb(ops.HashJoinExec(b_c, f_o, [(C("c_custkey", 0), C("o_custkey", 1))], None, "Inner", "CollectLeft"))
    p_j1 = ops.ProjectionExec([(C("o_orderkey", 1), "o_orderkey"), (C("o_orderdate", 3), "o_orderdate"), (C("o_shippriority", 4), "o_shippriority")], j1)
    b_j1 = BroadcastExec(p_j1, group)
    f_l = cb(ops.FilterExec(B(C("l_shipdate", 3), ">", L(Q3_DATE, pa.date32())), line))
    p_l = ops.ProjectionExec([(C("l_orderkey", 0), "l_orderkey"), (C("l_extendedprice", 1), "l_extendedprice"), (C("l_discount", 2), "l_discount")], f_l)
    j2 = cb(ops.HashJoinExec(b_j1, p_l, [(C("o_orderkey", 0), C("l_orderkey", 0))], None, "Inner", "CollectLeft"))
    p_j2 = ops.ProjectionExec([(C("o_orderdate", 1), "o_orderdate"), (C("o_shippriority", 2), "o_shippriority"), (C("l_orderkey", 3), "l_orderkey"),
                               (C("l_extendedprice", 4), "l_extendedprice"), (C("l_discount", 5), "l_discount")], j2)
    revenue = B(C("l_extendedprice", 3), "*", B(L(decimal.Decimal(1), pa.decimal128(20, 0)), "-", C("l_discount", 4)))
    gby = [(C("l_orderkey", 2), "l_orderkey"), (C("o_orderdate", 0), "o_orderdate"), (C("o_shippriority", 1), "o_shippriority")]
    aggr = [ops.AggregateFunctionExpr("SUM", revenue, "SUM(lineitem.l_extendedprice * Int64(1) - lineitem.l_discount)",
                                      input_field=ops.Field("rev", capi.DECIMAL128, 38, 4))]
    partial = ops.AggregateExec("Partial", gby, aggr, p_j2)
    s_a = cb(ShuffleExec(partial, [C("l_orderkey", 0), C("o_orderdate", 1), C("o_shippriority", 2)], group))
    final = ops.AggregateExec("FinalPartitioned", [(C("l_orderkey", 0), "l_orderkey"), (C("o_orderdate", 1), "o_orderdate"), (C("o_shippriority", 2), "o_shippriority")], aggr, s_a)
    proj = ops.ProjectionExec([(C("l_orderkey", 0), "l_orderkey"), (C("revenue", 3), "revenue"), (C("o_orderdate", 1), "o_orderdate"), (C("o_shippriority", 2), "o_shippriority")], final)
    return ops.SortExec([ops.PhysicalSortExpr(C("revenue", 1), descending=True, nulls_first=True),
                         ops.PhysicalSortExpr(C("o_orderdate", 2), descending=False, nulls_first=False)], proj, preserve_partitioning=True)


def q3_colocated_plan(tables: Dict[str, ops.RecordBatch], group=None, batch_size: int = 8192) -> ops.ExecutionPlan:
    """Q3 on N GPUs over CO-PARTITIONED shards: gen_device gives every rank a contiguous range of orders and the lineitems
    of exactly those orders (TPC-H files are clustered that way; dbgen -C/-S chunks have the same property), so equal
    o_orderkey / l_orderkey values already live on one rank.  The distribution requirement of the orders-lineitem HashJoinExec
    and of the GROUP BY l_orderkey, .. aggregation (hash_join.rs:520-527, aggregates/mod.rs:653-655: equal keys in one
    partition) is therefore met WITHOUT a RepartitionExec -- what EnforceDistribution does when its input already satisfies
    the requirement (core/src/physical_optimizer/enforce_distribution.rs).  Only the customer join has a real exchange step:
    customers are sharded by c_custkey, orders reference any customer, so the filtered customer keys (SF100: 24 MB) are
    all-gathered (PartitionMode::CollectLeft).  Every rank then runs join -> join -> AggregateExec(Single) -> SortExec on its
    own shard; the driver gathers the sorted partitions.  Same rows as q3_plan / q3_broadcast_plan / q3_distributed_plan."""
    import decimal
    import pyarrow as pa
    from .exchange import BroadcastExec
    C, L, B = ops.Column, ops.Literal, ops.BinaryExpr
    cust = ops.MemoryExec([[tables["customer"]]], _schema(CUSTOMER_SCHEMA))
    orders = ops.MemoryExec([[tables["orders"]]], _schema(ORDERS_SCHEMA))
    line = ops.MemoryExec([[tables["lineitem"]]], _schema(LINEITEM_SCHEMA))
    cb = lambda p: ops.CoalesceBatchesExec(p, batch_size)
    f_c = cb(ops.FilterExec(B(C("c_mktsegment", 1), "=", L(Q3_SEGMENT, pa.utf8())), cust))
    b_c = BroadcastExec(ops.ProjectionExec([(C("c_custkey", 0), "c_custkey")], f_c), group)
    f_o = cb(ops.FilterExec(B(C("o_orderdate", 2), "<", L(Q3_DATE, pa.date32())), orders))
    j1 = cb(ops.HashJoinExec(b_c, f_o, [(C("c_custkey", 0), C("o_custkey", 1))], None, "Inner", "CollectLeft"))
    p_j1 = ops.ProjectionExec([(C("o_orderkey", 1), "o_orderkey"), (C("o_orderdate", 3), "o_orderdate"), (C("o_shippriority", 4), "o_shippriority")], j1)
    f_l = cb(ops.FilterExec(B(C("l_shipdate", 3), ">", L(Q3_DATE, pa.date32())), line))
    p_l = ops.ProjectionExec([(C("l_orderkey", 0), "l_orderkey"), (C("l_extendedprice", 1), "l_extendedprice"), (C("l_discount", 2), "l_discount")], f_l)
    j2 = cb(ops.HashJoinExec(p_j1, p_l, [(C("o_orderkey", 0), C("l_orderkey", 0))], None, "Inner", "Partitioned"))
    p_j2 = ops.ProjectionExec([(C("o_orderdate", 1), "o_orderdate"), (C("o_shippriority", 2), "o_shippriority"), (C("l_orderkey", 3), "l_orderkey"),
                               (C("l_extendedprice", 4), "l_extendedprice"), (C("l_discount", 5), "l_discount")], j2)
    revenue = B(C("l_extendedprice", 3), "*", B(L(decimal.Decimal(1), pa.decimal128(20, 0)), "-", C("l_discount", 4)))
    agg = ops.AggregateExec("SinglePartitioned", [(C("l_orderkey", 2), "l_orderkey"), (C("o_orderdate", 0), "o_orderdate"), (C("o_shippriority", 1), "o_shippriority")],
                            [ops.AggregateFunctionExpr("SUM", revenue, "SUM(lineitem.l_extendedprice * Int64(1) - lineitem.l_discount)",
                                                       input_field=ops.Field("rev", capi.DECIMAL128, 38, 4))], p_j2)
    proj = ops.ProjectionExec([(C("l_orderkey", 0), "l_orderkey"), (C("revenue", 3), "revenue"), (C("o_orderdate", 1), "o_orderdate"), (C("o_shippriority", 2), "o_shippriority")], agg)
    return ops.SortExec([ops.PhysicalSortExpr(C("revenue", 1), descending=True, nulls_first=True),
                         ops.PhysicalSortExpr(C("o_orderdate", 2), descending=False, nulls_first=False)], proj, preserve_partitioning=True)


class Q3ColocatedStaged:
    """q3_colocated_plan built ONCE and executed per step: the plan is cut at its only exchange into two C++ segments -- (1) filter +
    project the local customers, (2) everything above the broadcast -- joined by a MemoryExec input slot that receives whatever the
    all-gather delivered (`MemoryExec.replace`); each execution runs `with_fresh_state` copies (no cached build side).  Building the
    ~25 plan nodes and their literals through ctypes costs ~0.9 ms per step, as much as a rank's whole share of the SF100 kernels
    at 8 GPUs.  Same operators, same results as q3_colocated_plan."""

    def __init__(self, tables: Dict[str, ops.RecordBatch], group=None, batch_size: int = 8192):
        import decimal
        import pyarrow as pa
        from .exchange import BroadcastExec
        C, L, B = ops.Column, ops.Literal, ops.BinaryExpr
        ctx = tables["customer"].ctx
        cust = ops.MemoryExec([[tables["customer"]]], _schema(CUSTOMER_SCHEMA))
        orders = ops.MemoryExec([[tables["orders"]]], _schema(ORDERS_SCHEMA))
        line = ops.MemoryExec([[tables["lineitem"]]], _schema(LINEITEM_SCHEMA))
        cb = lambda p: ops.CoalesceBatchesExec(p, batch_size)
        f_c = cb(ops.FilterExec(B(C("c_mktsegment", 1), "=", L(Q3_SEGMENT, pa.utf8())), cust))
        self.stage1 = ops.ProjectionExec([(C("c_custkey", 0), "c_custkey")], f_c)
        self.bcast = BroadcastExec(self.stage1, group)
        empty = ops.batch_from_arrow(ctx, pa.table({"c_custkey": pa.array([], type=pa.int64())}))
        self.slot = ops.MemoryExec([[empty]], empty.schema)
        f_o = cb(ops.FilterExec(B(C("o_orderdate", 2), "<", L(Q3_DATE, pa.date32())), orders))
        j1 = cb(ops.HashJoinExec(self.slot, f_o, [(C("c_custkey", 0), C("o_custkey", 1))], None, "Inner", "CollectLeft"))
        p_j1 = ops.ProjectionExec([(C("o_orderkey", 1), "o_orderkey"), (C("o_orderdate", 3), "o_orderdate"), (C("o_shippriority", 4), "o_shippriority")], j1)
        f_l = cb(ops.FilterExec(B(C("l_shipdate", 3), ">", L(Q3_DATE, pa.date32())), line))
        p_l = ops.ProjectionExec([(C("l_orderkey", 0), "l_orderkey"), (C("l_extendedprice", 1), "l_extendedprice"), (C("l_discount", 2), "l_discount")], f_l)
        j2 = cb(ops.HashJoinExec(p_j1, p_l, [(C("o_orderkey", 0), C("l_orderkey", 0))], None, "Inner", "Partitioned"))
        p_j2 = ops.ProjectionExec([(C("o_orderdate", 1), "o_orderdate"), (C("o_shippriority", 2), "o_shippriority"), (C("l_orderkey", 3), "l_orderkey"),
                                   (C("l_extendedprice", 4), "l_extendedprice"), (C("l_discount", 5), "l_discount")], j2)
        revenue = B(C("l_extendedprice", 3), "*", B(L(decimal.Decimal(1), pa.decimal128(20, 0)), "-", C("l_discount", 4)))
        agg = ops.AggregateExec("SinglePartitioned", [(C("l_orderkey", 2), "l_orderkey"), (C("o_orderdate", 0), "o_orderdate"), (C("o_shippriority", 1), "o_shippriority")],
                                [ops.AggregateFunctionExpr("SUM", revenue, "SUM(lineitem.l_extendedprice * Int64(1) - lineitem.l_discount)",
                                                           input_field=ops.Field("rev", capi.DECIMAL128, 38, 4))], p_j2)
        proj = ops.ProjectionExec([(C("l_orderkey", 0), "l_orderkey"), (C("revenue", 3), "revenue"), (C("o_orderdate", 1), "o_orderdate"), (C("o_shippriority", 2), "o_shippriority")], agg)
        self.stage2 = ops.SortExec([ops.PhysicalSortExpr(C("revenue", 1), descending=True, nulls_first=True),
                                    ops.PhysicalSortExpr(C("o_orderdate", 2), descending=False, nulls_first=False)], proj, preserve_partitioning=True)

    def schema(self):
        return self.stage2.schema()

    def output_partitioning(self):
        return self.stage2.output_partitioning()

    def execute(self, partition: int, context: ops.TaskContext):
        self.stage2.handle(context)                                   # the input slot exists before it is filled
        got = [b for b in self.bcast.execute(0, context)]
        self.slot.replace([got])
        yield from ops.with_fresh_state(self.stage2).execute(partition, context)


def q3_distributed_plan(tables: Dict[str, ops.RecordBatch], group=None, batch_size: int = 8192) -> ops.ExecutionPlan:
    """The reference's PARTITIONED Q3 plan (tpch/q3.slt.part physical_plan) with one output partition per GPU:
    every `RepartitionExec: partitioning=Hash(..)` becomes a ShuffleExec (device hash partition + RCCL all-to-all),
    joins run mode=Partitioned on the local partition, aggregation is Partial -> shuffle -> FinalPartitioned,
    and each rank sorts its partition (the driver gathers/merges the sorted partitions, ≙ SortPreservingMergeExec)."""
    import decimal
    import pyarrow as pa
    from .exchange import ShuffleExec
    C, L, B = ops.Column, ops.Literal, ops.BinaryExpr
    cust = ops.MemoryExec([[tables["customer"]]], _schema(CUSTOMER_SCHEMA))
    orders = ops.MemoryExec([[tables["orders"]]], _schema(ORDERS_SCHEMA))
    line = ops.MemoryExec([[tables["lineitem"]]], _schema(LINEITEM_SCHEMA))
    cb = lambda p: ops.CoalesceBatchesExec(p, batch_size)
    f_c = cb(ops.FilterExec(B(C("c_mktsegment", 1), "=", L(Q3_SEGMENT, pa.utf8())), cust))
    p_c = ops.ProjectionExec([(C("c_custkey", 0), "c_custkey")], f_c)
    s_c = cb(ShuffleExec(p_c, [C("c_custkey", 0)], group))
    f_o = cb(ops.FilterExec(B(C("o_orderdate", 2), "<", L(Q3_DATE, pa.date32())), orders))
    s_o = cb(ShuffleExec(f_o, [C("o_custkey", 1)], group))
    j1 = cb(ops.HashJoinExec(s_c, s_o, [(C("c_custkey", 0), C("o_custkey", 1))], None, "Inner", "Partitioned"))
    p_j1 = ops.ProjectionExec([(C("o_orderkey", 1), "o_orderkey"), (C("o_orderdate", 3), "o_orderdate"), (C("o_shippriority", 4), "o_shippriority")], j1)
    s_j1 = cb(ShuffleExec(p_j1, [C("o_orderkey", 0)], group))
    f_l = cb(ops.FilterExec(B(C("l_shipdate", 3), ">", L(Q3_DATE, pa.date32())), line))
    p_l = ops.ProjectionExec([(C("l_orderkey", 0), "l_orderkey"), (C("l_extendedprice", 1), "l_extendedprice"), (C("l_discount", 2), "l_discount")], f_l)
    s_l = cb(ShuffleExec(p_l, [C("l_orderkey", 0)], group))
    j2 = cb(ops.HashJoinExec(s_j1, s_l, [(C("o_orderkey", 0), C("l_orderkey", 0))], None, "Inner", "Partitioned"))
    p_j2 = ops.ProjectionExec([(C("o_orderdate", 1), "o_orderdate"), (C("o_shippriority", 2), "o_shippriority"), (C("l_orderkey", 3), "l_orderkey"),
                               (C("l_extendedprice", 4), "l_extendedprice"), (C("l_discount", 5), "l_discount")], j2)
    revenue = B(C("l_extendedprice", 3), "*", B(L(decimal.Decimal(1), pa.decimal128(20, 0)), "-", C("l_discount", 4)))
    gby = [(C("l_orderkey", 2), "l_orderkey"), (C("o_orderdate", 0), "o_orderdate"), (C("o_shippriority", 1), "o_shippriority")]
    aggr = [ops.AggregateFunctionExpr("SUM", revenue, "SUM(lineitem.l_extendedprice * Int64(1) - lineitem.l_discount)",
                                      input_field=ops.Field("rev", capi.DECIMAL128, 38, 4))]
    partial = ops.AggregateExec("Partial", gby, aggr, p_j2)
    s_a = cb(ShuffleExec(partial, [C("l_orderkey", 0), C("o_orderdate", 1), C("o_shippriority", 2)], group))
    final = ops.AggregateExec("FinalPartitioned", [(C("l_orderkey", 0), "l_orderkey"), (C("o_orderdate", 1), "o_orderdate"), (C("o_shippriority", 2), "o_shippriority")], aggr, s_a)
    proj = ops.ProjectionExec([(C("l_orderkey", 0), "l_orderkey"), (C("revenue", 3), "revenue"), (C("o_orderdate", 1), "o_orderdate"), (C("o_shippriority", 2), "o_shippriority")], final)
    return ops.SortExec([ops.PhysicalSortExpr(C("revenue", 1), descending=True, nulls_first=True),
                         ops.PhysicalSortExpr(C("o_orderdate", 2), descending=False, nulls_first=False)], proj, preserve_partitioning=True)
