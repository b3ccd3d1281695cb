"""N > 1 forms of BASELINE.json configs 4 and 5: TPC-H Q5 as the reference's fully partitioned plan (every join / aggregate input behind a
RepartitionExec Hash -> here a ShuffleExec: device hash partition + RCCL all-to-all) and the ClickBench Q28 shape (high-cardinality group-by on a
dictionary-encoded string key: Partial aggregate -> shuffle of the partial states on the key -> FinalPartitioned -> HAVING -> top 25).

Plan shapes: sqllogictest/test_files/tpch/q5.slt.part (physical_plan), benchmarks/queries/clickbench/queries.sql:29.
The data set does not depend on the world size: it is generated as VSHARDS = 8 virtual shards (seeded by shard number), rank r of W owns shards
[r * 8 / W, (r + 1) * 8 / W), so the result of the same scale factor is identical at 1, 2, 4 and 8 ranks (the checksum in the bench line)."""
from __future__ import annotations

import decimal
from typing import Dict, List

from . import capi
from . import physical_plan as ops

VSHARDS = 8
Q5_DATE_LO, Q5_DATE_HI = 8766, 9131          # 1994-01-01 .. 1995-01-01


def _my_shards(rank: int, world: int) -> range:
    assert VSHARDS % world == 0, "world size must divide 8"
    return range(rank * VSHARDS // world, (rank + 1) * VSHARDS // world)


def q5_tensors(sf: float, rank: int, world: int, device="cuda") -> Dict[str, "torch.Tensor"]:
    """Range-sharded orders / lineitem / customer / supplier columns of this rank (TPC-H shapes: SF x 1.5 M orders, 1..7 lines each, 150 K customers,
    10 K suppliers; keys dense, orders clustered on o_orderkey as dbgen writes them)."""
    import torch
    n_orders, n_cust, n_supp = int(1_500_000 * sf), max(25, int(150_000 * sf)), max(10, int(10_000 * sf))
    per = (n_orders + VSHARDS - 1) // VSHARDS
    cols = {k: [] for k in ("o_orderkey", "o_custkey", "o_orderdate", "l_orderkey", "l_suppkey", "l_price", "l_disc", "c_custkey", "c_nationkey", "s_suppkey", "s_nationkey")}
    for v in _my_shards(rank, world):
        g = torch.Generator(device=device); g.manual_seed(5_000_000 + v)
        lo, hi = v * per, min(n_orders, (v + 1) * per)
        i = torch.arange(lo, hi, dtype=torch.int64, device=device)
        okey = (i // 8) * 32 + (i % 8) + 1
        lines = torch.randint(1, 8, (hi - lo,), generator=g, device=device, dtype=torch.int64)
        lkey = torch.repeat_interleave(okey, lines)
        nl = lkey.numel()
        cols["o_orderkey"].append(okey)
        cols["o_custkey"].append(torch.randint(1, n_cust + 1, (hi - lo,), generator=g, device=device, dtype=torch.int64))
        cols["o_orderdate"].append(torch.randint(8035, 10441, (hi - lo,), generator=g, device=device, dtype=torch.int32))
        cols["l_orderkey"].append(lkey)
        cols["l_suppkey"].append(torch.randint(1, n_supp + 1, (nl,), generator=g, device=device, dtype=torch.int64))
        for name, a, b in (("l_price", 90000, 10494951), ("l_disc", 0, 11)):
            d = torch.zeros((nl, 2), dtype=torch.int64, device=device); d[:, 0] = torch.randint(a, b, (nl,), generator=g, device=device, dtype=torch.int64)
            cols[name].append(d)
        clo, chi = v * n_cust // VSHARDS, (v + 1) * n_cust // VSHARDS
        cols["c_custkey"].append(torch.arange(clo + 1, chi + 1, dtype=torch.int64, device=device))
        cols["c_nationkey"].append(torch.randint(0, 25, (chi - clo,), generator=g, device=device, dtype=torch.int64))
        slo, shi = v * n_supp // VSHARDS, (v + 1) * n_supp // VSHARDS
        cols["s_suppkey"].append(torch.arange(slo + 1, shi + 1, dtype=torch.int64, device=device))
        cols["s_nationkey"].append(torch.randint(0, 25, (shi - slo,), generator=g, device=device, dtype=torch.int64))
    return {k: (torch.cat(v) if len(v) > 1 else v[0]) for k, v in cols.items()}


def q5_tables(ctx, tt, rank: int):
    import pyarrow as pa
    W = lambda t, ty=capi.INT64: ctx.wrap_tensor(t, ty)
    D = lambda t: ctx.wrap_tensor(t, capi.DECIMAL128, 15, 2)
    mk = lambda names, arrays: ops.RecordBatch.from_arrays(ctx, names, arrays)
    nat = pa.table({"n_nationkey": pa.array(list(range(25)), type=pa.int64()), "n_name": pa.array([f"NATION{i:02d}" for i in range(25)]), "n_regionkey": pa.array([i % 5 for i in range(25)], type=pa.int64())})
    reg = pa.table({"r_regionkey": pa.array(list(range(5)), type=pa.int64()), "r_name": pa.array(["AFRICA", "AMERICA", "ASIA", "EUROPE", "MIDDLE EAST"])})
    if rank != 0:                              # nation and region are one file each: rank 0 scans them, the others contribute no rows to the repartition
        nat, reg = nat.slice(0, 0), reg.slice(0, 0)
    return {"customer": mk(["c_custkey", "c_nationkey"], [W(tt["c_custkey"]), W(tt["c_nationkey"])]),
            "orders": mk(["o_orderkey", "o_custkey", "o_orderdate"], [W(tt["o_orderkey"]), W(tt["o_custkey"]), W(tt["o_orderdate"], capi.DATE32)]),
            "lineitem": mk(["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"], [W(tt["l_orderkey"]), W(tt["l_suppkey"]), D(tt["l_price"]), D(tt["l_disc"])]),
            "supplier": mk(["s_suppkey", "s_nationkey"], [W(tt["s_suppkey"]), W(tt["s_nationkey"])]),
            "nation": ops.batch_from_arrow(ctx, nat), "region": ops.batch_from_arrow(ctx, reg)}


def q5_plan(tables, group=None, batch_size: int = 8192, native: bool = False) -> ops.ExecutionPlan:
    """tpch/q5.slt.part: every HashJoinExec mode=Partitioned over two RepartitionExec Hash inputs, AggregateExec Partial -> RepartitionExec Hash(n_name)
    -> FinalPartitioned, SortExec per partition (the caller gathers and merges, ≙ SortPreservingMergeExec)."""
    import pyarrow as pa
    from .exchange import ShuffleExec
    C, L, B, F = ops.Column, ops.Literal, ops.BinaryExpr, ops.Field
    mem = lambda name: ops.MemoryExec([[tables[name]]], tables[name].schema)
    cb = lambda p: ops.CoalesceBatchesExec(p, batch_size)
    sh = lambda p, keys: cb(ShuffleExec(p, keys, group, native=native))          # fixed-width rows: dfgpu_exchange when asked for
    shs = lambda p, keys: cb(ShuffleExec(p, keys, group))                        # rows with a Utf8 column (n_name): the host-side exchange
    hj = lambda l, r, on: cb(ops.HashJoinExec(l, r, on, None, "Inner", "Partitioned"))
    fo = cb(ops.FilterExec(B(B(C("o_orderdate", 2), ">=", L(Q5_DATE_LO, pa.date32())), "AND", B(C("o_orderdate", 2), "<", L(Q5_DATE_HI, pa.date32()))), mem("orders")))
    po = ops.ProjectionExec([(C("o_orderkey", 0), "o_orderkey"), (C("o_custkey", 1), "o_custkey")], fo)
    j1 = hj(sh(mem("customer"), [C("c_custkey", 0)]), sh(po, [C("o_custkey", 1)]), [(C("c_custkey", 0), C("o_custkey", 1))])
    p1 = ops.ProjectionExec([(C("c_nationkey", 1), "c_nationkey"), (C("o_orderkey", 2), "o_orderkey")], j1)
    j2 = hj(sh(p1, [C("o_orderkey", 1)]), sh(mem("lineitem"), [C("l_orderkey", 0)]), [(C("o_orderkey", 1), C("l_orderkey", 0))])
    p2 = ops.ProjectionExec([(C("c_nationkey", 0), "c_nationkey"), (C("l_suppkey", 3), "l_suppkey"), (C("l_extendedprice", 4), "l_extendedprice"), (C("l_discount", 5), "l_discount")], j2)
    j3 = hj(sh(mem("supplier"), [C("s_suppkey", 0), C("s_nationkey", 1)]), sh(p2, [C("l_suppkey", 1), C("c_nationkey", 0)]), [(C("s_suppkey", 0), C("l_suppkey", 1)), (C("s_nationkey", 1), C("c_nationkey", 0))])
    p3 = ops.ProjectionExec([(C("s_nationkey", 1), "s_nationkey"), (C("l_extendedprice", 4), "l_extendedprice"), (C("l_discount", 5), "l_discount")], j3)
    fr = cb(ops.FilterExec(B(C("r_name", 1), "=", L("ASIA", pa.utf8())), mem("region")))
    jn = hj(sh(ops.ProjectionExec([(C("r_regionkey", 0), "r_regionkey")], fr), [C("r_regionkey", 0)]), shs(mem("nation"), [C("n_regionkey", 2)]), [(C("r_regionkey", 0), C("n_regionkey", 2))])
    pn = ops.ProjectionExec([(C("n_nationkey", 1), "n_nationkey"), (C("n_name", 2), "n_name")], jn)
    j4 = hj(shs(pn, [C("n_nationkey", 0)]), sh(p3, [C("s_nationkey", 0)]), [(C("n_nationkey", 0), C("s_nationkey", 0))])
    rev = B(C("l_extendedprice", 3), "*", B(L(decimal.Decimal(1), pa.decimal128(20, 0)), "-", C("l_discount", 4)))
    aggr = [ops.AggregateFunctionExpr("SUM", rev, "revenue", input_field=F("r", capi.DECIMAL128, 38, 4))]
    partial = ops.AggregateExec("Partial", [(C("n_name", 1), "n_name")], aggr, j4)
    final = ops.AggregateExec("FinalPartitioned", [(C("n_name", 0), "n_name")], aggr, shs(partial, [C("n_name", 0)]))
    return ops.SortExec([ops.PhysicalSortExpr(C("revenue", 1), True, True)], final, preserve_partitioning=True)


Q5_OUTPUT = ["n_name", "revenue"]


def clickbench_tensors(nrows_total: int, card: int, rank: int, world: int, device="cuda"):
    import torch
    per = (nrows_total + VSHARDS - 1) // VSHARDS
    ids, length, w = [], [], []
    for v in _my_shards(rank, world):
        g = torch.Generator(device=device); g.manual_seed(28_000_000 + v)
        n = min(nrows_total, (v + 1) * per) - v * per
        ids.append(torch.randint(0, card, (n,), generator=g, device=device, dtype=torch.int32))
        length.append(torch.randint(0, 500, (n,), generator=g, device=device, dtype=torch.int32))
        w.append(torch.randint(0, 10**6, (n,), generator=g, device=device, dtype=torch.int64))
    cat = lambda x: torch.cat(x) if len(x) > 1 else x[0]
    return cat(ids), cat(length), cat(w)


def clickbench_batch(ctx, ids, length, w, card: int):
    """key: Dictionary(Int32, Utf8) over `card` URL-like values (entry 0 = the empty string the query filters out); every rank holds the same dictionary,
    as every file of one table would after dictionary unification."""
    import ctypes as C
    import pyarrow as pa
    words = pa.array([""] + [f"https://site{k}.example/{k * 7919 % 1000}" for k in range(1, card)], type=pa.utf8())
    dictionary = ctx.from_arrow(words)
    d = capi.ArrayDesc(); dd = dictionary.describe()
    d.type, d.key_type, d.length, d.null_count = capi.DICTIONARY, capi.INT32, ids.numel(), 0
    d.values = ids.data_ptr(); d.dictionary = C.pointer(dd)
    key = ctx.wrap_device(d, keepalive=(ids, dictionary, dd))
    return ops.RecordBatch.from_arrays(ctx, ["key", "len", "w"], [key, ctx.wrap_tensor(length, capi.INT32), ctx.wrap_tensor(w, capi.INT64)])


def clickbench_plan(batch, group=None, batch_size: int = 8192) -> ops.ExecutionPlan:
    """SELECT key, AVG(len) l, COUNT(*) c, MAX(w) FROM hits WHERE key <> '' GROUP BY key HAVING COUNT(*) > 3 ORDER BY l DESC LIMIT 25 -- per rank: Partial
    aggregate, shuffle of the partial states on the key (Utf8 values + AVG's (count, sum) + COUNT + MAX), FinalPartitioned, HAVING, top 25."""
    import pyarrow as pa
    from .exchange import ShuffleExec
    C, L, B, F = ops.Column, ops.Literal, ops.BinaryExpr, ops.Field
    src = ops.MemoryExec([[batch]], batch.schema)
    f = ops.CoalesceBatchesExec(ops.FilterExec(B(C("key", 0), "!=", L("", pa.utf8())), src), batch_size)
    proj = ops.ProjectionExec([(C("key", 0), "key"), (ops.CastExpr(C("len", 1), capi.FLOAT64), "lenf"), (C("w", 2), "w")], f)
    aggs = [ops.AggregateFunctionExpr("AVG", C("lenf", 1), "l", input_field=F("x", capi.FLOAT64)), ops.AggregateFunctionExpr("COUNT", None, "c"),
            ops.AggregateFunctionExpr("MAX", C("w", 2), "m", input_field=F("x", capi.INT64))]
    partial = ops.AggregateExec("Partial", [(C("key", 0), "k")], aggs, proj)
    shuffled = ops.CoalesceBatchesExec(ShuffleExec(partial, [C("k", 0)], group), batch_size)
    final = ops.AggregateExec("FinalPartitioned", [(C("k", 0), "k")], aggs, shuffled)
    having = ops.FilterExec(B(C("c", 2), ">", L(3, pa.int64())), final)
    return ops.SortExec([ops.PhysicalSortExpr(C("l", 1), True, True), ops.PhysicalSortExpr(C("k", 0), False, False)], having, fetch=25, preserve_partitioning=True)


CLICKBENCH_OUTPUT = ["k", "l", "c", "m"]


def shuffle_nodes(plan) -> List:
    from .exchange import ShuffleExec
    out, stack = [], [plan]
    while stack:
        p = stack.pop()
        if isinstance(p, ShuffleExec):
            out.append(p)
        stack += list(p.children())
    return out
