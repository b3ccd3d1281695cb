"""-m gpu: the other BASELINE.json configs as parity cases on small synthetic tables (plan shapes from the reference's
EXPLAIN goldens, sqllogictest/test_files/tpch/q{1,5,18}.slt.part and benchmarks/queries/clickbench/queries.sql:29),
run through the C++ operator layer and checked against the CPU oracle's restated operators (hash_join, GroupValues,
GroupsAccumulator, expression kernels, lexsort), bit-exact for integer / Decimal128 columns, 1e-9 relative for Float64."""
import decimal

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from helpers import rows_of, sort_rows
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(42)
FLOAT_RTOL = 1e-9


def dec(vals, p=15, s=2):
    return pa.array([decimal.Decimal(int(v)).scaleb(-s) for v in vals], type=pa.decimal128(p, s))


def run(plan, ctx, batch_size=8192):
    from dfgpu import physical_plan as ops
    bs = ops.collect(plan, ops.TaskContext(ctx, batch_size))
    if not bs:
        return None
    return pa.concat_tables([pa.table({f"c{i}": c.to_arrow() for i, c in enumerate(b.columns)}) for b in bs])


def src(ctx, tables, partitions=1):
    from dfgpu import physical_plan as ops
    batches = [ops.batch_from_arrow(ctx, t) for t in tables]
    return ops.MemoryExec([batches[i::partitions] for i in range(partitions)], batches[0].schema)


def oracle_join(lt, rt, lkeys, rkeys, jt="Inner"):
    res = po.hash_join([[lt[k] for k in lkeys]], [[rt[k] for k in rkeys]], jt, batch_size=1 << 40)
    cols, names = [], []
    if jt not in ("RightSemi", "RightAnti"):
        for n in lt.column_names:
            cols.append(po.take(lt[n], res.build_idx)); names.append(n)
    if jt not in ("LeftSemi", "LeftAnti"):
        for n in rt.column_names:
            cols.append(po.take(rt[n], res.probe_idx)); names.append(n)
    return pa.table(cols, names=names)


def oracle_agg(t, keys, aggs):
    """aggs = [(fun, column or None)] -> table of keys + aggregate columns (group order = first seen)."""
    g = po.Groups([t[k].type for k in keys])
    ids = g.intern([t[k] for k in keys])
    cols = g.emit()
    for fun, col in aggs:
        acc = po.Acc(fun, t[col].type if col else pa.int64())
        acc.update_batch(t[col] if col else None, ids, None, len(g))
        cols.append(acc.evaluate())
    return pa.table(cols, names=[f"c{i}" for i in range(len(cols))])


def assert_tables(got, want, float_cols=(), ordered=False):
    assert got.num_rows == want.num_rows and got.num_columns == want.num_columns
    for i in range(want.num_columns):
        assert got.column(i).type == want.column(i).type, f"column {i}: {got.column(i).type} vs {want.column(i).type}"
    gr, wr = rows_of([got.column(i) for i in range(got.num_columns)]), rows_of([want.column(i) for i in range(want.num_columns)])
    if not ordered:
        gr, wr = sort_rows(gr), sort_rows(wr)
    for x, y in zip(gr, wr):
        for i, (a, b) in enumerate(zip(x, y)):
            if i in float_cols and a is not None and b is not None:
                assert abs(a - b) <= FLOAT_RTOL * max(1.0, abs(b)), (i, a, b)
            else:
                assert a == b, (i, x, y)


def lineitem(n, money="decimal"):
    rf, ls = RNG.integers(0, 3, n), RNG.integers(0, 2, n)
    mk = (lambda v: dec(v)) if money == "decimal" else (lambda v: pa.array(np.asarray(v, dtype=np.float64) / 100.0))
    return pa.table({"l_quantity": mk(RNG.integers(1, 51, n) * 100), "l_extendedprice": mk(RNG.integers(90000, 10494951, n)), "l_discount": mk(RNG.integers(0, 11, n)),
                     "l_tax": mk(RNG.integers(0, 9, n)), "l_returnflag": pa.array(np.array(["A", "N", "R"])[rf]), "l_linestatus": pa.array(np.array(["F", "O"])[ls]),
                     "l_shipdate": pa.array(RNG.integers(8035, 10560, n).astype(np.int32)).cast(pa.date32())})


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("money", ["decimal", "float64"])
def test_q1_grouped_aggregate(ctx, money, fused):
    """tpch/q1.slt.part: Filter l_shipdate <= 10471 -> Projection (disc_price computed once) -> Aggregate Partial (2 Utf8 keys,
    4 SUM + 3 AVG + COUNT(*)) -> Repartition Hash -> FinalPartitioned -> Sort [l_returnflag, l_linestatus].  `fused`: the Partial
    aggregate looks through the projection and evaluates its arguments inside the accumulate pass (dfgpu_acc_update_batch_fused), which
    large batches do by default; otherwise node by node."""
    import dfgpu
    from dfgpu import physical_plan as ops
    ctx.set_option("fused_aggregate_min_rows", 0 if fused else -1)
    try:
        _q1_body(ctx, money)
    finally:
        ctx.set_option("fused_aggregate_min_rows", 1 << 20)


def _q1_body(ctx, money):
    import dfgpu
    from dfgpu import physical_plan as ops
    tabs = [lineitem(6000, money), lineitem(9000, money), lineitem(50, money)]
    C, L, B, F = ops.Column, ops.Literal, ops.BinaryExpr, ops.Field
    one = L(decimal.Decimal(1), pa.decimal128(20, 0)) if money == "decimal" else L(1.0, pa.float64())
    f = ops.CoalesceBatchesExec(ops.FilterExec(B(C("l_shipdate", 6), "<=", L(10471, pa.date32())), src(ctx, tabs, 2)), 8192)
    proj = ops.ProjectionExec([(B(C("l_extendedprice", 1), "*", B(one, "-", C("l_discount", 2))), "disc_price"), (C("l_quantity", 0), "l_quantity"), (C("l_extendedprice", 1), "l_extendedprice"),
                               (C("l_discount", 2), "l_discount"), (C("l_tax", 3), "l_tax"), (C("l_returnflag", 4), "l_returnflag"), (C("l_linestatus", 5), "l_linestatus")], f)
    charge = B(C("disc_price", 0), "*", B(one, "+", C("l_tax", 4)))
    m = (lambda p, s: F("x", dfgpu.capi.DECIMAL128, p, s)) if money == "decimal" else (lambda p, s: F("x", dfgpu.capi.FLOAT64))
    aggs = [ops.AggregateFunctionExpr("SUM", C("l_quantity", 1), "sum_qty", input_field=m(15, 2)), ops.AggregateFunctionExpr("SUM", C("l_extendedprice", 2), "sum_base_price", input_field=m(15, 2)),
            ops.AggregateFunctionExpr("SUM", C("disc_price", 0), "sum_disc_price", input_field=m(38, 4)), ops.AggregateFunctionExpr("SUM", charge, "sum_charge", input_field=m(38, 6)),
            ops.AggregateFunctionExpr("AVG", C("l_quantity", 1), "avg_qty", input_field=m(15, 2)), ops.AggregateFunctionExpr("AVG", C("l_extendedprice", 2), "avg_price", input_field=m(15, 2)),
            ops.AggregateFunctionExpr("AVG", C("l_discount", 3), "avg_disc", input_field=m(15, 2)), ops.AggregateFunctionExpr("COUNT", None, "count_order")]
    gby = [(C("l_returnflag", 5), "l_returnflag"), (C("l_linestatus", 6), "l_linestatus")]
    partial = ops.AggregateExec("Partial", gby, aggs, proj)
    rep = ops.CoalesceBatchesExec(ops.RepartitionExec(partial, ops.Partitioning.Hash([C("l_returnflag", 0), C("l_linestatus", 1)], 3)), 8192)
    final = ops.AggregateExec("FinalPartitioned", [(C("l_returnflag", 0), "l_returnflag"), (C("l_linestatus", 1), "l_linestatus")], aggs, rep)
    plan = ops.SortExec([ops.PhysicalSortExpr(C("l_returnflag", 0), False, False), ops.PhysicalSortExpr(C("l_linestatus", 1), False, False)], final)
    got = run(plan, ctx)
    # oracle: the same operators restated on the CPU
    whole = pa.concat_tables(tabs)
    keep = po.binary("<=", whole["l_shipdate"], pa.array([10471], type=pa.int32()).cast(pa.date32()), r_scalar=True)
    t = pa.table([po.filter_(whole[c], keep) for c in whole.column_names], names=whole.column_names)
    one_a = pa.array([decimal.Decimal(1)], type=pa.decimal128(20, 0)) if money == "decimal" else pa.array([1.0])
    disc_price = po.binary("*", t["l_extendedprice"], po.binary("-", one_a, t["l_discount"], l_scalar=True))
    chg = po.binary("*", disc_price, po.binary("+", one_a, t["l_tax"], l_scalar=True))
    t = t.append_column("disc_price", disc_price).append_column("charge", chg)
    want = oracle_agg(t, ["l_returnflag", "l_linestatus"], [("SUM", "l_quantity"), ("SUM", "l_extendedprice"), ("SUM", "disc_price"), ("SUM", "charge"),
                                                               ("AVG", "l_quantity"), ("AVG", "l_extendedprice"), ("AVG", "l_discount"), ("COUNT", None)])
    idx = po.lexsort_to_indices([want["c0"], want["c1"]], [False, False], [False, False])
    want = want.take(pa.array(idx))
    assert_tables(got, want, float_cols=range(2, 9) if money == "float64" else (), ordered=True)
    if money == "decimal":      # result dtypes of the reference's Q1 golden (scales 2, 4, 6; AVG scale 6)
        assert [got.column(i).type for i in (2, 4, 5, 6)] == [pa.decimal128(25, 2), pa.decimal128(38, 4), pa.decimal128(38, 6), pa.decimal128(19, 6)]


def test_q5_partitioned_multi_join_two_key(ctx):
    """tpch/q5.slt.part: chained HashJoinExec mode=Partitioned over RepartitionExec Hash, one join on TWO keys
    (l_suppkey, c_nationkey) = (s_suppkey, s_nationkey), filter r_name = ASIA, SUM by n_name (Utf8), sort revenue DESC."""
    from dfgpu import physical_plan as ops
    import dfgpu
    nc, no, nl, ns = 600, 3000, 12000, 80
    RNG = np.random.default_rng(77)          # own stream: the 5-nation sanity check below must not depend on which tests ran before
    customer = pa.table({"c_custkey": pa.array(np.arange(nc, dtype=np.int64)), "c_nationkey": pa.array(RNG.integers(0, 25, nc).astype(np.int64))})
    orders = pa.table({"o_orderkey": pa.array(np.arange(no, dtype=np.int64) * 4 + 1), "o_custkey": pa.array(RNG.integers(0, nc, no).astype(np.int64)),
                       "o_orderdate": pa.array(RNG.integers(8766, 9500, no).astype(np.int32)).cast(pa.date32())})
    line = pa.table({"l_orderkey": pa.array(RNG.integers(0, no, nl).astype(np.int64) * 4 + 1), "l_suppkey": pa.array(RNG.integers(0, ns, nl).astype(np.int64)),
                     "l_extendedprice": dec(RNG.integers(90000, 10494951, nl)), "l_discount": dec(RNG.integers(0, 11, nl))})
    supplier = pa.table({"s_suppkey": pa.array(np.arange(ns, dtype=np.int64)), "s_nationkey": pa.array(RNG.integers(0, 25, ns).astype(np.int64))})
    nation = pa.table({"n_nationkey": pa.array(np.arange(25, dtype=np.int64)), "n_name": pa.array([f"NATION{i:02d}" for i in range(25)]), "n_regionkey": pa.array((np.arange(25) % 5).astype(np.int64))})
    region = pa.table({"r_regionkey": pa.array(np.arange(5, dtype=np.int64)), "r_name": pa.array(["AFRICA", "AMERICA", "ASIA", "EUROPE", "MIDDLE EAST"])})
    C, L, B = ops.Column, ops.Literal, ops.BinaryExpr
    P = 4
    rep = lambda plan, keys: ops.CoalesceBatchesExec(ops.RepartitionExec(plan, ops.Partitioning.Hash(keys, P)), 8192)
    hj = lambda l, r, on: ops.CoalesceBatchesExec(ops.HashJoinExec(l, r, on, None, "Inner", "Partitioned"), 8192)
    fo = ops.FilterExec(B(B(C("o_orderdate", 2), ">=", L(8766, pa.date32())), "AND", B(C("o_orderdate", 2), "<", L(9131, pa.date32()))), src(ctx, [orders], 2))
    j1 = hj(rep(src(ctx, [customer], 2), [C("c_custkey", 0)]), rep(fo, [C("o_custkey", 1)]), [(C("c_custkey", 0), C("o_custkey", 1))])          # c_custkey,c_nationkey,o_orderkey,o_custkey,o_orderdate
    p1 = ops.ProjectionExec([(C("c_nationkey", 1), "c_nationkey"), (C("o_orderkey", 2), "o_orderkey")], j1)
    j2 = hj(rep(p1, [C("o_orderkey", 1)]), rep(src(ctx, [line], 3), [C("l_orderkey", 0)]), [(C("o_orderkey", 1), C("l_orderkey", 0))])       # c_nationkey,o_orderkey,l_orderkey,l_suppkey,ext,disc
    p2 = ops.ProjectionExec([(C("c_nationkey", 0), "c_nationkey"), (C("l_suppkey", 3), "l_suppkey"), (C("l_extendedprice", 4), "l_extendedprice"), (C("l_discount", 5), "l_discount")], j2)
    j3 = hj(rep(src(ctx, [supplier]), [C("s_suppkey", 0), C("s_nationkey", 1)]), rep(p2, [C("l_suppkey", 1), C("c_nationkey", 0)]),
            [(C("s_suppkey", 0), C("l_suppkey", 1)), (C("s_nationkey", 1), C("c_nationkey", 0))])                                                    # s_suppkey,s_nationkey,c_nationkey,l_suppkey,ext,disc
    p3 = ops.ProjectionExec([(C("s_nationkey", 1), "s_nationkey"), (C("l_extendedprice", 4), "l_extendedprice"), (C("l_discount", 5), "l_discount")], j3)
    fr = ops.FilterExec(B(C("r_name", 1), "=", L("ASIA", pa.utf8())), src(ctx, [region]))
    jn = hj(rep(ops.ProjectionExec([(C("r_regionkey", 0), "r_regionkey")], fr), [C("r_regionkey", 0)]), rep(src(ctx, [nation]), [C("n_regionkey", 2)]), [(C("r_regionkey", 0), C("n_regionkey", 2))])
    pn = ops.ProjectionExec([(C("n_nationkey", 1), "n_nationkey"), (C("n_name", 2), "n_name")], jn)
    j4 = hj(rep(pn, [C("n_nationkey", 0)]), rep(p3, [C("s_nationkey", 0)]), [(C("n_nationkey", 0), C("s_nationkey", 0))])                       # n_nationkey,n_name,s_nationkey,ext,disc
    rev = B(C("l_extendedprice", 3), "*", B(L(decimal.Decimal(1), pa.decimal128(20, 0)), "-", C("l_discount", 4)))
    aggs = [ops.AggregateFunctionExpr("SUM", rev, "revenue", input_field=ops.Field("r", dfgpu.capi.DECIMAL128, 38, 4))]
    partial = ops.AggregateExec("Partial", [(C("n_name", 1), "n_name")], aggs, j4)
    final = ops.AggregateExec("FinalPartitioned", [(C("n_name", 0), "n_name")], aggs, rep(partial, [C("n_name", 0)]))
    plan = ops.SortExec([ops.PhysicalSortExpr(C("revenue", 1), True, True)], final)
    got = run(plan, ctx)
    # oracle
    okeep = pc.and_(pc.greater_equal(orders["o_orderdate"].cast(pa.int32()), 8766), pc.less(orders["o_orderdate"].cast(pa.int32()), 9131))
    t = oracle_join(customer, orders.filter(okeep), ["c_custkey"], ["o_custkey"]).select(["c_nationkey", "o_orderkey"])
    t = oracle_join(t, line, ["o_orderkey"], ["l_orderkey"]).select(["c_nationkey", "l_suppkey", "l_extendedprice", "l_discount"])
    t = oracle_join(supplier, t, ["s_suppkey", "s_nationkey"], ["l_suppkey", "c_nationkey"]).select(["s_nationkey", "l_extendedprice", "l_discount"])
    nt = oracle_join(region.filter(pc.equal(region["r_name"], "ASIA")).select(["r_regionkey"]), nation, ["r_regionkey"], ["n_regionkey"]).select(["n_nationkey", "n_name"])
    t = oracle_join(nt, t, ["n_nationkey"], ["s_nationkey"])
    r = po.binary("*", t["l_extendedprice"], po.binary("-", pa.array([decimal.Decimal(1)], type=pa.decimal128(20, 0)), t["l_discount"], l_scalar=True))
    want = oracle_agg(t.append_column("rev", r), ["n_name"], [("SUM", "rev")])
    want = want.take(pa.array(po.lexsort_to_indices([want["c1"]], [True], [True])))
    assert want.num_rows == 5
    assert_tables(got, want, ordered=True)


def test_q18_semi_join_and_high_cardinality_groups(ctx):
    """tpch/q18.slt.part: subquery AggregateExec gby=[l_orderkey] SUM(l_quantity) (one Int64 key = GroupValuesPrimitive) ->
    FilterExec SUM > 300 -> HashJoinExec LeftSemi against the main 3-way join -> 5-key group-by -> sort."""
    from dfgpu import physical_plan as ops
    import dfgpu
    nc, no, nl = 500, 20000, 80000
    customer = pa.table({"c_custkey": pa.array(np.arange(nc, dtype=np.int64)), "c_name": pa.array([f"Customer#{i:09d}" for i in range(nc)])})
    orders = pa.table({"o_orderkey": pa.array(np.arange(no, dtype=np.int64) * 4 + 1), "o_custkey": pa.array(RNG.integers(0, nc, no).astype(np.int64)),
                       "o_totalprice": dec(RNG.integers(10**5, 5 * 10**7, no)), "o_orderdate": pa.array(RNG.integers(8035, 10440, no).astype(np.int32)).cast(pa.date32())})
    lkeys = np.concatenate([RNG.integers(0, no, nl - 4000), RNG.integers(0, 40, 4000)]).astype(np.int64) * 4 + 1      # a few heavy orders pass the HAVING
    line = pa.table({"l_orderkey": pa.array(lkeys), "l_quantity": dec(RNG.integers(1, 51, nl) * 100)})
    C, L, B = ops.Column, ops.Literal, ops.BinaryExpr
    qf = ops.Field("q", dfgpu.capi.DECIMAL128, 15, 2)
    sub = ops.AggregateExec("Single", [(C("l_orderkey", 0), "l_orderkey")], [ops.AggregateFunctionExpr("SUM", C("l_quantity", 1), "SUM(l_quantity)", input_field=qf)],
                            ops.CoalescePartitionsExec(src(ctx, [line.slice(0, 30000), line.slice(30000)], 2)))
    having = ops.CoalesceBatchesExec(ops.FilterExec(B(C("SUM(l_quantity)", 1), ">", L(decimal.Decimal(300), pa.decimal128(25, 2))), sub), 8192)
    j1 = ops.HashJoinExec(src(ctx, [customer]), src(ctx, [orders]), [(C("c_custkey", 0), C("o_custkey", 1))], None, "Inner", "CollectLeft")     # c_custkey,c_name,o_orderkey,o_custkey,o_totalprice,o_orderdate
    j2 = ops.HashJoinExec(j1, src(ctx, [line]), [(C("o_orderkey", 2), C("l_orderkey", 0))], None, "Inner", "CollectLeft")                       # + l_orderkey,l_quantity
    semi = ops.HashJoinExec(j2, ops.ProjectionExec([(C("l_orderkey", 0), "l_orderkey")], having), [(C("o_orderkey", 2), C("l_orderkey", 0))], None, "LeftSemi", "CollectLeft")
    gby = [(C("c_name", 1), "c_name"), (C("c_custkey", 0), "c_custkey"), (C("o_orderkey", 2), "o_orderkey"), (C("o_orderdate", 5), "o_orderdate"), (C("o_totalprice", 4), "o_totalprice")]
    agg = ops.AggregateExec("Single", gby, [ops.AggregateFunctionExpr("SUM", C("l_quantity", 7), "SUM(l_quantity)", input_field=qf)], semi)
    plan = ops.SortExec([ops.PhysicalSortExpr(C("o_totalprice", 4), True, True), ops.PhysicalSortExpr(C("o_orderdate", 3), False, False)], agg)
    got = run(plan, ctx)
    # oracle
    s = oracle_agg(line, ["l_orderkey"], [("SUM", "l_quantity")])
    big = s.filter(po.binary(">", s["c1"], pa.array([decimal.Decimal(300)], type=pa.decimal128(25, 2)), r_scalar=True)).select(["c0"]).rename_columns(["k"])
    assert 0 < big.num_rows < 200
    t = oracle_join(oracle_join(customer, orders, ["c_custkey"], ["o_custkey"]), line, ["o_orderkey"], ["l_orderkey"])
    t = oracle_join(t, big, ["o_orderkey"], ["k"], "LeftSemi")
    want = oracle_agg(t, ["c_name", "c_custkey", "o_orderkey", "o_orderdate", "o_totalprice"], [("SUM", "l_quantity")])
    want = want.take(pa.array(po.lexsort_to_indices([want["c4"], want["c3"]], [True, False], [True, False])))
    assert_tables(got, want, ordered=False)
    gl = rows_of([got["c4"], got["c3"]]); assert gl == sorted(gl, key=lambda r: (-r[0], r[1]))       # sorted by o_totalprice DESC, o_orderdate


@pytest.mark.parametrize("card,zipf", [(1000, True), (50000, False)])
def test_clickbench_style_string_key_groupby(ctx, card, zipf):
    """ClickBench Q28 shape (benchmarks/queries/clickbench/queries.sql:29): filter key <> '' -> GROUP BY a dictionary-encoded string key
    (GroupValuesByes): AVG(len), COUNT(*), MAX(int) -> HAVING COUNT(*) > k -> ORDER BY avg DESC LIMIT 25 (TopK via SortExec fetch)."""
    from dfgpu import physical_plan as ops
    import dfgpu
    n = 120000
    ids = (RNG.zipf(1.1, n) % card) if zipf else RNG.integers(0, card, n)
    keys = pa.array([("" if i % 97 == 0 else f"https://site{i}.example/{i * 7919 % 1000}") for i in ids], type=pa.utf8())
    t = pa.table({"key": keys.dictionary_encode(), "len": pa.array(RNG.integers(0, 500, n).astype(np.int32)), "w": pa.array(RNG.integers(0, 10**6, n).astype(np.int64))})
    C, L, B = ops.Column, ops.Literal, ops.BinaryExpr
    f = ops.CoalesceBatchesExec(ops.FilterExec(B(C("key", 0), "!=", L("", pa.utf8())), src(ctx, [t.slice(0, 50000), t.slice(50000)], 2)), 8192)
    proj = ops.ProjectionExec([(C("key", 0), "key"), (ops.CastExpr(C("len", 1), dfgpu.capi.FLOAT64), "lenf"), (C("w", 2), "w")], f)
    aggs = [ops.AggregateFunctionExpr("AVG", C("lenf", 1), "l", input_field=ops.Field("x", dfgpu.capi.FLOAT64)), ops.AggregateFunctionExpr("COUNT", None, "c"),
            ops.AggregateFunctionExpr("MAX", C("w", 2), "m", input_field=ops.Field("x", dfgpu.capi.INT64))]
    partial = ops.AggregateExec("Partial", [(C("key", 0), "k")], aggs, proj)
    final = ops.AggregateExec("FinalPartitioned", [(C("k", 0), "k")], aggs, ops.CoalesceBatchesExec(ops.RepartitionExec(partial, ops.Partitioning.Hash([C("k", 0)], 4)), 8192))
    having = ops.FilterExec(B(C("c", 2), ">", L(3, pa.int64())), final)
    plan = ops.SortExec([ops.PhysicalSortExpr(C("l", 1), True, True), ops.PhysicalSortExpr(C("k", 0), False, False)], having, fetch=25)
    got = run(plan, ctx)
    plain = pa.table({"key": keys, "lenf": t["len"].cast(pa.float64()), "w": t["w"]}).filter(pc.not_equal(keys, ""))
    w = oracle_agg(plain, ["key"], [("AVG", "lenf"), ("COUNT", None), ("MAX", "w")])
    w = w.filter(pc.greater(w["c2"], 3))
    w = w.take(pa.array(po.lexsort_to_indices([w["c1"], w["c0"]], [True, False], [True, False], fetch=25)))
    assert got.num_rows == min(25, w.num_rows)
    gr, wr = rows_of([got[c] for c in got.column_names]), rows_of([w[c] for c in w.column_names])
    for x, y in zip(gr, wr):
        assert x[0] == y[0] and x[2:] == y[2:] and abs(x[1] - y[1]) <= FLOAT_RTOL * max(1.0, abs(y[1]))


def test_aggregate_without_group_by(ctx):
    """AggregateStream (aggregates/no_grouping.rs): one output row, also for empty input (COUNT 0, SUM/MIN/AVG NULL); fused FilterExec honoured."""
    from dfgpu import physical_plan as ops
    import dfgpu
    t = pa.table({"v": pa.array(RNG.integers(-1000, 1000, 50000).astype(np.int64), mask=RNG.random(50000) < 0.1), "f": pa.array(RNG.normal(size=50000)),
                  "d": dec(RNG.integers(-10**6, 10**6, 50000))})
    C, L, B, F = ops.Column, ops.Literal, ops.BinaryExpr, ops.Field
    aggs = [ops.AggregateFunctionExpr("SUM", C("v", 0), "s", input_field=F("x", dfgpu.capi.INT64)), ops.AggregateFunctionExpr("COUNT", C("v", 0), "c"), ops.AggregateFunctionExpr("COUNT", None, "n"),
            ops.AggregateFunctionExpr("MIN", C("v", 0), "mn", input_field=F("x", dfgpu.capi.INT64)), ops.AggregateFunctionExpr("AVG", C("f", 1), "af", input_field=F("x", dfgpu.capi.FLOAT64)),
            ops.AggregateFunctionExpr("SUM", C("d", 2), "sd", input_field=F("x", dfgpu.capi.DECIMAL128, 15, 2)), ops.AggregateFunctionExpr("MAX", C("d", 2), "xd", input_field=F("x", dfgpu.capi.DECIMAL128, 15, 2))]
    for cut, parts in [(0, 1), (-5000, 2)]:          # second predicate selects nothing
        f = ops.FilterExec(B(C("v", 0), ">", L(cut, pa.int64())) if cut == 0 else B(C("v", 0), "<", L(cut, pa.int64())), src(ctx, [t.slice(0, 20000), t.slice(20000)], parts))
        partial = ops.AggregateExec("Partial", [], aggs, f)
        final = ops.AggregateExec("Final", [], aggs, partial)
        got = run(final, ctx)
        sel = t.filter(pc.fill_null(pc.greater(t["v"], 0) if cut == 0 else pc.less(t["v"], cut), False))
        assert got.num_rows == 1
        row = rows_of([got[c] for c in got.column_names])[0]
        if sel.num_rows:
            assert row[0] == pc.sum(sel["v"]).as_py() and row[1] == sel.num_rows - sel["v"].null_count and row[2] == sel.num_rows and row[3] == pc.min(sel["v"]).as_py()
            assert abs(row[4] - pc.mean(sel["f"]).as_py()) <= FLOAT_RTOL * abs(pc.mean(sel["f"]).as_py()) + 1e-12
            assert row[5] == sum(sel["d"].to_pylist()) and row[6] == max(sel["d"].to_pylist())
        else:
            assert row == [None, 0, 0, None, None, None, None]


def test_aggregate_grouping_sets_reference_vector(ctx):
    """aggregates/mod.rs:1353-1507 (`check_grouping_sets`, no spill): PhysicalGroupBy with the sets (a, NULL), (NULL, b), (a, b) over
    `some_data()` (:1256-1286), COUNT(1) Partial then Final over the merged partitions; both expected tables transcribed."""
    import dfgpu
    from dfgpu import physical_plan as ops
    a0, b0 = pa.array([2, 3, 4, 4], type=pa.uint32()), pa.array([1.0, 2.0, 3.0, 4.0])
    a1, b1 = pa.array([2, 3, 3, 4], type=pa.uint32()), pa.array([1.0, 2.0, 3.0, 4.0])
    tabs = [pa.table({"a": a0, "b": b0}), pa.table({"a": a1, "b": b1})]
    C, L = ops.Column, ops.Literal
    gb = ops.PhysicalGroupBy([(C("a", 0), "a"), (C("b", 1), "b")], [(L(None, pa.uint32()), "a"), (L(None, pa.float64()), "b")],
                             [[False, True], [True, False], [False, False]])
    aggs = [ops.AggregateFunctionExpr("COUNT", L(1, pa.int8()), "COUNT(1)")]
    partial = ops.AggregateExec("Partial", gb, aggs, src(ctx, tabs, 1))
    want = [[None, 1.0, 2], [None, 2.0, 2], [None, 3.0, 2], [None, 4.0, 2], [2, None, 2], [2, 1.0, 2], [3, None, 3], [3, 2.0, 2], [3, 3.0, 1], [4, None, 3], [4, 3.0, 1], [4, 4.0, 2]]
    assert ops.collect(ops.AggregateExec("Partial", gb, aggs, src(ctx, tabs, 1)), ops.TaskContext(ctx, 8192))[0].schema.names() == ["a", "b", "COUNT(1)[count]"]
    got = run(partial, ctx)
    assert sort_rows(rows_of([got.column(i) for i in range(3)])) == sort_rows(want)
    final = ops.AggregateExec("Final", [(C("a", 0), "a"), (C("b", 1), "b")], aggs, ops.AggregateExec("Partial", gb, aggs, src(ctx, tabs, 2)))
    got = run(final, ctx)
    assert got.num_rows == 12
    assert sort_rows(rows_of([got.column(i) for i in range(3)])) == sort_rows(want)
