"""-m gpu: the operator layer (FilterExec / ProjectionExec / AggregateExec / SortExec / RepartitionExec / CoalesceBatchesExec)
driven like the reference's own tests: MemoryExec source -> execute -> collect, results vs pyarrow / the CPU oracle."""
import decimal

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from helpers import rows_of, sort_rows
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(3)


def table(n, nulls=0.1):
    return pa.table({
        "k": pa.array(RNG.integers(0, 40, n).astype(np.int64), mask=RNG.random(n) < nulls),
        "d": pa.array(RNG.integers(8000, 9000, n).astype(np.int32), mask=RNG.random(n) < nulls).cast(pa.date32()),
        "v": pa.array([None if RNG.random() < nulls else decimal.Decimal(int(x)).scaleb(-2) for x in RNG.integers(-10**6, 10**6, n)], type=pa.decimal128(15, 2)),
        "f": pa.array(RNG.normal(size=n), mask=RNG.random(n) < nulls),
        "s": pa.array([None if RNG.random() < nulls else f"s{x}" for x in RNG.integers(0, 7, n)], type=pa.utf8()),
    })


def source(ctx, tabs, partitions=1):
    from dfgpu import physical_plan as ops
    batches = [ops.batch_from_arrow(ctx, t) for t in tabs]
    per = [batches[i::partitions] for i in range(partitions)]
    return ops.MemoryExec(per, batches[0].schema)


def collect_table(plan, tc):
    from dfgpu import physical_plan as ops
    bs = ops.collect(plan, tc)
    if not bs:
        return None
    return pa.concat_tables([pa.table({f"c{i}": c.to_arrow() for i, c in enumerate(b.columns)}) for b in bs])


def test_filter_projection_match_arrow(ctx, task_ctx):
    """FilterExec keeps input order and drops NULL predicate rows (filter.rs:222-225, :315-327); ProjectionExec evaluates per column."""
    from dfgpu import physical_plan as ops
    tabs = [table(5000), table(1), table(3000)]
    src = source(ctx, tabs)
    C, L, B = ops.Column, ops.Literal, ops.BinaryExpr
    pred = B(B(C("d", 1), ">", L(8400, pa.date32())), "AND", B(B(C("k", 0), "<", L(30, pa.int64())), "OR", ops.IsNullExpr(C("s", 4))))
    plan = ops.ProjectionExec([(C("k", 0), "k"), (B(C("v", 2), "*", B(L(decimal.Decimal(1), pa.decimal128(20, 0)), "-", C("v", 2))), "e"),
                               (ops.CastExpr(C("k", 0), 11), "kf"), (C("s", 4), "s")], ops.FilterExec(pred, src))
    got = collect_table(plan, task_ctx)
    whole = pa.concat_tables(tabs)
    m = pc.and_kleene(pc.greater(whole["d"], pa.scalar(8400, pa.int32()).cast(pa.date32())), pc.or_kleene(pc.less(whole["k"], 30), pc.is_null(whole["s"])))
    want = whole.filter(m)
    assert got["c0"].combine_chunks().equals(want["k"].combine_chunks())
    assert got["c3"].combine_chunks().equals(want["s"].combine_chunks())
    v = want["v"].combine_chunks()
    e = po.binary("*", v, po.binary("-", pa.array([decimal.Decimal(1)], type=pa.decimal128(20, 0)), v, l_scalar=True))
    assert got["c1"].combine_chunks().equals(e)
    assert got["c2"].combine_chunks().equals(want["k"].combine_chunks().cast(pa.float64()))


def test_filter_rejects_non_boolean_predicate(ctx, task_ctx):
    import dfgpu
    from dfgpu import physical_plan as ops
    plan = ops.FilterExec(ops.Column("k", 0), source(ctx, [table(10)]))
    with pytest.raises(dfgpu.DfgpuError):
        list(plan.execute(0, task_ctx))


@pytest.mark.parametrize("partitions", [1, 3])
def test_aggregate_partial_repartition_final_vs_single(ctx, task_ctx, partitions):
    """Two-phase plan (Partial -> RepartitionExec Hash(keys) -> FinalPartitioned, physical_planner.rs:802-850) == Single == oracle."""
    import dfgpu
    from dfgpu import physical_plan as ops
    tabs = [table(4000), table(2500), table(1), table(3000)]
    C = ops.Column
    F = ops.Field
    aggs = [ops.AggregateFunctionExpr("SUM", C("v", 2), "SUM(v)", input_field=F("v", dfgpu.capi.DECIMAL128, 15, 2)),
            ops.AggregateFunctionExpr("AVG", C("v", 2), "AVG(v)", input_field=F("v", dfgpu.capi.DECIMAL128, 15, 2)),
            ops.AggregateFunctionExpr("COUNT", C("f", 3), "COUNT(f)"), ops.AggregateFunctionExpr("COUNT", None, "COUNT(*)"),
            ops.AggregateFunctionExpr("MIN", C("d", 1), "MIN(d)", input_field=F("d", dfgpu.capi.DATE32)),
            ops.AggregateFunctionExpr("MAX", C("f", 3), "MAX(f)", input_field=F("f", dfgpu.capi.FLOAT64)),
            ops.AggregateFunctionExpr("AVG", C("f", 3), "AVG(f)", input_field=F("f", dfgpu.capi.FLOAT64))]
    gby = [(C("k", 0), "k"), (C("s", 4), "s")]
    single = ops.AggregateExec("Single", gby, aggs, ops.CoalescePartitionsExec(source(ctx, tabs, partitions)))
    partial = ops.AggregateExec("Partial", gby, aggs, source(ctx, tabs, partitions))
    rep = ops.RepartitionExec(partial, ops.Partitioning.Hash([C("k", 0), C("s", 1)], 4))
    final = ops.AggregateExec("FinalPartitioned", [(C("k", 0), "k"), (C("s", 1), "s")], aggs, rep)
    a, b = collect_table(single, task_ctx), collect_table(final, task_ctx)
    ra, rb = sort_rows(rows_of([a[c] for c in a.column_names])), sort_rows(rows_of([b[c] for c in b.column_names]))
    assert len(ra) == len(rb)
    for x, y in zip(ra, rb):
        assert x[:6] == y[:6] and x[7] == y[7]                                   # exact: keys, decimal SUM/AVG, counts, MIN, MAX
        assert (x[8] is None and y[8] is None) or abs(x[8] - y[8]) <= 1e-9 * max(1.0, abs(x[8]))      # Float64 AVG: 1e-9 relative
    # oracle: same groups and SUM/COUNT values
    whole = pa.concat_tables(tabs)
    og = po.Groups([pa.int64(), pa.utf8()])
    ids = og.intern([whole["k"], whole["s"]])
    osum = po.Acc("SUM", pa.decimal128(15, 2)); osum.update_batch(whole["v"], ids, None, len(og))
    ocnt = po.Acc("COUNT", pa.int64()); ocnt.update_batch(None, ids, None, len(og))
    want = sort_rows(rows_of(og.emit() + [osum.evaluate(), ocnt.evaluate()]))
    assert [r[:3] + [r[5]] for r in ra] == want
    # Single mode on one partition emits groups in first-seen order like the reference (primitive.rs:137-141)
    if partitions == 1:
        first = rows_of([a["c0"], a["c1"]])
        assert first == rows_of(og.emit())


def test_sort_exec_and_fetch(ctx, task_ctx):
    from dfgpu import physical_plan as ops
    tabs = [table(3000), table(2000)]
    C = ops.Column
    sort = ops.SortExec([ops.PhysicalSortExpr(C("k", 0), descending=True, nulls_first=False), ops.PhysicalSortExpr(C("f", 3), descending=False, nulls_first=True)], source(ctx, tabs))
    got = collect_table(sort, task_ctx)
    whole = pa.concat_tables(tabs)
    idx = po.lexsort_to_indices([whole["k"], whole["f"]], [True, False], [False, True])
    want = whole.take(pa.array(idx))
    for i, name in enumerate(["k", "d", "v", "f", "s"]):
        assert got[f"c{i}"].combine_chunks().equals(want[name].combine_chunks())
    top = collect_table(ops.SortExec([ops.PhysicalSortExpr(C("k", 0), True, False), ops.PhysicalSortExpr(C("f", 3), False, True)], source(ctx, tabs), fetch=17), task_ctx)
    assert top.num_rows == 17 and top["c0"].combine_chunks().equals(want["k"].combine_chunks().slice(0, 17))


def test_repartition_conserves_rows_and_routes_by_hash(ctx, task_ctx):
    """repartition/mod.rs:952-1031 (many_to_many etc.): every input row appears exactly once; equal keys share a partition."""
    from dfgpu import physical_plan as ops
    tabs = [table(2000, 0.0), table(3000, 0.0), table(50, 0.0)]
    rep = ops.RepartitionExec(source(ctx, tabs, 3), ops.Partitioning.Hash([ops.Column("k", 0)], 5))
    seen, total = {}, 0
    for p in range(5):
        for b in rep.execute(p, task_ctx):
            ks = b.columns[0].to_arrow().to_pylist()
            total += len(ks)
            for k in set(ks):
                assert seen.setdefault(k, p) == p
    assert total == sum(t.num_rows for t in tabs)
    rr = ops.RepartitionExec(source(ctx, tabs, 1), ops.Partitioning.RoundRobinBatch(2))
    assert sum(b.num_rows for p in range(2) for b in rr.execute(p, task_ctx)) == total


def test_coalesce_batches(ctx, task_ctx):
    """coalesce_batches.rs tests: small batches are concatenated up to target_batch_size, order preserved."""
    from dfgpu import physical_plan as ops
    tabs = [table(8) for _ in range(10)]
    co = ops.CoalesceBatchesExec(source(ctx, tabs), 21)
    sizes = [b.num_rows for b in co.execute(0, task_ctx)]
    assert sizes == [24, 24, 24, 8]
    got = collect_table(co, task_ctx)
    assert got["c0"].combine_chunks().equals(pa.concat_tables(tabs)["k"].combine_chunks())


def test_with_fresh_state_reexecutes_from_scratch(ctx, task_ctx):
    """≙ ExecutionPlan::with_new_children (physical-plan/src/lib.rs:198-201): a copy of the plan without the OnceAsync build side /
    pulled RepartitionExec input gives the same rows again, any number of times, and leaves the template usable."""
    import pyarrow as pa
    from dfgpu import physical_plan as ops
    rng = np.random.default_rng(5)
    l = pa.table({"k": pa.array(rng.integers(0, 500, 3000)), "v": pa.array(rng.integers(0, 10**6, 3000))})
    r = pa.table({"k": pa.array(rng.integers(0, 500, 9000)), "w": pa.array(rng.integers(0, 10**6, 9000))})
    mk = lambda t: ops.MemoryExec([[ops.batch_from_arrow(ctx, t)]], ops.batch_from_arrow(ctx, t).schema)
    join = ops.HashJoinExec(mk(l), ops.RepartitionExec(mk(r), ops.Partitioning.Hash([ops.Column("k", 0)], 3)), [(ops.Column("k", 0), ops.Column("k", 0))], None, "Inner", "CollectLeft")
    plan = ops.SortExec([ops.PhysicalSortExpr(ops.Column("v", 1), False, False), ops.PhysicalSortExpr(ops.Column("w", 3), False, False)], ops.CoalescePartitionsExec(join))
    rows = lambda p: [c.to_arrow().to_pylist() for b in ops.collect(p, task_ctx) for c in [b.columns[1], b.columns[3]]]
    first = rows(plan)
    assert len(first[0]) > 0
    for _ in range(3):
        assert rows(ops.with_fresh_state(plan)) == first


# SortPreservingMergeExec known answers transcribed from physical-plan/src/sorts/sort_preserving_merge.rs `mod tests`
# (sort key (b, c), default SortOptions; the TimestampNanosecond column c is carried as its Int64 payload):
SPM_B1 = {"a": [1, 2, 7, 9, 3], "c": [8, 7, 6, 5, 8]}
SPM_CASES = [
    ("test_merge_interleave :283-328", dict(SPM_B1, b=["a", "c", "e", "g", "j"]), {"a": [10, 20, 70, 90, 30], "b": ["b", "d", "f", "h", "j"], "c": [4, 6, 2, 2, 6]},
     [(1, "a", 8), (10, "b", 4), (2, "c", 7), (20, "d", 6), (7, "e", 6), (70, "f", 2), (9, "g", 5), (90, "h", 2), (30, "j", 6), (3, "j", 8)]),
    ("test_merge_some_overlap :350-395", dict(SPM_B1, b=["a", "b", "c", "d", "e"]), {"a": [70, 90, 30, 100, 110], "b": ["c", "d", "e", "f", "g"], "c": [4, 6, 2, 2, 6]},
     [(1, "a", 8), (2, "b", 7), (70, "c", 4), (7, "c", 6), (9, "d", 5), (90, "d", 6), (30, "e", 2), (3, "e", 8), (100, "f", 2), (110, "g", 6)]),
    ("test_merge_no_overlap :398-443", dict(SPM_B1, b=["a", "b", "c", "d", "e"]), {"a": [10, 20, 70, 90, 30], "b": ["f", "g", "h", "i", "j"], "c": [4, 6, 2, 2, 6]},
     [(1, "a", 8), (2, "b", 7), (7, "c", 6), (9, "d", 5), (3, "e", 8), (10, "f", 4), (20, "g", 6), (70, "h", 2), (90, "i", 2), (30, "j", 6)]),
]


def _spm_table(d):
    return pa.table({"a": pa.array(d["a"], type=pa.int32()), "b": pa.array(d["b"], type=pa.utf8()), "c": pa.array(d["c"], type=pa.int64())})


@pytest.mark.parametrize("ref,b1,b2,expected", SPM_CASES, ids=[c[0].split()[0] for c in SPM_CASES])
def test_sort_preserving_merge_reference_known_answers(ctx, task_ctx, ref, b1, b2, expected):
    from dfgpu import physical_plan as ops
    bs = [ops.batch_from_arrow(ctx, _spm_table(b)) for b in (b1, b2)]
    src = ops.MemoryExec([[bs[0]], [bs[1]]], bs[0].schema)
    keys = [ops.PhysicalSortExpr(ops.Column("b", 1), False, True), ops.PhysicalSortExpr(ops.Column("c", 2), False, True)]
    merge = ops.SortPreservingMergeExec(keys, src)
    assert merge.output_partitioning().partition_count() == 1
    out = ops.collect(merge, task_ctx)
    got = [r for b in out for r in zip(*[c.to_arrow().to_pylist() for c in b.columns])]
    assert got == expected, ref


def test_sort_preserving_merge_is_stable_across_partitions_and_applies_fetch(ctx, task_ctx):
    """test_stable_sort (sort_preserving_merge.rs:947-1020): 10 partitions of (batch_number, ["A", "B"]) merged on `value` only
    come out in partition order within equal values; with_fetch keeps the first rows of that order; one input partition is
    passed through; no sort expressions is the reference's Internal error."""
    import dfgpu
    from dfgpu import physical_plan as ops
    parts = [[ops.batch_from_arrow(ctx, pa.table({"batch_number": pa.array([i, i], type=pa.int32()), "value": pa.array(["A", "B"])}))] for i in range(10)]
    src = ops.MemoryExec(parts, parts[0][0].schema)
    key = [ops.PhysicalSortExpr(ops.Column("value", 1), False, True)]
    rows = lambda plan: [r for b in ops.collect(plan, task_ctx) for r in zip(*[c.to_arrow().to_pylist() for c in b.columns])]
    want = [(i, "A") for i in range(10)] + [(i, "B") for i in range(10)]
    assert rows(ops.SortPreservingMergeExec(key, src)) == want
    assert rows(ops.SortPreservingMergeExec(key, src, fetch=13)) == want[:13]
    single = ops.MemoryExec([parts[3]], parts[0][0].schema)
    assert rows(ops.SortPreservingMergeExec(key, single)) == [(3, "A"), (3, "B")]
    with pytest.raises(dfgpu.DfgpuError) as e:
        rows(ops.SortPreservingMergeExec([], src))
    assert "Sort expressions cannot be empty for streaming merge" in str(e.value)


def test_projection_over_dense_selection_does_not_compact_and_dropped_rows_cannot_raise(ctx, task_ctx):
    """FilterExec -> ProjectionExec with a dense selection: the device evaluates `a / b` and a Decimal128 product over the FULL columns
    with the selection as row selection (include/dfgpu.h dfgpu_ctx_set_row_selection) -- rows the filter dropped (b = 0, an
    overflowing decimal) must not raise, the kept rows' values equal the reference's compact-then-evaluate order; a kept row that
    divides by zero still raises the reference's error."""
    import dfgpu
    from dfgpu import physical_plan as ops
    n = 4000
    a = RNG.integers(-10**6, 10**6, n).astype(np.int64)
    b = RNG.integers(1, 50, n).astype(np.int64)
    keep = RNG.random(n) < 0.8
    b[~keep] = 0                                              # every dropped row would divide by zero
    big = decimal.Decimal(10**37)
    d = [big if not k else decimal.Decimal(int(v)) for k, v in zip(keep, RNG.integers(1, 1000, n))]        # dropped rows would overflow d * d
    t = pa.table({"a": pa.array(a), "b": pa.array(b), "keep": pa.array(keep), "d": pa.array(d, type=pa.decimal128(38, 0))})
    C, B = ops.Column, ops.BinaryExpr
    src = ops.MemoryExec([[ops.batch_from_arrow(ctx, t)]], ops.batch_from_arrow(ctx, t).schema)
    plan = ops.ProjectionExec([(B(C("a", 0), "/", C("b", 1)), "q"), (B(C("d", 3), "*", C("d", 3)), "dd"), (C("a", 0), "a")], ops.FilterExec(C("keep", 2), src))
    out = ops.collect(plan, task_ctx)
    got = pa.Table.from_arrays([pa.concat_arrays([x.columns[i].to_arrow() for x in out]) for i in range(3)], names=["q", "dd", "a"])
    kt = t.filter(pa.array(keep))
    assert got["a"].to_pylist() == kt["a"].to_pylist()
    assert got["q"].to_pylist() == po.binary("/", kt["a"].combine_chunks(), kt["b"].combine_chunks()).to_pylist()
    assert got["dd"].combine_chunks().equals(po.binary("*", kt["d"].combine_chunks(), kt["d"].combine_chunks()))
    b2 = b.copy(); b2[np.flatnonzero(keep)[7]] = 0              # a KEPT row divides by zero
    t2 = t.set_column(1, "b", pa.array(b2))
    src2 = ops.MemoryExec([[ops.batch_from_arrow(ctx, t2)]], ops.batch_from_arrow(ctx, t2).schema)
    with pytest.raises(dfgpu.DfgpuError) as e:
        ops.collect(ops.ProjectionExec([(B(C("a", 0), "/", C("b", 1)), "q")], ops.FilterExec(C("keep", 2), src2)), task_ctx)
    assert "Divide by zero" in str(e.value)


@pytest.mark.parametrize("money", ["decimal", "float64", "decimal_nulls", "int64"])
@pytest.mark.parametrize("shape", ["x*(1-y)", "(1+y)*x", "x-(y/2)", "(y-1)/x"])
def test_fused_two_level_arithmetic_equals_node_by_node(ctx, task_ctx, money, shape):
    """BinaryExpr over `x op (literal op2 y)` (TPC-H revenue / charge expressions) runs as ONE device pass (dfgpu_binary_fused2) for
    Decimal128 / Float64 columns without NULLs: values and result TYPE must equal the oracle's node-by-node evaluation, in all four
    operand orders; nullable or integer operands take the node-by-node path through the same code."""
    from dfgpu import physical_plan as ops
    n = 5000
    if money.startswith("decimal"):
        mk = lambda lo, hi: pa.array([None if (money == "decimal_nulls" and RNG.random() < 0.1) else decimal.Decimal(int(v)).scaleb(-2) for v in RNG.integers(lo, hi, n)], type=pa.decimal128(15, 2))
        x, y = mk(1, 10**9), mk(1, 100)
        lit = lambda v: (ops.Literal(decimal.Decimal(v), pa.decimal128(20, 0)), pa.array([decimal.Decimal(v)], type=pa.decimal128(20, 0)))
    elif money == "float64":
        x, y = pa.array(RNG.normal(size=n) * 1000 + 5), pa.array(RNG.random(n) + 0.5)
        lit = lambda v: (ops.Literal(float(v), pa.float64()), pa.array([float(v)]))
    else:
        x, y = pa.array(RNG.integers(1, 10**6, n).astype(np.int64)), pa.array(RNG.integers(3, 100, n).astype(np.int64))
        lit = lambda v: (ops.Literal(int(v), pa.int64()), pa.array([int(v)], type=pa.int64()))
    C, B = ops.Column, ops.BinaryExpr
    one, one_a = lit(1); two, two_a = lit(2)
    expr, want = {
        "x*(1-y)": (B(C("x", 0), "*", B(one, "-", C("y", 1))), lambda: po.binary("*", x, po.binary("-", one_a, y, l_scalar=True))),
        "(1+y)*x": (B(B(one, "+", C("y", 1)), "*", C("x", 0)), lambda: po.binary("*", po.binary("+", one_a, y, l_scalar=True), x)),
        "x-(y/2)": (B(C("x", 0), "-", B(C("y", 1), "/", two)), lambda: po.binary("-", x, po.binary("/", y, two_a, r_scalar=True))),
        "(y-1)/x": (B(B(C("y", 1), "-", one), "/", C("x", 0)), lambda: po.binary("/", po.binary("-", y, one_a, r_scalar=True), x)),
    }[shape]
    t = pa.table({"x": x, "y": y})
    b = ops.batch_from_arrow(ctx, t)
    out = ops.collect(ops.ProjectionExec([(expr, "e")], ops.MemoryExec([[b]], b.schema)), task_ctx)
    got = pa.concat_arrays([o.columns[0].to_arrow() for o in out])
    w = want()
    assert got.type == w.type, f"{got.type} vs {w.type}"
    if pa.types.is_floating(w.type):
        assert np.array_equal(np.asarray(got).view(np.uint64), np.asarray(w).view(np.uint64))          # same IEEE operations in the same order
    else:
        assert got.equals(w)


def test_lent_device_memory_outlives_its_python_owner_until_the_plan_is_gone(ctx, task_ctx):
    """dfgpu_array_wrap_device_owned: a torch tensor wrapped zero-copy stays referenced by the library while any array, slice or C++ plan node
    (MemoryExec keeps raw device pointers) refers to it -- the Python objects may go first -- and is released when the last of them is gone."""
    import gc
    import torch
    from dfgpu import capi, device, physical_plan as ops
    before = device.lent_memory_owners()
    t = torch.arange(200000, dtype=torch.int64, device="cuda")
    arr = ctx.wrap_tensor(t, capi.INT64)
    batch = ops.RecordBatch.from_arrays(ctx, ["x"], [arr])
    plan = ops.FilterExec(ops.BinaryExpr(ops.Column("x", 0), "<", ops.Literal(1000, pa.int64())), ops.MemoryExec([[batch]], batch.schema))
    plan.handle(task_ctx)                                  # the C++ nodes exist now
    assert device.lent_memory_owners() == before + 1
    del t, arr, batch
    gc.collect(); torch.cuda.empty_cache()
    junk = torch.full((200000,), -1, dtype=torch.int64, device="cuda")       # would reuse the freed block if the tensor had died
    rows = sum(b.materialize().num_rows for b in plan.execute(0, task_ctx))
    assert rows == 1000
    assert device.lent_memory_owners() == before + 1
    del plan, junk
    gc.collect()
    assert device.lent_memory_owners() == before


@pytest.mark.parametrize("fetch", [None, 0, 1, 777, 10**9])
@pytest.mark.parametrize("budget", [64, 1000, 50_000])
def test_sort_preserving_merge_in_bounded_steps(ctx, task_ctx, budget, fetch):
    """The k-way merge with `spm_merge_rows` far below the input (sorts/merge.rs:38-110 holds one batch per stream; here one chunk per stream): 7 sorted partitions of
    uneven length in batches of uneven size, one of them empty, keys with long runs of ties across and inside partitions, NULL keys, a descending second key.  The output is
    the rows of the stable sort of the partitions concatenated in partition order -- ties: lower partition first, then arrival order (the `src` column shows both) -- which is
    what the merge's tie rule (lower stream index) produces, checked against an independent numpy model; with the default budget the same plan takes one step."""
    from dfgpu import physical_plan as ops
    rng = np.random.default_rng(budget)
    parts, model = [], []
    for p, n in enumerate([5000, 1, 0, 12345, 300, 7000, 2500]):
        k = np.sort(rng.integers(0, 40, n)); null = rng.random(n) < 0.03
        v = rng.integers(0, 5, n)
        # sorted on (k ASC NULLS FIRST, v DESC): NULL keys first, then by k, v descending inside a key
        o = np.lexsort((-v, k, ~null)); k, v, null = k[o], v[o], null[o]
        t = pa.table({"k": pa.array(k, mask=null), "v": pa.array(v), "src": pa.array(np.arange(n) + p * 10**6)})
        cuts = [0] + sorted(set(rng.integers(0, n + 1, 6).tolist())) + [n] if n else [0, 0]
        parts.append([t.slice(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])])
        model += [((0, 0) if null[i] else (1, int(k[i])), -int(v[i]), p, i) for i in range(n)]
    schema = ops.batch_from_arrow(ctx, parts[0][0]).schema
    src = ops.MemoryExec([[ops.batch_from_arrow(ctx, b) for b in bs] for bs in parts], schema)
    keys = [ops.PhysicalSortExpr(ops.Column("k", 0), False, True), ops.PhysicalSortExpr(ops.Column("v", 1), True, False)]
    want = [p * 10**6 + i for _, _, p, i in sorted(model)]
    if fetch is not None:
        want = want[:fetch]
    ctx.set_option("spm_merge_rows", budget)
    try:
        out = [b.to_arrow() for b in ops.SortPreservingMergeExec(keys, src, fetch=fetch).execute(0, task_ctx)]
    finally:
        ctx.set_option("spm_merge_rows", 1 << 25)
    got = pa.concat_tables(out)["src"].to_pylist() if out else []
    assert got == want
    if fetch is None and budget < 50_000:
        assert len(out) > 3, "several merge steps"
    one = pa.concat_tables([b.to_arrow() for b in ops.SortPreservingMergeExec(keys, src, fetch=fetch).execute(0, task_ctx)]) if want else None
    assert (one["src"].to_pylist() if one is not None else []) == want


@pytest.mark.parametrize("shape", ["every_probe_row_matches", "key_columns_only", "partial_match_with_payload"])
def test_inner_join_looks_build_rows_up_when_a_build_column_is_read(ctx, task_ctx, shape):
    """HashJoinExec Inner over a unique rank-indexed build (dfgpu_join_probe_deferred / dfgpu_join_lookup): the build rows of the matched pairs are computed when a
    build-side column is read, for the rows still wanted by then -- left for later when every probe row matched or when only (aliased) key columns leave the build side,
    at once otherwise.  A FilterExec that keeps 1 row in 500 sits above the join; rows equal the eager path's (option off) and pyarrow's join; the profile shows over how
    many rows the lookup ran."""
    from dfgpu import physical_plan as ops
    rng = np.random.default_rng(len(shape))
    nb, npr = 50_000, 400_000
    bk = np.arange(nb, dtype=np.int64) * 3 + 7                                   # sorted, unique: the rank index
    build = pa.table({"k": pa.array(bk), "pay": pa.array(rng.integers(0, 10**6, nb)), "s": pa.array([f"b{i % 97}" for i in range(nb)])})
    if shape == "key_columns_only":
        build = build.select(["k"])
    pk = bk[rng.integers(0, nb, npr)] if shape != "partial_match_with_payload" else np.where(rng.random(npr) < 0.4, bk[rng.integers(0, nb, npr)], -5)
    probe = pa.table({"pk": pa.array(pk), "v": pa.array(rng.integers(0, 1000, npr))})
    mk = lambda t: (lambda b: ops.MemoryExec([[b]], b.schema))(ops.batch_from_arrow(ctx, t))
    C = ops.Column

    def run():
        join = ops.HashJoinExec(mk(build), mk(probe), [(C("k", 0), C("pk", 0))], None, "Inner", "CollectLeft")
        nbc = build.num_columns
        plan = ops.FilterExec(ops.BinaryExpr(ops.BinaryExpr(C("v", nbc + 1), "%", ops.Literal(500, pa.int64())), "=", ops.Literal(3, pa.int64())), join)
        ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read()
        try:
            out = pa.concat_tables([b.to_arrow() for b in plan.execute(0, task_ctx)])          # probe batches above batch_size are answered in one piece (not in reference-sized chunks)
            prof = ctx.profile_read()
        finally:
            ctx.profile_enable(False)
        return out, prof

    lazy, prof = run()
    ctx.set_option("join_lazy_build_rows", 0)
    try:
        eager, prof0 = run()
    finally:
        ctx.set_option("join_lazy_build_rows", 1)
    assert lazy.equals(eager)
    keep = np.isin(pk, bk) & (probe["v"].to_numpy() % 500 == 3)                 # the join's rows that pass the filter, computed directly
    assert lazy.num_rows == int(keep.sum()) and sorted(lazy["v"].to_pylist()) == sorted(probe["v"].to_numpy()[keep].tolist())
    assert sorted(lazy["pk"].to_pylist()) == sorted(pk[keep].tolist()) and lazy["k"].to_pylist() == lazy["pk"].to_pylist()
    if "pay" in build.column_names:
        pay_of = dict(zip(bk.tolist(), build["pay"].to_pylist()))
        assert lazy["pay"].to_pylist() == [pay_of[k] for k in lazy["k"].to_pylist()]
    assert "k_probe_lookup_rank" in prof0                                       # eager: every matched pair
    if shape == "key_columns_only":
        assert "k_probe_lookup_rank" not in prof                               # nobody ever asks for a build row
    else:
        assert "k_probe_lookup_rank" in prof


@pytest.mark.parametrize("above", ["hash_right", "hash_full", "smj_left", "smj_full"])
def test_lazy_build_columns_through_a_nullable_index(ctx, task_ctx, above):
    """A deferred Inner HashJoinExec output (build rows not looked up yet) that becomes the NULL-able side of an outer join above it: the outer join takes the lazy
    columns through indices with a validity buffer, and the unmatched rows must come out NULL in them -- the lookup has to keep the index array's NULLs.
    Rows equal the eager path's (join_lazy_build_rows = 0) and pyarrow's joins."""
    from dfgpu import physical_plan as ops
    rng = np.random.default_rng(5)
    nb, npr, nx = 20_000, 60_000, 30_000
    bk = np.arange(nb, dtype=np.int64) * 3 + 7
    build = pa.table({"k": pa.array(bk), "pay": pa.array(rng.integers(0, 10**6, nb)), "s": pa.array([f"b{i % 97}" for i in range(nb)])})
    pk = np.sort(bk[rng.integers(0, nb, npr)])                                   # every probe row matches: the lookup is left for later
    probe = pa.table({"pk": pa.array(pk), "v": pa.array(np.arange(npr, dtype=np.int64))})
    xk = np.sort(rng.integers(0, 3 * nb + 200, nx).astype(np.int64))             # about a third of these find a pk
    x = pa.table({"xk": pa.array(xk), "w": pa.array(rng.integers(0, 99, nx))})
    mk = lambda t: (lambda b: ops.MemoryExec([[b]], b.schema))(ops.batch_from_arrow(ctx, t))
    C = ops.Column

    def run():
        j1 = ops.HashJoinExec(mk(build), mk(probe), [(C("k", 0), C("pk", 0))], None, "Inner", "CollectLeft")          # k, pay, s, pk, v
        if above.startswith("hash"):
            plan = ops.HashJoinExec(j1, mk(x), [(C("pk", 3), C("xk", 0))], None, "Right" if above == "hash_right" else "Full", "CollectLeft")
        else:
            plan = ops.SortMergeJoinExec(mk(x), j1, [(C("xk", 0), C("pk", 3))], "Left" if above == "smj_left" else "Full")
        return pa.concat_tables([b.to_arrow() for b in plan.execute(0, task_ctx)])

    lazy = run()
    ctx.set_option("join_lazy_build_rows", 0)
    try:
        eager = run()
    finally:
        ctx.set_option("join_lazy_build_rows", 1)
    assert lazy.equals(eager)
    j1_arrow = probe.join(build, keys="pk", right_keys="k", join_type="inner", coalesce_keys=False)
    want = x.join(j1_arrow, keys="xk", right_keys="pk", join_type="left outer" if above in ("hash_right", "smj_left") else "full outer", coalesce_keys=False)
    cols = ["xk", "w", "pk", "v", "k", "pay", "s"]
    key = lambda r: tuple((v is None, v) for v in r)
    rows = lambda t: sorted(zip(*[t[c].to_pylist() for c in cols]), key=key)
    assert lazy.num_rows == want.num_rows and rows(lazy) == rows(want)
    unmatched = ~np.isin(xk, pk)
    assert unmatched.any() and lazy["pay"].null_count == int(unmatched.sum()) == lazy["s"].null_count == lazy["k"].null_count


@pytest.mark.parametrize("build_filter", [False, True])
def test_inner_join_of_key_columns_answers_with_a_selection_over_the_probe_batch(ctx, task_ctx, build_filter):
    """HashJoinExec Inner over a unique sorted build that contributes only its key column (dfgpu_join_probe_selection): the output is the probe batch under the match bits
    -- no index vector, no count read back -- and the next join builds straight from the probe side's base columns under that selection (TPC-H Q3: customer x orders ->
    build of the join with lineitem).  Rows equal the path with the option off and pyarrow's joins; the profile shows no compaction for the first join."""
    from dfgpu import physical_plan as ops
    rng = np.random.default_rng(17 + build_filter)
    nc, no, nl = 30_000, 200_000, 600_000
    cust = pa.table({"c_key": pa.array(np.arange(1, nc + 1, dtype=np.int64)), "seg": pa.array(rng.integers(0, 5, nc).astype(np.int32))})
    okey = (np.arange(no, dtype=np.int64) // 8) * 32 + np.arange(no, dtype=np.int64) % 8 + 1                 # sorted, sparse: TPC-H's order keys
    orders = pa.table({"o_key": pa.array(okey), "o_cust": pa.array(rng.integers(1, nc + nc // 2, no).astype(np.int64)), "o_date": pa.array(rng.integers(8000, 9000, no).astype(np.int32))})
    lkey = np.sort(okey[rng.integers(0, no, nl)])
    line = pa.table({"l_key": pa.array(lkey), "l_val": pa.array(rng.integers(0, 1000, nl).astype(np.int64))})
    mk = lambda t: (lambda b: ops.MemoryExec([[b]], b.schema))(ops.batch_from_arrow(ctx, t))
    C, L, B = ops.Column, ops.Literal, ops.BinaryExpr

    def run():
        c = mk(cust)
        if build_filter:
            c = ops.CoalesceBatchesExec(ops.FilterExec(B(C("seg", 1), "=", L(2, pa.int32())), c), 8192)
        c = ops.ProjectionExec([(C("c_key", 0), "c_key")], c)
        o = ops.CoalesceBatchesExec(ops.FilterExec(B(C("o_date", 2), "<", L(8600, pa.int32())), mk(orders)), 8192)
        j1 = ops.CoalesceBatchesExec(ops.HashJoinExec(c, o, [(C("c_key", 0), C("o_cust", 1))], None, "Inner", "CollectLeft"), 8192)       # c_key, o_key, o_cust, o_date
        p1 = ops.ProjectionExec([(C("o_key", 1), "o_key"), (C("o_date", 3), "o_date")], j1)
        j2 = ops.HashJoinExec(p1, mk(line), [(C("o_key", 0), C("l_key", 0))], None, "Inner", "CollectLeft")                                   # o_key, o_date, l_key, l_val
        ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read()
        try:
            out = pa.concat_tables([b.to_arrow() for b in j2.execute(0, task_ctx)])
            prof = ctx.profile_read()
        finally:
            ctx.profile_enable(False)
        return out, prof

    sel, prof = run()
    ctx.set_option("join_selection_output", 0)
    try:
        plain, prof0 = run()
    finally:
        ctx.set_option("join_selection_output", 1)
    assert sel.equals(plain)
    keep_c = cust.filter(pc.equal(cust["seg"], 2)) if build_filter else cust
    j1a = orders.filter(pc.less(orders["o_date"], 8600)).join(keep_c.select(["c_key"]), keys="o_cust", right_keys="c_key", join_type="inner")
    want = line.join(j1a.select(["o_key", "o_date"]), keys="l_key", right_keys="o_key", join_type="inner", coalesce_keys=False)
    assert sel.num_rows == want.num_rows and sel.num_rows > 1000
    rows = lambda t: sorted(zip(t["o_key"].to_pylist(), t["o_date"].to_pylist(), t["l_key"].to_pylist(), t["l_val"].to_pylist()))
    assert rows(sel) == rows(want)
    assert sel["l_key"].to_pylist() == sorted(sel["l_key"].to_pylist())             # probe order kept
    syncs = lambda p: sum(v[0] for k, v in p.items() if k.startswith("sync:"))
    assert syncs(prof) < syncs(prof0), (prof, prof0)                                 # the first join's count is gone, and the second build's


def test_having_over_a_clustered_group_by_reads_the_keys_of_the_groups_it_keeps(ctx, task_ctx):
    """AggregateExec(Single) over a clustered key -> FilterExec (HAVING) -> ProjectionExec: the group keys leave the aggregate as pending gathers (key column at the first row of
    each run), so only the kept groups' keys are gathered (TPC-H Q18).  Rows equal the path that stores the keys at once and pyarrow's group_by + filter."""
    from dfgpu import physical_plan as ops
    rng = np.random.default_rng(8)
    ng = 120_000
    reps = rng.integers(1, 8, ng)
    k = np.repeat(np.arange(ng, dtype=np.int64) * 5 + 2, reps)
    q = rng.integers(1, 51, len(k)).astype(np.int64)
    t = pa.table({"k": pa.array(k), "q": pa.array(q)})
    C, L, B = ops.Column, ops.Literal, ops.BinaryExpr

    def run():
        b = ops.batch_from_arrow(ctx, t)
        agg = ops.AggregateExec("Single", [(C("k", 0), "k")], [ops.AggregateFunctionExpr("SUM", C("q", 1), "s", input_field=ops.Field("q", dfgpu_capi().INT64))], ops.MemoryExec([[b]], b.schema))
        having = ops.CoalesceBatchesExec(ops.FilterExec(B(C("s", 1), ">", L(250, pa.int64())), agg), 8192)
        plan = ops.ProjectionExec([(C("k", 0), "k"), (C("s", 1), "s")], having)
        ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read()
        try:
            out = pa.concat_tables([x.to_arrow() for x in plan.execute(0, task_ctx)]); prof = ctx.profile_read()
        finally:
            ctx.profile_enable(False)
        return out, prof

    lazy, prof = run()
    ctx.set_option("group_lazy_keys", 0)
    try:
        eager, prof0 = run()
    finally:
        ctx.set_option("group_lazy_keys", 1)
    assert lazy.equals(eager)
    ref = t.group_by("k", use_threads=False).aggregate([("q", "sum")])
    ref = ref.filter(pc.greater(ref["q_sum"], 250))
    assert lazy.num_rows == ref.num_rows and lazy.num_rows > 10 and lazy["k"].to_pylist() == ref["k"].to_pylist() and lazy["s"].to_pylist() == ref["q_sum"].to_pylist()


def dfgpu_capi():
    from dfgpu import capi
    return capi
