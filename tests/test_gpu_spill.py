"""-m gpu: AggregateExec under a state budget (option agg_spill_state_bytes ≙ the MemoryPool reservation of row_hash.rs:664-771).  Reference behaviour restated:
non-Partial modes spill the key-sorted state and, at the end of input, re-aggregate the merge of the spills and the remaining state, which leaves the groups in KEY
order (update_merged_stream switches to GroupOrdering::Full); Partial mode instead emits whole batch_size multiples of its groups early (emit_early_if_necessary).
Like the reference's own spill tests (aggregates/mod.rs:1803, :1888: assert_batches_sorted_eq) results are compared as sets against pyarrow's group_by; the key order of
the spilled output and the early batches' sizes are checked on top."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

pytestmark = pytest.mark.gpu


def _table(n, seed, card):
    rng = np.random.default_rng(seed)
    return pa.table({"k": pa.array(rng.integers(0, card, n), mask=rng.random(n) < 0.01), "s": pa.array(np.array(["a", "bb", "", "ccc"], dtype=object)[rng.integers(0, 4, n)], pa.utf8()),
                     "v": pa.array(rng.integers(-100, 100, n)), "w": pa.array(rng.random(n), mask=rng.random(n) < 0.05)})


def _plan(ops, capi, ctx, batches, mode, keys, names, src=None):
    C, F = ops.Column, ops.Field
    scan = src if src is not None else ops.MemoryExec([[ops.batch_from_arrow(ctx, b) for b in batches]], ops.batch_from_arrow(ctx, batches[0]).schema)
    aggs = [ops.AggregateFunctionExpr("SUM", C("v", names.index("v")), "sv", input_field=F("v", capi.INT64)), ops.AggregateFunctionExpr("AVG", C("w", names.index("w")), "aw", input_field=F("w", capi.FLOAT64)),
            ops.AggregateFunctionExpr("COUNT", None, "c"), ops.AggregateFunctionExpr("MIN", C("v", names.index("v")), "mn", input_field=F("v", capi.INT64))]
    return ops.AggregateExec(mode, [(C(k, names.index(k)), k) for k in keys], aggs, scan)


def _check(out: pa.Table, t: pa.Table, keys):
    ref = t.group_by(keys, use_threads=False).aggregate([("v", "sum"), ("w", "mean"), ([], "count_all"), ("v", "min")])
    order = [(k, "ascending") for k in keys]
    a, b = out.sort_by(order), ref.sort_by(order)
    assert a.num_rows == b.num_rows
    for k in keys:
        assert a[k].combine_chunks().equals(b[k].combine_chunks()), k
    assert a["sv"].combine_chunks().equals(b["v_sum"].combine_chunks()) and a["c"].combine_chunks().equals(b["count_all"].combine_chunks()) and a["mn"].combine_chunks().equals(b["v_min"].combine_chunks())
    x, y = a["aw"].combine_chunks(), b["w_mean"].combine_chunks()
    assert x.is_valid().equals(y.is_valid()) and np.allclose(x.fill_null(0).to_numpy(zero_copy_only=False), y.fill_null(0).to_numpy(zero_copy_only=False), rtol=1e-9)


@pytest.mark.parametrize("keys", [["k"], ["s", "k"]])
@pytest.mark.parametrize("ranges", [1, 5, 16])
def test_single_mode_spills_and_merges_in_key_order(ctx, keys, ranges):
    from dfgpu import capi, physical_plan as ops
    t = _table(120_000, 3, 30_000)
    batches = [t.slice(o, 8000) for o in range(0, t.num_rows, 8000)]
    base = ctx.get_option("live_bytes")
    ctx.set_option("agg_spill_state_bytes", 256 << 10); ctx.set_option("agg_spill_ranges", ranges)
    try:
        out = [b.to_arrow() for b in _plan(ops, capi, ctx, batches, "Single", keys, t.column_names).execute(0, ops.TaskContext(ctx, 8192))]
    finally:
        ctx.set_option("agg_spill_state_bytes", 0); ctx.set_option("agg_spill_ranges", 16)
    whole = pa.concat_tables(out)
    _check(whole, t, keys)
    # the merged stream is key-ordered (NULLs first): a first-seen-order output of random keys would not be
    assert whole.equals(whole.sort_by([(k, "ascending") for k in keys], null_placement="at_start"))
    assert len(out) >= min(ranges, 2)
    del out, whole
    assert ctx.get_option("live_bytes") <= base + (1 << 20)


def test_final_mode_spills_partial_states(ctx):
    """Partial (no budget) -> Final under a budget: the Final stage spills the states it merged so far, AVG's (count, sum) pair included."""
    from dfgpu import capi, physical_plan as ops
    t = _table(100_000, 9, 20_000)
    batches = [t.slice(o, 5000) for o in range(0, t.num_rows, 5000)]
    names = t.column_names
    # one Partial per input batch (each its own partition), merged by Final
    parts = ops.MemoryExec([[ops.batch_from_arrow(ctx, b)] for b in batches], ops.batch_from_arrow(ctx, batches[0]).schema)
    partial = _plan(ops, capi, ctx, None, "Partial", ["k"], names, src=parts)
    C, F = ops.Column, ops.Field
    final = ops.AggregateExec("Final", [(C("k", 0), "k")], [ops.AggregateFunctionExpr("SUM", C("v", 2), "sv", input_field=F("v", capi.INT64)), ops.AggregateFunctionExpr("AVG", C("w", 3), "aw", input_field=F("w", capi.FLOAT64)),
                                                           ops.AggregateFunctionExpr("COUNT", None, "c"), ops.AggregateFunctionExpr("MIN", C("v", 2), "mn", input_field=F("v", capi.INT64))], partial)
    ctx.set_option("agg_spill_state_bytes", 128 << 10)
    try:
        out = pa.concat_tables([b.to_arrow() for b in final.execute(0, ops.TaskContext(ctx, 8192))])
    finally:
        ctx.set_option("agg_spill_state_bytes", 0)
    _check(out, t, ["k"])
    assert out.equals(out.sort_by([("k", "ascending")], null_placement="at_start"))


def test_partial_mode_emits_early_in_batch_size_multiples(ctx):
    from dfgpu import capi, physical_plan as ops
    t = _table(100_000, 11, 50_000)
    batches = [t.slice(o, 10_000) for o in range(0, t.num_rows, 10_000)]
    names = t.column_names
    partial = _plan(ops, capi, ctx, batches, "Partial", ["k"], names)
    ctx.set_option("agg_spill_state_bytes", 64 << 10)
    try:
        early = [b.to_arrow() for b in partial.execute(0, ops.TaskContext(ctx, 1024))]
    finally:
        ctx.set_option("agg_spill_state_bytes", 0)
    assert len(early) > 2 and all(b.num_rows % 1024 == 0 for b in early[:-1])
    # the early batches are partial states: a Final stage over them gives the answer
    C, F = ops.Column, ops.Field
    states = ops.MemoryExec([[ops.batch_from_arrow(ctx, b) for b in early]], ops.batch_from_arrow(ctx, early[0]).schema)
    final = ops.AggregateExec("Final", [(C("k", 0), "k")], [ops.AggregateFunctionExpr("SUM", C("v", 2), "sv", input_field=F("v", capi.INT64)), ops.AggregateFunctionExpr("AVG", C("w", 3), "aw", input_field=F("w", capi.FLOAT64)),
                                                           ops.AggregateFunctionExpr("COUNT", None, "c"), ops.AggregateFunctionExpr("MIN", C("v", 2), "mn", input_field=F("v", capi.INT64))], states)
    _check(pa.concat_tables([b.to_arrow() for b in final.execute(0, ops.TaskContext(ctx, 8192))]), t, ["k"])


# ------------------------------------------------------------------ SortExec under an input budget (option sort_spill_bytes ≙ ExternalSorter's reservation, sorts/sort.rs:283-313)
def _sort_table(n, seed):
    rng = np.random.default_rng(seed)
    words = np.array(["", "a", "ab", "b", "zebra", "Zebra", "ää"], dtype=object)
    return pa.table({"k": pa.array(rng.integers(0, 500, n), mask=rng.random(n) < 0.02), "s": pa.array(words[rng.integers(0, len(words), n)], pa.utf8(), mask=rng.random(n) < 0.05),
                     "f": pa.array(rng.random(n)), "row": pa.array(np.arange(n))})


@pytest.mark.parametrize("ranges", [1, 3, 16])
@pytest.mark.parametrize("keys", [[("k", False, True)], [("s", True, False), ("k", False, False)], [("f", True, True)]], ids=["int-asc", "utf8-desc-then-int", "float-desc"])
def test_sort_spills_sorted_runs_and_merges_them(ctx, keys, ranges):
    """The spilled sort returns exactly the rows, in exactly the order, of the in-memory sort (both are stable, so ties -- many here: 500 distinct k, 7 distinct s --
    keep their arrival order; the `row` column shows it); spill_count / spilled_rows are reported as the reference's SortExec metrics are (sort.rs:229-231); the device holds
    no more afterwards than before.  The reference's own spill test (sort.rs:1052-1100 test_sort_spill) checks the row count, the metrics and that the output is sorted."""
    from dfgpu import physical_plan as ops
    t = _sort_table(200_000, 9)
    batches = [t.slice(o, 8192) for o in range(0, t.num_rows, 8192)]
    names = t.column_names
    mk = lambda: ops.SortExec([ops.PhysicalSortExpr(ops.Column(k, names.index(k)), d, nf) for k, d, nf in keys],
                              ops.MemoryExec([[ops.batch_from_arrow(ctx, b) for b in batches]], ops.batch_from_arrow(ctx, batches[0]).schema))
    want = pa.concat_tables([b.to_arrow() for b in mk().execute(0, ops.TaskContext(ctx, 8192))])
    base = ctx.get_option("live_bytes")
    ctx.set_option("sort_spill_bytes", 1 << 20); ctx.set_option("sort_spill_ranges", ranges)
    try:
        plan = mk(); tc = ops.TaskContext(ctx, 8192)
        out = [b.to_arrow() for b in plan.execute(0, tc)]
        kv = [m for m in plan.metrics(tc) if m["name"] == "SortExec"][0]
    finally:
        ctx.set_option("sort_spill_bytes", 0); ctx.set_option("sort_spill_ranges", 16)
    got = pa.concat_tables(out)
    assert got.num_rows == t.num_rows
    assert got.equals(want), "rows and row order of the in-memory sort"
    assert len(out) >= min(ranges, 2)
    assert int(kv["spill_count"]) >= 5 and int(kv["spilled_rows"]) == t.num_rows and int(kv["spilled_bytes"]) > t.num_rows * 20
    del out, got, want, plan
    assert ctx.get_option("live_bytes") <= base + (1 << 20)


def test_sort_with_fetch_or_within_budget_does_not_spill(ctx):
    from dfgpu import physical_plan as ops
    t = _sort_table(50_000, 4)
    batches = [t.slice(o, 8192) for o in range(0, t.num_rows, 8192)]
    src = lambda: ops.MemoryExec([[ops.batch_from_arrow(ctx, b) for b in batches]], ops.batch_from_arrow(ctx, batches[0]).schema)
    ctx.set_option("sort_spill_bytes", 1 << 20)
    try:
        topk = ops.SortExec([ops.PhysicalSortExpr(ops.Column("f", 2), True, True)], src(), fetch=10)
        tc = ops.TaskContext(ctx, 8192)
        rows = pa.concat_tables([b.to_arrow() for b in topk.execute(0, tc)])
        assert rows.num_rows == 10 and topk.metrics(tc)[0]["spill_count"] == 0
        ctx.set_option("sort_spill_bytes", 1 << 30)
        whole = ops.SortExec([ops.PhysicalSortExpr(ops.Column("f", 2), True, True)], src())
        assert sum(b.num_rows for b in whole.execute(0, tc)) == t.num_rows and whole.metrics(tc)[0]["spill_count"] == 0
    finally:
        ctx.set_option("sort_spill_bytes", 0)


def test_sort_spill_reference_shape(ctx):
    """sorts/sort.rs:1051-1100 test_sort_spill: 100 partitions of make_partition(100) (Int32 i = 0..100) through CoalescePartitionsExec into a SortExec whose budget holds a
    few batches: 10 000 rows out, spill_count > 0, spilled_bytes > 0, the first value 0, the output sorted, all memory returned.  (The reference sees 2 output batches of
    its merge; here one batch per key range.)"""
    from dfgpu import physical_plan as ops
    part = pa.table({"i": pa.array(np.arange(100, dtype=np.int32))})
    b0 = ops.batch_from_arrow(ctx, part)
    src = ops.MemoryExec([[ops.batch_from_arrow(ctx, part)] for _ in range(100)], b0.schema)
    plan = ops.SortExec([ops.PhysicalSortExpr(ops.Column("i", 0), False, True)], ops.CoalescePartitionsExec(src))
    base = ctx.get_option("live_bytes")
    ctx.set_option("sort_spill_bytes", 12288); ctx.set_option("collect_metrics", 1)
    try:
        tc = ops.TaskContext(ctx, 8192)
        out = pa.concat_tables([b.to_arrow() for b in plan.execute(0, tc)])
        m = plan.metrics(tc)[0]
    finally:
        ctx.set_option("sort_spill_bytes", 0); ctx.set_option("collect_metrics", 0)
    v = out["i"].to_numpy()
    assert len(v) == 10000 and m["output_rows"] == 10000 and m["spill_count"] > 0 and m["spilled_bytes"] > 0
    assert v[0] == 0 and v[-1] == 99 and np.all(np.diff(v) >= 0) and np.array_equal(np.bincount(v), np.full(100, 100))
    del out, plan, src, b0
    assert ctx.get_option("live_bytes") <= base + (1 << 20)
