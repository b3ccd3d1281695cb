"""GroupOrdering (aggregates/order/): the oracle's restatement against hand-worked schedules (CPU), and AggregateExec's ordered modes on the device against it."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from oracle import pyoracle as po


def test_oracle_full_ordering_schedule():
    # full.rs doc example shape: all groups in front of the current (last) one leave after every batch
    b = lambda *v: [pa.array(list(v), pa.int64())]
    assert po.group_ordering_emits([b(1, 1, 2, 3), b(3, 3), b(3, 4, 5), b(5)]) == [2, 2, 1]          # {1,2} | - | {3,4} | - | end: {5}
    assert po.group_ordering_emits([b(7, 7, 7)]) == [1]
    assert po.group_ordering_emits([b(1), b(2), b(3)]) == [1, 1, 1]
    assert po.group_ordering_emits([]) == []


def test_oracle_partial_ordering_schedule():
    # sorted on key 0 only; groups (a, x): everything in front of the first group of the latest a-value leaves
    b = lambda a, x: [pa.array(a, pa.int64()), pa.array(x, pa.utf8())]
    got = po.group_ordering_emits([b([1, 1, 1, 2], ["p", "q", "p", "p"]), b([2, 2, 3], ["q", "p", "z"]), b([3, 3], ["y", "z"])], order_indices=[0])
    assert got == [2, 2, 2]            # (1,p),(1,q) | (2,p),(2,q) | end: (3,z),(3,y)
    assert po.group_ordering_emits([b([5, 5], ["a", "b"]), b([5], ["c"])], order_indices=[0]) == [3]


def _sorted_table(n, seed, nkeys):
    rng = np.random.default_rng(seed)
    a = np.sort(rng.integers(0, max(2, n // 40), n))
    cols = {"a": pa.array(a), "v": pa.array(rng.integers(-1000, 1000, n)), "w": pa.array(rng.random(n) * 100, mask=rng.random(n) < 0.1)}
    if nkeys == 2:
        cols["b"] = pa.array(np.array(["x", "y", "z", "long string key"], dtype=object)[rng.integers(0, 4, n)], pa.utf8())
    return pa.table(cols)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["Single", "Partial"])
@pytest.mark.parametrize("shape", ["sorted_one_key", "partially_sorted_two_keys", "sorted_two_keys"])
def test_ordered_aggregate_emits_finished_groups_per_batch(ctx, shape, mode):
    """AggregateExec with InputOrderMode::Sorted / PartiallySorted: the rows of every output batch equal the oracle's GroupOrdering schedule, their concatenation equals the
    Linear-mode output of the same plan (same group order, same values / states), and that equals pyarrow's group_by."""
    from dfgpu import capi, physical_plan as ops
    nkeys = 1 if shape == "sorted_one_key" else 2
    t = _sorted_table(6000, 5, nkeys)
    if shape == "sorted_two_keys":
        t = t.sort_by([("a", "ascending"), ("b", "ascending")])
    sizes = [1000, 1, 999, 2500, 1500]
    batches, off = [], 0
    for s in sizes:
        batches.append(t.slice(off, s)); off += s
    keys = ["a"] if nkeys == 1 else ["a", "b"]
    order = "Sorted" if shape != "partially_sorted_two_keys" else ("PartiallySorted", [0])
    C, F = ops.Column, ops.Field
    names = t.column_names

    def plan(order_mode):
        scan = ops.MemoryExec([[ops.batch_from_arrow(ctx, b) for b in batches]], ops.batch_from_arrow(ctx, batches[0]).schema)
        aggs = [ops.AggregateFunctionExpr("SUM", C("v", names.index("v")), "s", input_field=F("v", capi.INT64)), ops.AggregateFunctionExpr("AVG", C("w", names.index("w")), "m", input_field=F("w", capi.FLOAT64)),
                ops.AggregateFunctionExpr("COUNT", None, "c")]
        return ops.AggregateExec(mode, [(C(k, names.index(k)), k) for k in keys], aggs, scan, input_order_mode=order_mode)
    tc = ops.TaskContext(ctx, 8192)
    got = [b.to_arrow() for b in plan(order).execute(0, tc)]
    want_rows = po.group_ordering_emits([[b[k].combine_chunks() for k in keys] for b in batches], None if order == "Sorted" else [0])
    assert [b.num_rows for b in got] == want_rows
    assert len(got) > 3
    linear = pa.concat_tables([b.to_arrow() for b in plan("Linear").execute(0, tc)])
    whole = pa.concat_tables(got)
    assert whole.schema.names == linear.schema.names
    for name in whole.schema.names:
        a, b = whole[name].combine_chunks(), linear[name].combine_chunks()
        if pa.types.is_floating(a.type):
            assert np.allclose(a.fill_null(0).to_numpy(zero_copy_only=False), b.fill_null(0).to_numpy(zero_copy_only=False), rtol=1e-12) and a.is_valid().equals(b.is_valid()), name
        else:
            assert a.equals(b), name
    if mode == "Single":
        ref = t.group_by(keys, use_threads=False).aggregate([("v", "sum"), ("w", "mean"), ([], "count_all")]).sort_by([(k, "ascending") for k in keys])
        mine = whole.sort_by([(k, "ascending") for k in keys])
        for k in keys:
            assert mine[k].combine_chunks().equals(ref[k].combine_chunks())
        assert mine["s"].combine_chunks().equals(ref["v_sum"].combine_chunks()) and mine["c"].combine_chunks().equals(ref["count_all"].combine_chunks())
        assert np.allclose(mine["m"].combine_chunks().fill_null(-1).to_numpy(zero_copy_only=False), ref["w_mean"].combine_chunks().fill_null(-1).to_numpy(zero_copy_only=False), rtol=1e-12)


@pytest.mark.gpu
def test_ordered_aggregate_keeps_only_open_groups(ctx):
    """Sorted mode over many batches: HBM held by the group table stays bounded by one batch's groups (the point of GroupOrdering), and a FilterExec selection in
    front of the aggregate (rows dropped by a mask) does not disturb the schedule's result."""
    from dfgpu import capi, physical_plan as ops
    n, per = 200_000, 10_000
    a = np.arange(n) // 4
    t = pa.table({"a": pa.array(a), "v": pa.array(np.ones(n, dtype=np.int64)), "keep": pa.array((np.arange(n) % 3) != 0)})
    batches = [t.slice(o, per) for o in range(0, n, per)]
    C, F = ops.Column, ops.Field
    scan = ops.MemoryExec([[ops.batch_from_arrow(ctx, b) for b in batches]], ops.batch_from_arrow(ctx, batches[0]).schema)
    filt = ops.FilterExec(C("keep", 2), scan)
    agg = ops.AggregateExec("Single", [(C("a", 0), "a")], [ops.AggregateFunctionExpr("SUM", C("v", 1), "s", input_field=F("v", capi.INT64))], filt, input_order_mode="Sorted")
    out = [b.to_arrow() for b in agg.execute(0, ops.TaskContext(ctx, 8192))]
    assert len(out) == len(batches) + 1 or len(out) == len(batches)
    assert max(b.num_rows for b in out) <= per // 4 + 1
    whole = pa.concat_tables(out)
    sel = t.filter(t["keep"])
    ref = sel.group_by("a", use_threads=False).aggregate([("v", "sum")])
    assert whole["a"].combine_chunks().equals(ref["a"].combine_chunks()) and whole["s"].combine_chunks().equals(ref["v_sum"].combine_chunks())


@pytest.mark.gpu
def test_ordered_aggregate_edge_inputs(ctx):
    """Empty input, empty batches between full ones, a single group spanning every batch, one-row batches: the schedule still follows the oracle."""
    from dfgpu import capi, physical_plan as ops
    C, F = ops.Column, ops.Field
    sch_t = pa.table({"a": pa.array([], pa.int64()), "v": pa.array([], pa.int64())})

    def run(batches, order="Sorted"):
        scan = ops.MemoryExec([[ops.batch_from_arrow(ctx, b) for b in batches]], ops.batch_from_arrow(ctx, sch_t).schema)
        agg = ops.AggregateExec("Single", [(C("a", 0), "a")], [ops.AggregateFunctionExpr("SUM", C("v", 1), "s", input_field=F("v", capi.INT64))], scan, input_order_mode=order)
        return [b.to_arrow() for b in agg.execute(0, ops.TaskContext(ctx, 8192))]
    mk = lambda a: pa.table({"a": pa.array(a, pa.int64()), "v": pa.array([1] * len(a), pa.int64())})
    assert run([]) == [] and run([sch_t]) == []
    cases = [[mk([5, 5]), sch_t, mk([5]), mk([5, 5, 5])],                       # one group throughout: nothing leaves before the end
             [mk([1]), mk([2]), mk([3])], [mk([1, 2, 3]), sch_t, sch_t, mk([3, 4])], [mk([7])]]
    for batches in cases:
        got = run(batches)
        want = po.group_ordering_emits([[b["a"].combine_chunks()] for b in batches if b.num_rows])
        assert [b.num_rows for b in got] == want, (batches, want)
        whole = pa.concat_tables(got)
        ref = pa.concat_tables(batches).group_by("a", use_threads=False).aggregate([("v", "sum")])
        assert whole["a"].combine_chunks().equals(ref["a"].combine_chunks()) and whole["s"].combine_chunks().equals(ref["v_sum"].combine_chunks())
