"""-m gpu: GroupValues + GroupsAccumulator on device vs the CPU oracle: group ids in first-seen order (bit-exact),
integer / Decimal128 aggregates bit-exact, Float64 SUM/AVG within 1e-9 relative (north_star tolerance)."""
import decimal

import numpy as np
import pyarrow as pa
import pytest

from oracle import pyoracle as po
from test_gpu_core import rand_array
from test_gpu_join import keycols

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(5)
FLOAT_RTOL = 1e-9          # BASELINE.json north_star: "within 1e-9 relative for floating-point SUM/AVG"
KIND = {"SUM": 0, "AVG": 1, "COUNT": 2, "MIN": 3, "MAX": 4}


@pytest.mark.parametrize("kinds,card", [(["int64"], 50), (["int64"], 100000), (["int32", "utf8"], 30), (["utf8"], 500), (["dict"], 40),
                                        (["decimal", "int32", "float64"], 20), (["int64"], 3)])
def test_intern_first_seen_ids_and_emit(ctx, kinds, card):
    import dfgpu
    gv, og = dfgpu.GroupValues(ctx, len(kinds)), None
    for n in (5000, 1, 0, 20000, 777):          # several batches: ids keep growing in first-seen order (primitive.rs:137-141)
        cols = keycols(kinds, n, 0.1, card)
        if og is None:
            og = po.Groups([c.type for c in cols])
        ids = gv.intern([ctx.from_arrow(c) for c in cols]).to_numpy()
        assert np.array_equal(ids.astype(np.int64), og.intern(cols))
        assert len(gv) == len(og)
    for got, want in zip(gv.emit(), og.emit()):
        assert got.to_arrow().equals(want)
    assert gv.size() > 0


def test_intern_under_forced_collisions_and_mask(ctx):
    import dfgpu
    cols = keycols(["int64", "utf8"], 3000, 0.1, 12)
    ctx.set_option("force_hash_collisions", 1)
    try:
        gv = dfgpu.GroupValues(ctx, 2)
        ids = gv.intern([ctx.from_arrow(c) for c in cols]).to_numpy()
    finally:
        ctx.set_option("force_hash_collisions", 0)
    og = po.Groups([c.type for c in cols])
    assert np.array_equal(ids.astype(np.int64), og.intern(cols))
    m = RNG.random(3000) < 0.5
    gv2 = dfgpu.GroupValues(ctx, 2)
    ids2 = gv2.intern([ctx.from_arrow(c) for c in cols], mask=ctx.from_arrow(pa.array(m))).to_numpy()
    og2 = po.Groups([c.type for c in cols])
    want = og2.intern([c.filter(pa.array(m)) for c in cols])
    assert np.array_equal(ids2[m].astype(np.int64), want) and (ids2[~m] == 0xFFFFFFFF).all()


def test_clustered_keys_are_numbered_by_runs_like_the_hash_path(ctx):
    """GROUP BY over batches that arrive clustered on their keys (≙ GroupOrdering::Full): groups.hip numbers groups by runs without a
    hash table.  Ids must equal the oracle's first-seen ids across: a clustered batch, a second clustered batch that continues above
    the first, then a batch that breaks the order (falls back: the numbered groups are hashed, the table is built), then more rows."""
    import dfgpu
    def batch(lo, n, width):
        k0 = np.sort(RNG.integers(lo, lo + width, n)).astype(np.int64)
        return [pa.array(k0), pa.array((k0 * 7 % 13).astype(np.int32)), pa.array([None if v % 5 == 0 else f"s{v % 11}" for v in k0], type=pa.utf8())]
    b1, b2 = batch(0, 5000, 800), batch(1000, 7000, 900)
    b3 = [pa.array(RNG.integers(0, 2500, 3000).astype(np.int64))]
    b3 += [pa.array((np.asarray(b3[0]) * 7 % 13).astype(np.int32)), pa.array([None if v % 5 == 0 else f"s{v % 11}" for v in np.asarray(b3[0])], type=pa.utf8())]
    b4 = batch(5000, 100, 50)
    for runs in (1, 0):
        ctx.set_option("group_run_detection", runs)
        try:
            gv, og = dfgpu.GroupValues(ctx, 3), po.Groups([c.type for c in b1])
            for b in (b1, b2, b3, b4):
                got = gv.intern([ctx.from_arrow(c) for c in b]).to_numpy().astype(np.int64)
                assert np.array_equal(got, og.intern(b)), f"runs={runs}"
            assert len(gv) == len(og)
            for a, w in zip(gv.emit(), og.emit()):
                assert a.to_arrow().equals(w)
        finally:
            ctx.set_option("group_run_detection", 1)


def test_first_key_sorted_but_other_keys_alternating_is_not_a_run(ctx):
    """(a,1),(a,2),(a,1): sorted on the first column yet equal keys are not adjacent -- the run path must decline."""
    import dfgpu
    k0 = pa.array(np.repeat(np.arange(2000, dtype=np.int64), 3))
    k1 = pa.array(np.tile(np.array([1, 2, 1], dtype=np.int32), 2000))
    gv, og = dfgpu.GroupValues(ctx, 2), po.Groups([k0.type, k1.type])
    got = gv.intern([ctx.from_arrow(k0), ctx.from_arrow(k1)]).to_numpy().astype(np.int64)
    assert np.array_equal(got, og.intern([k0, k1])) and len(gv) == 4000


def test_dictionary_keys_through_canonical_ids_match_value_interning(ctx):
    """Dictionary key columns are interned as u32 ids of the distinct dictionary VALUES (groups.hip canon mode): a dictionary that
    repeats a value under two codes, holds a NULL value and NULL codes must give exactly the oracle's groups / first-seen ids / emitted
    keys -- over batches that share the dictionary (slices of one device array), then over a batch with ANOTHER dictionary (drops back
    to value keys, the numbered groups are re-hashed), with a plain Int64 second key column, and with the option switched off."""
    import dfgpu
    words = pa.array(["x", "y", None, "x", "z", "w", "y"], type=pa.utf8())          # codes 0/3 and 1/6 carry equal values; code 2 is a NULL value
    n = 9000
    codes = pa.array(RNG.integers(0, 7, n).astype(np.int32), mask=RNG.random(n) < 0.05)
    col = pa.DictionaryArray.from_arrays(codes, words)
    k2 = pa.array(RNG.integers(0, 4, n).astype(np.int64))
    other = pa.array([None if v % 7 == 0 else "xyzwq"[v % 5] for v in RNG.integers(0, 100, 3000)], type=pa.utf8()).dictionary_encode()
    other2 = pa.array(RNG.integers(0, 4, 3000).astype(np.int64))
    for canon in (1, 0):
        ctx.set_option("group_dictionary_canon", canon)
        try:
            dev, dev2 = ctx.from_arrow(col), ctx.from_arrow(k2)
            gv, og = dfgpu.GroupValues(ctx, 2), po.Groups([pa.utf8(), pa.int64()])
            for lo, hi in ((0, 4000), (4000, 4001), (4001, n)):
                got = gv.intern([dev.slice(lo, hi - lo), dev2.slice(lo, hi - lo)]).to_numpy().astype(np.int64)
                assert np.array_equal(got, og.intern([col.slice(lo, hi - lo), k2.slice(lo, hi - lo)])), f"canon={canon} rows {lo}:{hi}"
            got = gv.intern([ctx.from_arrow(other), ctx.from_arrow(other2)]).to_numpy().astype(np.int64)      # a different dictionary
            assert np.array_equal(got, og.intern([other, other2])), f"canon={canon} other dictionary"
            got = gv.intern([dev.slice(0, 500), dev2.slice(0, 500)]).to_numpy().astype(np.int64)
            assert np.array_equal(got, og.intern([col.slice(0, 500), k2.slice(0, 500)]))
            assert len(gv) == len(og)
            for a, w in zip(gv.emit(), og.emit()):
                assert a.to_arrow().equals(w)
        finally:
            ctx.set_option("group_dictionary_canon", 1)


def test_all_dictionary_keys_with_a_small_domain_use_the_dense_map(ctx):
    """Two dictionary key columns whose canonical domains multiply to <= 4096 (TPC-H Q1's l_returnflag, l_linestatus): groups.hip indexes
    a dense composite map (k_dense_first / k_dense_ids) instead of hashing.  Same first-seen ids / emitted keys as the oracle over batches
    sharing the dictionaries (with a fused selection mask), NULL codes and a repeated dictionary value included; a batch with other
    dictionaries then drops back to value keys."""
    import dfgpu
    w1 = pa.array(["A", "N", "R", "A", None], type=pa.utf8())                    # codes 0 and 3 carry the same value; code 4 is a NULL value
    w2 = pa.array(["F", "O"], type=pa.utf8())
    n = 20000
    c1 = pa.DictionaryArray.from_arrays(pa.array(RNG.integers(0, 5, n).astype(np.int8), mask=RNG.random(n) < 0.03), w1)
    c2 = pa.DictionaryArray.from_arrays(pa.array(RNG.integers(0, 2, n).astype(np.int8)), w2)
    o1 = pa.array([["N", "R", "X"][v] for v in RNG.integers(0, 3, 2000)], type=pa.utf8()).dictionary_encode()
    o2 = pa.array([["O", "F"][v] for v in RNG.integers(0, 2, 2000)], type=pa.utf8()).dictionary_encode()
    d1, d2 = ctx.from_arrow(c1), ctx.from_arrow(c2)
    gv, og = dfgpu.GroupValues(ctx, 2), po.Groups([pa.utf8(), pa.utf8()])
    m = RNG.random(6000) < 0.5
    ids = gv.intern([d1.slice(0, 6000), d2.slice(0, 6000)], mask=ctx.from_arrow(pa.array(m))).to_numpy()
    want = og.intern([c1.slice(0, 6000).filter(pa.array(m)), c2.slice(0, 6000).filter(pa.array(m))])
    assert np.array_equal(ids[m].astype(np.int64), want) and (ids[~m] == 0xFFFFFFFF).all()
    for lo, hi in ((6000, 6001), (6001, n)):
        got = gv.intern([d1.slice(lo, hi - lo), d2.slice(lo, hi - lo)]).to_numpy().astype(np.int64)
        assert np.array_equal(got, og.intern([c1.slice(lo, hi - lo), c2.slice(lo, hi - lo)]))
    got = gv.intern([ctx.from_arrow(o1), ctx.from_arrow(o2)]).to_numpy().astype(np.int64)             # other dictionaries
    assert np.array_equal(got, og.intern([o1, o2]))
    got = gv.intern([d1.slice(0, 100), d2.slice(0, 100)]).to_numpy().astype(np.int64)
    assert np.array_equal(got, og.intern([c1.slice(0, 100), c2.slice(0, 100)]))
    assert len(gv) == len(og)
    for a, w in zip(gv.emit(), og.emit()):
        assert a.to_arrow().equals(w)


def value_array(kind, n):
    if kind == "decimal":
        return pa.array([None if RNG.random() < 0.1 else decimal.Decimal(int(v)).scaleb(-2) for v in RNG.integers(-10**13, 10**13, n)], type=pa.decimal128(15, 2))
    return rand_array(kind, n, 0.1, RNG)


def check_equal(got, want, floats):
    assert got.type == want.type, f"{got.type} vs {want.type}"
    assert np.array_equal(np.asarray(got.is_null()), np.asarray(want.is_null()))
    if floats:
        g, w = np.asarray(got.fill_null(0), dtype=np.float64), np.asarray(want.fill_null(0), dtype=np.float64)
        assert np.allclose(g, w, rtol=FLOAT_RTOL, atol=1e-9 * np.abs(w).max() if len(w) else 0)
    else:
        assert got.equals(want)


AGG_CASES = [("SUM", "int64"), ("SUM", "int32"), ("SUM", "uint64"), ("SUM", "float64"), ("SUM", "decimal"), ("AVG", "float64"), ("AVG", "decimal"),
             ("COUNT", "int64"), ("COUNT", "utf8"), ("MIN", "int32"), ("MAX", "int64"), ("MIN", "float64"), ("MAX", "float32"), ("MIN", "decimal"), ("MAX", "decimal"),
             ("MIN", "uint8"), ("MAX", "date32")]


@pytest.mark.parametrize("fun,kind", AGG_CASES, ids=[f"{f}-{k}" for f, k in AGG_CASES])
@pytest.mark.parametrize("ngroups", [4, 8, 9, 300, 5000])
def test_accumulators_update_evaluate_state_merge(ctx, fun, kind, ngroups):
    """Partial (update_batch over several batches, growing group count, opt_filter) -> state -> Final (merge_batch) -> evaluate."""
    import dfgpu
    first = value_array(kind, 10)
    f = dfgpu.operators.field_of_array("v", ctx.from_arrow(first))
    acc, oacc = dfgpu.GroupsAccumulator(ctx, KIND[fun], f.dtype, f.precision, f.scale), po.Acc(fun, first.type)
    total = 0
    for n, hi, use_filter in [(4000, ngroups // 2 + 1, False), (6000, ngroups, True), (10, ngroups, False)]:
        v = value_array(kind, n)
        g = RNG.integers(0, hi, n)
        total = max(total, hi)
        filt = pa.array(RNG.random(n) < 0.7, mask=RNG.random(n) < 0.05) if use_filter else None
        acc.update_batch(ctx.from_arrow(v), ctx.from_arrow(pa.array(g.astype(np.uint32))), ctx.from_arrow(filt) if filt is not None else None, total)
        oacc.update_batch(v, g, filt, total)
    floats = kind.startswith("float") and fun in ("SUM", "AVG")
    st, ost = acc.state(), oacc.state()
    assert len(st) == len(ost)
    for a, b in zip(st, ost):
        check_equal(a.to_arrow(), b, floats)
    check_equal(acc.evaluate().to_arrow(), oacc.evaluate(), floats)
    # Final: merge the partial state twice (two "partitions") under a permutation of group ids
    perm = RNG.permutation(total)
    fin, ofin = dfgpu.GroupsAccumulator(ctx, KIND[fun], f.dtype, f.precision, f.scale), po.Acc(fun, first.type)
    for _ in range(2):
        fin.merge_batch(st, ctx.from_arrow(pa.array(perm.astype(np.uint32))), None, total)
        ofin.merge_batch(ost, perm, None, total)
    check_equal(fin.evaluate().to_arrow(), ofin.evaluate(), floats)
    assert acc.size() > 0


@pytest.mark.parametrize("fun,kind", AGG_CASES, ids=[f"{f}-{k}" for f, k in AGG_CASES])
def test_accumulators_skewed_groups_use_the_lds_cache(ctx, fun, kind):
    """Heavy hitters + a long tail (Zipf(1.1) over 5000 groups, 300 K rows): more groups than the 1024-entry per-workgroup LDS cache
    of k_acc_cached, so hot groups accumulate in LDS while the tail falls through to the global atomics -- same results as the
    oracle's row-by-row loop (acc.hip k_acc_cached; prim_op.rs:101-109)."""
    import dfgpu
    n, ng = 300000, 5000
    v = value_array(kind, n)
    g = (RNG.zipf(1.1, n) % ng).astype(np.int64)
    f = dfgpu.operators.field_of_array("v", ctx.from_arrow(v.slice(0, 10)))
    acc, oacc = dfgpu.GroupsAccumulator(ctx, KIND[fun], f.dtype, f.precision, f.scale), po.Acc(fun, v.type)
    filt = pa.array(RNG.random(n) < 0.9)
    acc.update_batch(ctx.from_arrow(v), ctx.from_arrow(pa.array(g.astype(np.uint32))), ctx.from_arrow(filt), ng)
    oacc.update_batch(v, g, filt, ng)
    floats = kind.startswith("float") and fun in ("SUM", "AVG")
    for a, b in zip(acc.state(), oacc.state()):
        check_equal(a.to_arrow(), b, floats)
    check_equal(acc.evaluate().to_arrow(), oacc.evaluate(), floats)


@pytest.mark.parametrize("ngroups", [1, 6, 8, 9])
def test_update_batch_multi_equals_separate_updates(ctx, ngroups):
    """dfgpu_acc_update_batch_multi (all accumulators of a batch at once; SUM / AVG neighbours over <= 8 groups share a pass, SUM(x) and
    AVG(x) share the load of x) must leave every accumulator exactly as separate update_batch calls do -- TPC-H Q1's aggregate list over
    Float64 and Decimal128 columns with NULLs, a filter, and a COUNT(*) in between; 9 groups take the single-accumulator paths."""
    import dfgpu
    n = 30000
    g = RNG.integers(0, ngroups, n)
    f64 = lambda: pa.array(RNG.normal(size=n) * 100, mask=RNG.random(n) < 0.1)
    dec = lambda: pa.array([None if RNG.random() < 0.1 else decimal.Decimal(int(v)).scaleb(-2) for v in RNG.integers(-10**9, 10**9, n)], type=pa.decimal128(15, 2))
    x, y, d1, d2, i64 = f64(), f64(), dec(), dec(), pa.array(RNG.integers(-10**6, 10**6, n).astype(np.int64))
    filt = pa.array(RNG.random(n) < 0.8)
    spec = [("SUM", x), ("AVG", x), ("SUM", y), ("AVG", y), ("AVG", x), ("COUNT", None), ("SUM", d1), ("AVG", d1), ("SUM", d2), ("SUM", i64), ("SUM", i64), ("MIN", x)]
    for use_filter in (False, True):
        fl = filt if use_filter else None
        accs, oaccs = [], []
        for fun, col in spec:
            t = col.type if col is not None else pa.int64()
            fld = dfgpu.operators.field_of_array("v", ctx.from_arrow(col.slice(0, 4))) if col is not None else None
            accs.append(dfgpu.GroupsAccumulator(ctx, KIND[fun], fld.dtype if fld else dfgpu.capi.INT64, fld.precision if fld else 0, fld.scale if fld else 0))
            oaccs.append(po.Acc(fun, t))
        dev = {id(c): ctx.from_arrow(c) for _, c in spec if c is not None}
        gd = ctx.from_arrow(pa.array(g.astype(np.uint32)))
        fd = ctx.from_arrow(fl) if fl is not None else None
        dfgpu.GroupsAccumulator.update_batch_multi(ctx, accs, [dev[id(c)] if c is not None else None for _, c in spec], [fd] * len(spec), gd, ngroups)
        for (fun, col), a, o in zip(spec, accs, oaccs):
            o.update_batch(col, g, fl, ngroups)
            floats = col is not None and pa.types.is_floating(col.type) and fun in ("SUM", "AVG")
            for st, ost in zip(a.state(), o.state()):
                check_equal(st.to_arrow(), ost, floats)
            check_equal(a.evaluate().to_arrow(), o.evaluate(), floats)


def test_count_star_and_resize_only_update(ctx):
    import dfgpu
    acc = dfgpu.GroupsAccumulator(ctx, KIND["COUNT"], dfgpu.capi.INT64)
    g = RNG.integers(0, 6, 1000)
    acc.update_batch(None, ctx.from_arrow(pa.array(g.astype(np.uint32))), None, 6)
    acc.update_batch(None, ctx.from_arrow(pa.array([], type=pa.uint32())), None, 9)        # resize only
    got = acc.evaluate().to_numpy()
    assert np.array_equal(got, np.concatenate([np.bincount(g, minlength=6), [0, 0, 0]]))


def test_avg_decimal_overflow_is_an_error(ctx):
    import dfgpu
    v = pa.array([decimal.Decimal(10**37)] * 4, type=pa.decimal128(38, 0))
    acc = dfgpu.GroupsAccumulator(ctx, KIND["AVG"], dfgpu.capi.DECIMAL128, 38, 0)
    acc.update_batch(ctx.from_arrow(v), ctx.from_arrow(pa.array([0, 0, 0, 0], type=pa.uint32())), None, 1)
    with pytest.raises(dfgpu.DfgpuError) as e:
        acc.evaluate()
    assert "Arithmetic Overflow in AvgAccumulator" in str(e.value)


def test_aggregate_exec_partial_final_matches_reference_vector(ctx, task_ctx):
    """check_aggregates (physical-plan/src/aggregates/mod.rs:1256-1286 data, :1509-1613): AVG(b) GROUP BY a over two
    batches: Partial state (count,sum) = a=2:(2,2.0) 3:(3,7.0) 4:(3,11.0) -- wait for the merged Final: 2->1.0, 3->2.3333333333333335, 4->3.6666666666666665."""
    import dfgpu
    from dfgpu import physical_plan as ops
    b1 = pa.table({"a": pa.array([2, 3, 4, 4], type=pa.uint32()), "b": pa.array([1.0, 2.0, 3.0, 4.0])})
    b2 = pa.table({"a": pa.array([2, 3, 3, 4], type=pa.uint32()), "b": pa.array([1.0, 2.0, 3.0, 4.0])})
    src = ops.MemoryExec([[ops.batch_from_arrow(ctx, b1), ops.batch_from_arrow(ctx, b2)]], ops.batch_from_arrow(ctx, b1).schema)
    aggr = [ops.AggregateFunctionExpr("AVG", ops.Column("b", 1), "AVG(b)", input_field=ops.Field("b", dfgpu.capi.FLOAT64))]
    partial = ops.AggregateExec("Partial", [(ops.Column("a", 0), "a")], aggr, src)
    pb = ops.collect(partial, task_ctx)[0]
    rows = sort_rows_local(pb)
    assert rows == [[2, 2, 2.0], [3, 3, 7.0], [4, 3, 11.0]]          # aggregates/mod.rs:1555-1566
    final = ops.AggregateExec("Final", [(ops.Column("a", 0), "a")], aggr, partial)
    fb = ops.collect(final, task_ctx)[0]
    assert sort_rows_local(fb) == [[2, 1.0], [3, 2.3333333333333335], [4, 3.6666666666666665]]      # :1592-1603


def sort_rows_local(batch):
    from helpers import rows_of, sort_rows
    return sort_rows(rows_of([c.to_arrow() for c in batch.columns]))


def _fused_inputs(money, n):
    if money == "decimal":
        mk = lambda lo, hi, p=15, s=2: pa.array([decimal.Decimal(int(v)).scaleb(-s) for v in RNG.integers(lo, hi, n)], type=pa.decimal128(p, s))
        one = pa.array([decimal.Decimal(1)], type=pa.decimal128(20, 0))
        types = {"qty": (15, 2), "price": (38, 4), "charge": (38, 6)}
    else:
        mk = lambda lo, hi, p=0, s=0: pa.array(RNG.integers(lo, hi, n).astype(np.float64) / 100.0)
        one = pa.array([1.0])
        types = {"qty": (0, 0), "price": (0, 0), "charge": (0, 0)}
    return mk(90000, 10494951), mk(0, 11), mk(0, 9), mk(100, 5001), one, types


@pytest.mark.parametrize("ngroups,masked", [(1, True), (6, False), (6, True), (8, False)])
@pytest.mark.parametrize("money", ["float64", "decimal"])
def test_update_batch_fused_equals_node_by_node(ctx, money, ngroups, masked):
    """dfgpu_acc_update_batch_fused (argument expressions evaluated inside the accumulate pass, kernel compiled for the expression DAG) must
    leave every accumulator as dfgpu_binary node by node + update_batch_multi do: TPC-H Q1's list -- SUM(qty), SUM(price * (1 - disc)),
    SUM(price * (1 - disc) * (1 + tax)), AVG(qty), AVG(disc), COUNT(*) -- over two batches (ragged tail, growing group count), skipped
    group ids and, for the single group, a filter."""
    import dfgpu
    T = dfgpu.capi.DECIMAL128 if money == "decimal" else dfgpu.capi.FLOAT64
    nodes = [("column", 0, 0), ("column", 1, 0), ("scalar", 2, 0), ("-", 2, 1), ("*", 0, 3), ("column", 3, 0), ("+", 2, 5), ("*", 4, 6), ("column", 4, 0)]
    acc_nodes = [8, 4, 7, 8, 1, -1]
    def make(types):
        spec = [("SUM", types["qty"]), ("SUM", types["price"]), ("SUM", types["charge"]), ("AVG", types["qty"]), ("AVG", (15, 2) if money == "decimal" else (0, 0)), ("COUNT", (0, 0))]
        return [dfgpu.GroupsAccumulator(ctx, KIND[f], dfgpu.capi.INT64 if f == "COUNT" else T, p, s) for f, (p, s) in spec]
    fused = plain = None
    for n, total in ((50001, max(1, ngroups - 2)), (1234, ngroups)):
        ext, disc, tax, qty, one, types = _fused_inputs(money, n)
        if fused is None:
            fused, plain = make(types), make(types)
        g = RNG.integers(0, total, n).astype(np.uint32)
        filt = None
        if masked and ngroups > 1:
            g[RNG.random(n) < 0.2] = 0xFFFFFFFF
        elif masked:
            filt = ctx.from_arrow(pa.array(RNG.random(n) < 0.7))
        d = {k: ctx.from_arrow(v) for k, v in dict(ext=ext, disc=disc, tax=tax, qty=qty, one=one).items()}
        gd = ctx.from_arrow(pa.array(g))
        dfgpu.GroupsAccumulator.update_batch_fused(ctx, fused, acc_nodes, nodes, [d["ext"], d["disc"], d["one"], d["tax"], d["qty"]], gd, filt, total)
        price = ctx.binary(2, d["ext"], ctx.binary(1, d["one"], d["disc"], lhs_scalar=True))
        charge = ctx.binary(2, price, ctx.binary(0, d["one"], d["tax"], lhs_scalar=True))
        dfgpu.GroupsAccumulator.update_batch_multi(ctx, plain, [d["qty"], price, charge, d["qty"], d["disc"], None], [filt] * 6, gd, total)
    for a, b in zip(fused, plain):
        for x, y in zip(a.state(), b.state()):
            check_equal(x.to_arrow(), y.to_arrow(), money == "float64")
        check_equal(a.evaluate().to_arrow(), b.evaluate().to_arrow(), money == "float64")


def test_update_batch_fused_overflow_and_unsupported_shapes(ctx):
    """Checked Decimal128 arithmetic inside the fused pass raises for rows an accumulator sees and only for those; shapes it does not
    take (nullable column, more than 8 groups, MIN / MAX) answer NOT_IMPLEMENTED before anything is accumulated."""
    import dfgpu
    big = pa.array([decimal.Decimal(10**37), decimal.Decimal(5), decimal.Decimal(10**37)], type=pa.decimal128(38, 0))
    hundred = pa.array([decimal.Decimal(100)], type=pa.decimal128(20, 0))
    nodes = [("column", 0, 0), ("scalar", 1, 0), ("*", 0, 1)]
    cols = [ctx.from_arrow(big), ctx.from_arrow(hundred)]
    mk = lambda kind="SUM": dfgpu.GroupsAccumulator(ctx, KIND[kind], dfgpu.capi.DECIMAL128, 38, 0)
    gids = lambda v: ctx.from_arrow(pa.array(v, type=pa.uint32()))
    with pytest.raises(dfgpu.DfgpuError) as e:
        dfgpu.GroupsAccumulator.update_batch_fused(ctx, [mk()], [2], nodes, cols, gids([0, 0, 0]), None, 1)
    assert "verflow" in str(e.value)
    acc = mk()
    dfgpu.GroupsAccumulator.update_batch_fused(ctx, [acc], [2], nodes, cols, gids([0xFFFFFFFF, 0, 0xFFFFFFFF]), None, 1)       # the overflowing rows are skipped
    assert acc.evaluate().to_arrow().to_pylist() == [decimal.Decimal(500)]
    acc = mk()
    dfgpu.GroupsAccumulator.update_batch_fused(ctx, [acc], [2], nodes, cols, gids([0, 0, 0]), ctx.from_arrow(pa.array([False, True, False])), 1)
    assert acc.evaluate().to_arrow().to_pylist() == [decimal.Decimal(500)]
    nullable = [ctx.from_arrow(pa.array([decimal.Decimal(1), None, decimal.Decimal(3)], type=pa.decimal128(38, 0))), cols[1]]
    for args in (([mk()], [2], nodes, nullable, gids([0, 0, 0]), None, 1), ([mk()], [2], nodes, cols, gids([0, 8, 3]), None, 9), ([mk("MIN")], [2], nodes, cols, gids([0, 0, 0]), None, 1)):
        with pytest.raises(dfgpu.DfgpuError) as e:
            dfgpu.GroupsAccumulator.update_batch_fused(ctx, *args)
        assert e.value.kind == "NotImplemented"


@pytest.mark.parametrize("nkeys,masked", [(1, False), (2, True), (2, False)])
def test_deferred_dense_group_ids_feed_accumulators_like_stored_ids(ctx, nkeys, masked):
    """dfgpu_groups_intern_deferred over dictionary keys with a small composite domain hands back ids that are not written yet: the fused
    accumulate pass computes them from the code columns, every other accumulator entry point (and export) writes them out first.  States
    must equal those fed with dfgpu_groups_intern's stored ids, over two batches (the second one meets new groups)."""
    import dfgpu
    w1, w2 = pa.array(["A", "N", "R"], type=pa.utf8()), pa.array(["F", "O"], type=pa.utf8())
    T = dfgpu.capi.FLOAT64
    nodes = [("column", 0, 0), ("column", 1, 0), ("scalar", 2, 0), ("-", 2, 1), ("*", 0, 3)]
    mk = lambda: [dfgpu.GroupsAccumulator(ctx, KIND["SUM"], T), dfgpu.GroupsAccumulator(ctx, KIND["AVG"], T), dfgpu.GroupsAccumulator(ctx, KIND["COUNT"], dfgpu.capi.INT64)]
    gv = {k: dfgpu.GroupValues(ctx, nkeys) for k in ("stored", "fused", "multi", "export")}
    accs = {k: mk() for k in ("stored", "fused", "multi")}
    for n, hi1 in ((40001, 2), (9000, 3)):
        c1 = pa.DictionaryArray.from_arrays(pa.array(RNG.integers(0, hi1, n).astype(np.int8)), w1)
        c2 = pa.DictionaryArray.from_arrays(pa.array(RNG.integers(0, 2, n).astype(np.int8)), w2)
        keys = [ctx.from_arrow(c1), ctx.from_arrow(c2)][:nkeys]
        mask = ctx.from_arrow(pa.array(RNG.random(n) < 0.6)) if masked else None
        x, y, one = ctx.from_arrow(pa.array(RNG.normal(size=n))), ctx.from_arrow(pa.array(RNG.random(n))), ctx.from_arrow(pa.array([1.0]))
        ids = gv["stored"].intern(keys, mask)
        assert np.array_equal(gv["export"].intern(keys, mask, deferred=True).to_numpy(), ids.to_numpy())
        total = len(gv["stored"])
        dfgpu.GroupsAccumulator.update_batch_fused(ctx, accs["stored"], [4, 1, -1], nodes, [x, y, one], ids, None, total)
        dfgpu.GroupsAccumulator.update_batch_fused(ctx, accs["fused"], [4, 1, -1], nodes, [x, y, one], gv["fused"].intern(keys, mask, deferred=True), None, total)
        price = ctx.binary(2, x, ctx.binary(1, one, y, lhs_scalar=True))
        dfgpu.GroupsAccumulator.update_batch_multi(ctx, accs["multi"], [price, y, None], [None] * 3, gv["multi"].intern(keys, mask, deferred=True), total)
        assert len(gv["fused"]) == len(gv["multi"]) == total
    for k in ("fused", "multi"):
        for a, b in zip(accs[k], accs["stored"]):
            for s1, s2 in zip(a.state(), b.state()):
                check_equal(s1.to_arrow(), s2.to_arrow(), True)
    for a, b in zip(gv["fused"].emit(), gv["stored"].emit()):
        assert a.to_arrow().equals(b.to_arrow())


def test_deferred_run_number_ids_feed_accumulators_like_stored_ids(ctx):
    """A clustered batch of >= 2^20 rows interned through dfgpu_groups_intern_deferred hands back run numbers that are not written yet:
    the plain SUM / AVG / COUNT pass derives them from the run-head bits, MIN (another kernel) and export write them out first."""
    import dfgpu
    n = (1 << 20) + 4321
    keys = np.repeat(np.arange(n // 3 + 1, dtype=np.int64) * 7, 3)[:n]
    kd = ctx.from_arrow(pa.array(keys))
    x = pa.array(RNG.normal(size=n))
    d = pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in RNG.integers(-10**9, 10**9, 4096)] * (n // 4096 + 1), type=pa.decimal128(15, 2)).slice(0, n)
    xd, dd = ctx.from_arrow(x), ctx.from_arrow(d)
    spec = [("SUM", xd, dfgpu.capi.FLOAT64, 0, 0), ("AVG", dd, dfgpu.capi.DECIMAL128, 15, 2), ("COUNT", None, dfgpu.capi.INT64, 0, 0), ("MIN", xd, dfgpu.capi.FLOAT64, 0, 0), ("SUM", dd, dfgpu.capi.DECIMAL128, 15, 2)]
    out = {}
    for deferred in (False, True):
        gv = dfgpu.GroupValues(ctx, 1)
        accs = [dfgpu.GroupsAccumulator(ctx, KIND[f], t, p, s) for f, _, t, p, s in spec]
        if deferred:
            assert np.array_equal(dfgpu.GroupValues(ctx, 1).intern([kd], deferred=True).to_numpy(), out["ids"])
        ids = gv.intern([kd], deferred=deferred)
        for a, (f, v, _, _, _) in zip(accs, spec):
            a.update_batch(v, ids, None, len(gv))
        if not deferred:
            out["ids"] = ids.to_numpy()
            assert np.array_equal(out["ids"], np.arange(n) // 3)
        out[deferred] = [a.evaluate().to_arrow() for a in accs]
    for (f, _, t, _, _), a, b in zip(spec, out[True], out[False]):
        check_equal(a, b, t == dfgpu.capi.FLOAT64 and f != "MIN")


@pytest.mark.parametrize("money", ["float64", "decimal"])
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_update_batch_fused_random_expression_dags(ctx, money, seed):
    """Random + - * trees over four columns and two literals (either operand side, shared subtrees, a bare column, a bare product): the
    generated kernel must agree with dfgpu_binary node by node -- bit-exact for Decimal128 (result types and checked arithmetic of every
    node included: an overflow must be raised by both or by neither), 1e-9 relative for Float64."""
    import dfgpu
    rng = np.random.default_rng(100 + seed)
    n, G = 30011, 5
    if money == "decimal":
        mk = lambda: pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-5000, 5000, n)], type=pa.decimal128(9, 2))
        lits = [pa.array([decimal.Decimal(1)], type=pa.decimal128(20, 0)), pa.array([decimal.Decimal("2.5")], type=pa.decimal128(3, 1))]
        T = dfgpu.capi.DECIMAL128
    else:
        mk = lambda: pa.array(rng.normal(size=n) * 10)
        lits = [pa.array([1.0]), pa.array([-0.25])]
        T = dfgpu.capi.FLOAT64
    host = [mk() for _ in range(4)] + lits
    cols = [ctx.from_arrow(a) for a in host]
    nodes = [("column", c, 0) for c in range(4)] + [("scalar", 4, 0), ("scalar", 5, 0)]
    arrays = list(cols)                               # node index -> evaluated array (node by node)
    is_scalar = [False] * 4 + [True, True]
    ops_code = {"+": 0, "-": 1, "*": 2}
    depth = [0] * 6
    for _ in range(7):
        while True:
            l, r = int(rng.integers(0, len(nodes))), int(rng.integers(0, len(nodes)))
            if not (is_scalar[l] and is_scalar[r]) and depth[l] + depth[r] <= 3:
                break
        op = "+-*"[int(rng.integers(0, 3))]
        if money == "decimal" and op == "*" and (depth[l] > 1 or depth[r] > 1):
            op = "+"                                  # keep the scale of nested products inside 38 digits
        nodes.append((op, l, r)); depth.append(max(depth[l], depth[r]) + (2 if op == "*" else 1)); is_scalar.append(False)
        arrays.append(ctx.binary(ops_code[op], arrays[l], arrays[r], lhs_scalar=is_scalar[l], rhs_scalar=is_scalar[r]))
    picks = [0, len(nodes) - 1, len(nodes) - 2, len(nodes) - 4, len(nodes) - 1]
    kinds = ["SUM", "SUM", "AVG", "SUM", "AVG"]
    def accs():
        out = []
        for k, nd in zip(kinds, picks):
            f = dfgpu.operators.field_of_array("v", arrays[nd])
            out.append(dfgpu.GroupsAccumulator(ctx, KIND[k], T, f.precision, f.scale))
        return out + [dfgpu.GroupsAccumulator(ctx, KIND["COUNT"], dfgpu.capi.INT64)]
    g = rng.integers(0, G, n).astype(np.uint32); g[rng.random(n) < 0.1] = 0xFFFFFFFF
    gd = ctx.from_arrow(pa.array(g))
    fused, plain = accs(), accs()
    dfgpu.GroupsAccumulator.update_batch_fused(ctx, fused, picks + [-1], nodes, cols, gd, None, G)
    dfgpu.GroupsAccumulator.update_batch_multi(ctx, plain, [arrays[nd] for nd in picks] + [None], [None] * 6, gd, G)
    for a, b in zip(fused, plain):
        check_equal(a.evaluate().to_arrow(), b.evaluate().to_arrow(), money == "float64")


def test_count_takes_the_avg_count_delta_in_large_many_group_batches(ctx):
    """update_batch_multi over >= 2^20 rows and more than 8 groups: COUNT(*) / COUNT(NULL-free column) with the same filter as an AVG over
    a NULL-free column receive the AVG's per-group count delta of the batch instead of their own pass; COUNT of a nullable column and a
    COUNT under another filter keep theirs.  States must equal separate update_batch calls, over two batches with growing groups."""
    import dfgpu
    n = (1 << 20) + 777
    def build():
        return [dfgpu.GroupsAccumulator(ctx, KIND["AVG"], dfgpu.capi.FLOAT64), dfgpu.GroupsAccumulator(ctx, KIND["COUNT"], dfgpu.capi.INT64),
                dfgpu.GroupsAccumulator(ctx, KIND["COUNT"], dfgpu.capi.INT64), dfgpu.GroupsAccumulator(ctx, KIND["COUNT"], dfgpu.capi.INT64),
                dfgpu.GroupsAccumulator(ctx, KIND["COUNT"], dfgpu.capi.INT64), dfgpu.GroupsAccumulator(ctx, KIND["SUM"], dfgpu.capi.FLOAT64)]
    multi, single = build(), build()
    for total in (3000, 5000):
        g = RNG.integers(0, total, n).astype(np.uint32); g[RNG.random(n) < 0.05] = 0xFFFFFFFF
        x = ctx.from_arrow(pa.array(RNG.normal(size=n)))
        y = ctx.from_arrow(pa.array(RNG.integers(0, 100, n)))
        z = ctx.from_arrow(pa.array(RNG.integers(0, 100, n), mask=RNG.random(n) < 0.3))
        f = ctx.from_arrow(pa.array(RNG.random(n) < 0.5))
        gd = ctx.from_arrow(pa.array(g))
        vals, filts = [x, None, y, z, None, x], [None, None, None, None, f, None]
        dfgpu.GroupsAccumulator.update_batch_multi(ctx, multi, vals, filts, gd, total)
        for a, v, fl in zip(single, vals, filts):
            a.update_batch(v, gd, fl, total)
    for a, b in zip(multi, single):
        for s1, s2 in zip(a.state(), b.state()):
            check_equal(s1.to_arrow(), s2.to_arrow(), True)


@pytest.mark.parametrize("small_first", [False, True])
def test_single_dictionary_key_with_a_large_domain_uses_the_direct_map(ctx, small_first):
    """One dictionary key column whose canonical domain is beyond the dense map (here 6000 values, two codes sharing a value, a NULL value,
    NULL codes): from 2^16 rows on groups.hip numbers groups through dmap[canonical id] instead of a hash table.  First-seen ids and
    emitted keys must equal the oracle's over batches sharing the dictionary (one with a fused mask, one single row), also when a small
    first batch went through the hash table (the map is seeded from its groups), and after a batch with ANOTHER dictionary dropped the
    groups back to value keys."""
    import dfgpu
    words = [f"w{k:05d}" for k in range(6000)]
    words[17] = words[4000]; words[99] = None
    dictionary = pa.array(words, type=pa.utf8())
    n = 200000
    codes = pa.array(RNG.integers(0, 6000, n).astype(np.int32), mask=RNG.random(n) < 0.02)
    col = pa.DictionaryArray.from_arrays(codes, dictionary)
    dev = ctx.from_arrow(col)
    gv, og = dfgpu.GroupValues(ctx, 1), po.Groups([pa.utf8()])
    cuts = [(0, 300), (300, 100000)] if small_first else [(0, 100000)]
    for lo, hi in cuts + [(100000, 100001)]:
        got = gv.intern([dev.slice(lo, hi - lo)]).to_numpy().astype(np.int64)
        assert np.array_equal(got, og.intern([col.slice(lo, hi - lo)])), f"rows {lo}:{hi}"
    m = RNG.random(n - 100001) < 0.5
    got = gv.intern([dev.slice(100001, n - 100001)], mask=ctx.from_arrow(pa.array(m))).to_numpy()
    want = og.intern([col.slice(100001, n - 100001).filter(pa.array(m))])
    assert np.array_equal(got[m].astype(np.int64), want) and (got[~m] == 0xFFFFFFFF).all()
    other = pa.array([None if v % 11 == 0 else f"w{v:05d}" for v in RNG.integers(5000, 7000, 70000)], type=pa.utf8()).dictionary_encode()
    got = gv.intern([ctx.from_arrow(other)]).to_numpy().astype(np.int64)                    # another dictionary: value keys from here on
    assert np.array_equal(got, og.intern([other]))
    got = gv.intern([dev.slice(0, 70000)]).to_numpy().astype(np.int64)
    assert np.array_equal(got, og.intern([col.slice(0, 70000)]))
    assert len(gv) == len(og)
    for a, w in zip(gv.emit(), og.emit()):
        assert a.to_arrow().equals(w)


def test_new_paths_on_empty_masked_and_degenerate_inputs(ctx):
    """Edge cases of the fused accumulate pass, the deferred ids and the direct map: zero rows, every row masked out or skipped, NULL-only
    codes, a single group -- nothing is accumulated that should not be, no group appears that should not."""
    import dfgpu
    T = dfgpu.capi.FLOAT64
    one = ctx.from_arrow(pa.array([1.0]))
    nodes = [("column", 0, 0), ("scalar", 1, 0), ("*", 0, 1)]
    # fused: zero rows, then rows that are all skipped (group id NONE) or all filtered
    acc = dfgpu.GroupsAccumulator(ctx, KIND["SUM"], T)
    dfgpu.GroupsAccumulator.update_batch_fused(ctx, [acc], [2], nodes, [ctx.from_arrow(pa.array([], type=pa.float64())), one], ctx.from_arrow(pa.array([], type=pa.uint32())), None, 3)
    assert acc.evaluate().to_arrow().to_pylist() == [None, None, None]
    x = ctx.from_arrow(pa.array(np.arange(5000, dtype=np.float64)))
    dfgpu.GroupsAccumulator.update_batch_fused(ctx, [acc], [2], nodes, [x, one], ctx.from_arrow(pa.array(np.full(5000, 0xFFFFFFFF, dtype=np.uint32))), None, 3)
    dfgpu.GroupsAccumulator.update_batch_fused(ctx, [acc], [2], nodes, [x, one], ctx.from_arrow(pa.array(np.zeros(5000, dtype=np.uint32))), ctx.from_arrow(pa.array(np.zeros(5000, dtype=bool))), 3)
    assert acc.evaluate().to_arrow().to_pylist() == [None, None, None]
    dfgpu.GroupsAccumulator.update_batch_fused(ctx, [acc], [2], nodes, [x, one], ctx.from_arrow(pa.array(np.ones(5000, dtype=np.uint32))), None, 3)
    assert acc.evaluate().to_arrow().to_pylist() == [None, float(np.arange(5000).sum()), None]
    # deferred dense ids with every row masked out, then a batch that brings the groups
    w = pa.array(["A", "B"], type=pa.utf8())
    codes = pa.DictionaryArray.from_arrays(pa.array(RNG.integers(0, 2, 3000).astype(np.int8)), w)
    gv = dfgpu.GroupValues(ctx, 1)
    ids = gv.intern([ctx.from_arrow(codes)], mask=ctx.from_arrow(pa.array(np.zeros(3000, dtype=bool))), deferred=True)
    assert len(gv) == 0 and (ids.to_numpy() == 0xFFFFFFFF).all()
    ids = gv.intern([ctx.from_arrow(codes)], deferred=True)
    assert len(gv) == 2 and set(ids.to_numpy().tolist()) == {0, 1}
    # direct map: NULL-only codes are one group; an all-masked batch adds nothing
    big = pa.array([f"v{k}" for k in range(5000)], type=pa.utf8())
    n = 70000
    nulls = pa.DictionaryArray.from_arrays(pa.array(np.zeros(n, dtype=np.int32), mask=np.ones(n, dtype=bool)), big)
    gv = dfgpu.GroupValues(ctx, 1)
    assert (gv.intern([ctx.from_arrow(nulls)]).to_numpy() == 0).all() and len(gv) == 1 and gv.emit()[0].to_arrow().to_pylist() == [None]
    some = pa.DictionaryArray.from_arrays(pa.array(RNG.integers(0, 5000, n).astype(np.int32)), big)
    got = gv.intern([ctx.from_arrow(some)], mask=ctx.from_arrow(pa.array(np.zeros(n, dtype=bool)))).to_numpy()
    assert (got == 0xFFFFFFFF).all() and len(gv) == 1


@pytest.mark.parametrize("scenario", ["plain", "after_runs", "nulls_arrive", "marker_key", "uint64", "int32", "date32"])
def test_primitive_key_table_matches_the_oracle(ctx, scenario):
    """One 8-byte integer key column without NULLs goes through the primitive-key table (key, group id and first-row word in one slot).
    First-seen ids and emitted keys must equal the oracle's over several batches (a masked one, a single row, one that forces the table
    to grow), after a clustered first batch (run numbering), when a later batch brings NULLs (the groups move to the general table),
    and when a key equals the table's empty marker (-1: the column is banned from the primitive table)."""
    import dfgpu
    rng = np.random.default_rng(21)
    t = {"uint64": pa.uint64(), "int32": pa.int32(), "date32": pa.date32()}.get(scenario, pa.int64())
    gv, og = dfgpu.GroupValues(ctx, 1), po.Groups([t])
    def check(arr, mask=None):
        got = gv.intern([ctx.from_arrow(arr)], mask=ctx.from_arrow(pa.array(mask)) if mask is not None else None).to_numpy()
        if mask is None:
            assert np.array_equal(got.astype(np.int64), og.intern([arr]))
        else:
            assert np.array_equal(got[mask].astype(np.int64), og.intern([arr.filter(pa.array(mask))])) and (got[~mask] == 0xFFFFFFFF).all()
        assert len(gv) == len(og)
    mk = {"uint64": lambda v: pa.array(v.astype(np.uint64)), "int32": lambda v: pa.array((v - 1000).astype(np.int32)),       # 4-byte keys incl. negative ones
          "date32": lambda v: pa.array(v.astype(np.int32)).cast(pa.date32())}.get(scenario, lambda v: pa.array(v.astype(np.int64)))
    if scenario == "after_runs":
        check(mk(np.repeat(np.arange(5000), 3) * 11))                                   # clustered: run numbering, no table yet
    check(mk(rng.integers(0, 3000, 20000) * 7))
    check(mk(rng.integers(0, 3000, 1)))
    check(mk(rng.integers(0, 400000, 300000) * 13), mask=rng.random(300000) < 0.7)       # more new groups than the first table holds
    if scenario == "nulls_arrive":
        check(pa.array(rng.integers(0, 5000, 40000) * 7, mask=rng.random(40000) < 0.1))
    if scenario == "marker_key":
        v = rng.integers(0, 5000, 40000) * 7; v[::97] = -1
        check(pa.array(v.astype(np.int64)))
    check(mk(rng.integers(0, 500000, 100000) * 13))
    for a, w in zip(gv.emit(), og.emit()):
        assert a.to_arrow().equals(w)


def test_avg_merge_of_a_partial_state_without_values_stays_null(ctx):
    """AVG's merge_batch (average.rs:472-509) adds the partial counts without touching the null state; only a non-NULL partial sum marks its group as seen.  A Partial stage emits
    (count 0, NULL sum) for a group whose argument was NULL in every row: merged alone the group stays NULL, merged with a real partial state it is that state's average."""
    import dfgpu
    acc = dfgpu.GroupsAccumulator(ctx, dfgpu.capi.AGG_AVG, dfgpu.capi.FLOAT64)
    counts = ctx.from_arrow(pa.array([0, 2, 0, 3], type=pa.uint64()))
    sums = ctx.from_arrow(pa.array([None, 5.0, None, 9.0], type=pa.float64()))
    gids = ctx.from_arrow(pa.array([0, 1, 2, 2], type=pa.uint32()))
    acc.merge_batch([counts, sums], gids, None, 3)
    assert acc.evaluate().to_arrow().to_pylist() == [None, 2.5, 3.0]


def test_run_mode_group_keys_stay_ungathered_until_needed(ctx):
    """A clustered first batch numbers its groups by runs; the stored keys are then "key column c at the first row of run g" and are not gathered (dfgpu_groups_emit_deferred
    hands out the columns and the first rows).  emit(), a second batch (whose first run may continue the last group), EmitTo::First all see gathered keys; with the option off
    the keys are gathered at once.  Same ids, same keys either way."""
    import dfgpu
    n = 400_000
    k0 = np.repeat(np.arange(n // 4, dtype=np.int64) * 3 + 1, 4)                  # 100 000 runs of 4
    k1 = (k0 % 7).astype(np.int32)                                                # constant within a run
    cols = lambda a, b: [ctx.from_arrow(pa.array(a)), ctx.from_arrow(pa.array(b))]
    gv = dfgpu.GroupValues(ctx, 2)
    ids = gv.intern(cols(k0, k1)).to_numpy()
    assert np.array_equal(ids, np.arange(n) // 4)
    d = gv.emit_deferred()
    assert d is not None
    src, rows = d
    assert np.array_equal(rows.to_numpy(), np.arange(0, n, 4)) and np.array_equal(src[0].to_numpy(), k0) and np.array_equal(src[1].to_numpy(), k1)
    # a second batch: its first run continues group 99 999, then new keys
    k0b = np.concatenate([np.full(3, k0[-1]), np.repeat(np.arange(5, dtype=np.int64) * 3 + k0[-1] + 3, 2)]); k1b = (k0b % 7).astype(np.int32)
    ids2 = gv.intern(cols(k0b, k1b)).to_numpy()
    assert ids2.tolist() == [n // 4 - 1] * 3 + [n // 4 + i // 2 for i in range(10)]
    assert gv.emit_deferred() is None                                            # gathered by now
    e = gv.emit()
    want0 = np.concatenate([k0[::4], k0b[3::2]])
    assert np.array_equal(e[0].to_numpy(), want0) and np.array_equal(e[1].to_numpy(), (want0 % 7).astype(np.int32))
    ctx.set_option("group_lazy_keys", 0)
    try:
        gv2 = dfgpu.GroupValues(ctx, 2); gv2.intern(cols(k0, k1))
        assert gv2.emit_deferred() is None and np.array_equal(gv2.emit()[0].to_numpy(), k0[::4])
    finally:
        ctx.set_option("group_lazy_keys", 1)
