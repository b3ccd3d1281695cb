"""-m gpu: partitioned pre-aggregation (csrc/pagg.hip, dfgpu_agg_preaggregate) against the CPU oracle.

The partial rows of a batch, interned and merged (GroupValues::intern + GroupsAccumulator::merge_batch), must give exactly what the oracle's
row-by-row update gives: the same groups in the same first-seen order, bit-exact integer / count states, Float64 sums within 1e-9
relative.  Thresholds are lowered through the ctx options; the LDS table is made to overflow (more groups per partition than slots) to
cover the flush path, and the key that equals the table's EMPTY marker (-1) gets its own slot."""
import numpy as np
import pyarrow as pa
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(20262)
FLOAT_RTOL = 1e-9
KIND = {"SUM": 0, "AVG": 1, "COUNT": 2, "MIN": 3, "MAX": 4}


class forced:
    def __init__(self, ctx, force=1, min_rows=1):
        self.ctx, self.opts = ctx, {"agg_partitioned": 1, "agg_partitioned_force": force, "agg_partitioned_min_rows": min_rows}

    def __enter__(self):
        self.saved = {k: self.ctx.get_option(k) for k in self.opts}
        for k, v in self.opts.items():
            self.ctx.set_option(k, v)
        self.ctx.profile_select(None); self.ctx.profile_enable(True); self.ctx.profile_read()
        return self

    def kernels(self):
        return set(self.ctx.profile_read())

    def __exit__(self, *a):
        self.ctx.profile_enable(False)
        for k, v in self.saved.items():
            self.ctx.set_option(k, v)


def run_device(ctx, key, aggs, mask=None):
    """aggs: list of (kind name, pyarrow values or None).  -> (emitted keys, [evaluated arrays]) after preaggregate + intern + merge"""
    import dfgpu
    kd = ctx.from_arrow(key)
    dev = {}                                                 # one device column per distinct argument (SUM(x), COUNT(x) and MIN(x) read the same column, as in a plan)
    vals = [dev.setdefault(id(v), ctx.from_arrow(v)) if v is not None else None for _, v in aggs]
    pk, states = dfgpu.agg_preaggregate(ctx, kd, [KIND[k] for k, _ in aggs], vals, mask=ctx.from_arrow(pa.array(mask)) if mask is not None else None)
    gv = dfgpu.GroupValues(ctx, 1)
    gids = gv.intern([pk])
    out = []
    for (k, v), st in zip(aggs, states):
        acc = dfgpu.GroupsAccumulator(ctx, KIND[k], dfgpu.capi.INT64 if v is None else {pa.int64(): dfgpu.capi.INT64, pa.uint64(): dfgpu.capi.UINT64, pa.float64(): dfgpu.capi.FLOAT64}[v.type])
        acc.merge_batch(st, gids, None, len(gv))
        out.append(acc.evaluate().to_arrow())
    return gv.emit()[0].to_arrow(), out, len(pk)


def run_oracle(key, aggs, mask=None):
    if mask is not None:
        key = key.filter(pa.array(mask)); aggs = [(k, v.filter(pa.array(mask)) if v is not None else None) for k, v in aggs]
    og = po.Groups([key.type]); gids = og.intern([key])
    out = []
    for k, v in aggs:
        acc = po.Acc(k, pa.int64() if v is None else v.type)
        acc.update_batch(v, gids, None, len(og))
        out.append(acc.evaluate())
    return og.emit()[0], out


def compare(got_keys, got, want_keys, want):
    assert got_keys.equals(want_keys)                # same groups, same first-seen order
    for g, w in zip(got, want):
        assert g.type == w.type and len(g) == len(w)
        if pa.types.is_floating(g.type):
            assert g.is_null().equals(w.is_null())       # a group that saw no value is NULL on both sides
            a, b = g.fill_null(0.0).to_numpy(zero_copy_only=False), w.fill_null(0.0).to_numpy(zero_copy_only=False)
            assert np.allclose(a, b, rtol=FLOAT_RTOL, atol=0.0)
        else:
            assert g.equals(w)


CASES = [(200000, 5000, np.int64), (300000, 120000, np.int64), (100000, 100000, np.int64), (150000, 40, np.int64), (250000, 30000, np.int32), (4096, 4096, np.int64), (1, 1, np.int64)]


@pytest.mark.parametrize("n,card,dt", CASES, ids=[f"{c[0]}r-{c[1]}g-{np.dtype(c[2]).name}" for c in CASES])
def test_preaggregate_then_merge_equals_row_by_row(ctx, n, card, dt):
    lo, hi = (-(1 << 62), 1 << 62) if dt == np.int64 else (-(1 << 31), 1 << 31)
    pool = RNG.integers(lo, hi, card, dtype=np.int64).astype(dt)
    pool[0] = -1                                       # the EMPTY marker of the LDS table as a real key
    key = pa.array(pool[RNG.integers(0, card, n)])
    vi = pa.array(RNG.integers(-10**9, 10**9, n).astype(np.int64)); vf = pa.array(RNG.random(n) * 1000 - 300)
    aggs = [("SUM", vi), ("COUNT", None), ("MIN", vi), ("MAX", vi), ("AVG", vf), ("SUM", vf), ("COUNT", vi)]
    with forced(ctx) as f:
        gk, got, m = run_device(ctx, key, aggs)
        assert "pa_aggregate" in f.kernels()
    wk, want = run_oracle(key, aggs)
    compare(gk, got, wk, want)
    assert m >= len(wk)


def test_table_overflow_flushes_and_still_merges_exactly(ctx):
    """The sample (every 2nd row of 2.2 M) sees 100 keys, the other rows are all distinct: 64 partitions of ~17 000 groups each against
    4096-slot tables -> the workgroups flush again and again, keys come back in several partial rows, the merge adds them up exactly."""
    n = 2200000
    key = np.arange(n, dtype=np.int64) * 7919 + 12345
    key[0::2] = RNG.integers(0, 100, len(key[0::2])) * 3
    key = pa.array(key)
    vu = pa.array(RNG.integers(0, 10**12, n).astype(np.uint64))
    aggs = [("SUM", vu), ("MAX", vu), ("COUNT", None)]
    with forced(ctx) as f:
        gk, got, m = run_device(ctx, key, aggs)
    wk, want = run_oracle(key, aggs)
    compare(gk, got, wk, want)
    assert m > len(wk)                                   # some key came back in more than one partial row


def test_skewed_keys_cut_the_hot_partition_into_slices(ctx):
    """Zipf keys: one partition holds a large share of the rows and is aggregated by several workgroups (slices), each leaving its own
    partial rows for the hot keys -- merged downstream, results exact."""
    n = 3000000
    key = pa.array(((np.random.default_rng(5).zipf(1.2, n) % 200000) * 7919).astype(np.int64))
    v = pa.array(RNG.integers(-1000, 1000, n).astype(np.int64)); f = pa.array(RNG.random(n))
    aggs = [("SUM", v), ("COUNT", None), ("MIN", v), ("AVG", f)]
    with forced(ctx):
        gk, got, m = run_device(ctx, key, aggs)
    wk, want = run_oracle(key, aggs)
    compare(gk, got, wk, want)


def test_fused_selection_mask(ctx):
    n = 200000
    key = pa.array(RNG.integers(0, 20000, n).astype(np.int64)); v = pa.array(RNG.integers(0, 1000, n).astype(np.int64)); mask = RNG.random(n) < 0.4
    aggs = [("SUM", v), ("COUNT", None)]
    with forced(ctx):
        gk, got, _ = run_device(ctx, key, aggs, mask)
    wk, want = run_oracle(key, aggs, mask)
    compare(gk, got, wk, want)


def test_sample_declines_clustered_few_and_unsupported_shapes(ctx):
    import dfgpu
    n = 300000
    v = ctx.from_arrow(pa.array(np.arange(n, dtype=np.int64)))
    with forced(ctx, force=0):
        for key in (np.sort(RNG.integers(0, 50000, n)), RNG.integers(0, 100, n)):          # clustered; few groups
            with pytest.raises(dfgpu.DfgpuError):
                dfgpu.agg_preaggregate(ctx, ctx.from_arrow(pa.array(key.astype(np.int64))), [0], [v])
        with pytest.raises(dfgpu.DfgpuError):                                                # a Float64 key cannot be packed
            dfgpu.agg_preaggregate(ctx, ctx.from_arrow(pa.array(RNG.random(n))), [0], [v])
        pk, _ = dfgpu.agg_preaggregate(ctx, ctx.from_arrow(pa.array(RNG.integers(0, 50000, n), mask=RNG.random(n) < 0.1)), [0], [v])     # a nullable key is taken since round 3 (NULL = a group of its own)
        assert len(pk) >= 49000 and pk.null_count == 1
        pk, _ = dfgpu.agg_preaggregate(ctx, ctx.from_arrow(pa.array(RNG.integers(0, 50000, n).astype(np.int64))), [0], [v])     # taken
        assert len(pk) >= 49000


def test_aggregate_exec_takes_the_partitioned_path_and_matches(ctx):
    """AggregateExec(Single) GROUP BY an unclustered Int64 key through the plan layer: partial rows interned + merged == the ordinary path."""
    import dfgpu
    from dfgpu import capi, physical_plan as ops
    n = 400000
    k = RNG.integers(0, 60000, n).astype(np.int64) * 104729; v = RNG.integers(-1000, 1000, n).astype(np.int64); f = RNG.random(n)
    batch = ops.batch_from_arrow(ctx, pa.table({"k": pa.array(k), "v": pa.array(v), "f": pa.array(f)}))
    C, F = ops.Column, ops.Field
    aggs = lambda: [ops.AggregateFunctionExpr("SUM", C("v", 1), "s", input_field=F("v", capi.INT64)), ops.AggregateFunctionExpr("COUNT", None, "c"),
                    ops.AggregateFunctionExpr("AVG", C("f", 2), "a", input_field=F("f", capi.FLOAT64)), ops.AggregateFunctionExpr("MAX", C("v", 1), "m", input_field=F("v", capi.INT64))]
    tc = ops.TaskContext(ctx, batch_size=8192)
    res = []
    for on in (1, 0):
        with forced(ctx, force=0) as fz:
            ctx.set_option("agg_partitioned", on)
            plan = ops.AggregateExec("Single", [(C("k", 0), "k")], aggs(), ops.MemoryExec([[batch]], batch.schema))
            cols = [[c.to_arrow() for c in b.materialize().columns] for b in plan.execute(0, tc)]
            ran = fz.kernels()
        assert ("pa_aggregate" in ran) == bool(on)
        res.append([pa.concat_arrays([c[i] for c in cols]) for i in range(5)])
    a, b = res
    assert a[0].equals(b[0]) and a[1].equals(b[1]) and a[2].equals(b[2]) and a[4].equals(b[4])         # keys in the same (first-seen) order, SUM / COUNT / MAX exact
    assert np.allclose(a[3].to_numpy(), b[3].to_numpy(), rtol=FLOAT_RTOL, atol=0.0)


def test_two_level_partition_for_millions_of_groups(ctx):
    """5 M groups in 16 M rows: more than 2048 partitions would be needed for one pass, so the rows are split twice (P1 on the high hash bits, then a
    stable split on the low bits): every group still leaves in exactly one partial row, ids in first-seen order, sums exact."""
    n, card = 16_000_000, 5_000_000
    rng = np.random.default_rng(99)
    key = pa.array(rng.integers(0, card, n).astype(np.int64) * 2654435761 - 7)
    vi = pa.array(rng.integers(-10**6, 10**6, n).astype(np.int64))
    aggs = [("SUM", vi), ("COUNT", None), ("MAX", vi)]
    with forced(ctx, force=0) as f:
        gk, got, m = run_device(ctx, key, aggs)
        ran = f.kernels()
    assert "pa_scatter2" in ran and "pa_bounds" in ran
    wk, want = run_oracle(key, aggs)
    compare(gk, got, wk, want)
    assert m == len(wk)                                   # no key was split over partial rows


@pytest.mark.parametrize("nbatches", [1, 2, 3])
def test_first_batch_keeps_its_keys_out_of_the_hash_table_until_a_second_batch_arrives(ctx, nbatches):
    """One fully pre-aggregated batch is emitted from its partial rows (ids 0 .. n-1, no table); with further batches the keys are interned after all --
    both ways the result equals the ordinary path: same groups, same first-seen order, exact sums."""
    from dfgpu import capi, physical_plan as ops
    n = 300000
    batches = []
    for b in range(nbatches):
        k = RNG.integers(0, 50000 + 20000 * b, n).astype(np.int64) * 7919; v = RNG.integers(-1000, 1000, n).astype(np.int64)
        batches.append(ops.batch_from_arrow(ctx, pa.table({"k": pa.array(k), "v": pa.array(v)})))
    C, F = ops.Column, ops.Field
    aggs = lambda: [ops.AggregateFunctionExpr("SUM", C("v", 1), "s", input_field=F("v", capi.INT64)), ops.AggregateFunctionExpr("COUNT", None, "c"), ops.AggregateFunctionExpr("MIN", C("v", 1), "m", input_field=F("v", capi.INT64))]
    tc = ops.TaskContext(ctx, batch_size=8192)
    res = []
    for on in (1, 0):
        with forced(ctx, force=0):
            ctx.set_option("agg_partitioned", on)
            plan = ops.AggregateExec("Single", [(C("k", 0), "k")], aggs(), ops.MemoryExec([batches], batches[0].schema))
            cols = [[c.to_arrow() for c in b.materialize().columns] for b in plan.execute(0, tc)]
        res.append([pa.concat_arrays([c[i] for c in cols]) for i in range(4)])
    for a, b in zip(*res):
        assert a.equals(b)


# ---------------------------------------------------------------------------------------------- round 3: TPC-H shapes -- several key columns, NULL keys, Decimal128 sums
def run_device_multi(ctx, keys, aggs, mask=None):
    import dfgpu
    kd = [ctx.from_arrow(k) for k in keys]
    dev = {}                                                 # one device column per distinct argument (SUM(x), COUNT(x) and MIN(x) read the same column, as in a plan)
    vals = [dev.setdefault(id(v), ctx.from_arrow(v)) if v is not None else None for _, v in aggs]
    pk, states = dfgpu.agg_preaggregate(ctx, kd, [KIND[k] for k, _ in aggs], vals, mask=ctx.from_arrow(pa.array(mask)) if mask is not None else None)
    gv = dfgpu.GroupValues(ctx, len(keys))
    gids = gv.intern(pk)
    out = []
    for (k, v), st in zip(aggs, states):
        if v is None:
            acc = dfgpu.GroupsAccumulator(ctx, KIND[k], dfgpu.capi.INT64)
        elif pa.types.is_decimal(v.type):
            acc = dfgpu.GroupsAccumulator(ctx, KIND[k], dfgpu.capi.DECIMAL128, v.type.precision, v.type.scale)
        else:
            acc = dfgpu.GroupsAccumulator(ctx, KIND[k], {pa.int64(): dfgpu.capi.INT64, pa.uint64(): dfgpu.capi.UINT64, pa.float64(): dfgpu.capi.FLOAT64}[v.type])
        acc.merge_batch(st, gids, None, len(gv))
        out.append(acc.evaluate().to_arrow())
    return [a.to_arrow() for a in gv.emit()], out, len(pk[0])


def run_oracle_multi(keys, aggs, mask=None):
    if mask is not None:
        m = pa.array(mask); keys = [k.filter(m) for k in keys]; aggs = [(k, v.filter(m) if v is not None else None) for k, v in aggs]
    og = po.Groups([k.type for k in keys]); gids = og.intern(keys)
    out = []
    for k, v in aggs:
        acc = po.Acc(k, pa.int64() if v is None else v.type)
        acc.update_batch(v, gids, None, len(og))
        out.append(acc.evaluate())
    return og.emit(), out


def dec_array(values, precision=15, scale=2):
    import decimal
    return pa.array([decimal.Decimal(int(x)).scaleb(-scale) for x in values], type=pa.decimal128(precision, scale))


def compare_multi(gk, got, wk, want):
    assert len(gk) == len(wk)
    for a, b in zip(gk, wk):
        assert a.type == b.type and a.equals(b)          # same groups, same first-seen order, NULL keys where the oracle has them
    for g, w in zip(got, want):
        assert g.type == w.type and g.equals(w), (g.type, w.type)


@pytest.mark.parametrize("n,groups", [(300000, 20000), (120000, 110000), (400000, 300)], ids=["15-rows-per-group", "mostly-distinct", "hot-groups"])
def test_three_key_columns_decimal_sum_avg_count(ctx, n, groups):
    """GROUP BY (Int64, Date32, Int32) with SUM / AVG over Decimal128(15,2) and COUNT(*): Q3's aggregate shape.  Values of both signs: the
    128-bit sum is two LDS cells, the low add's returned value carries into the high one."""
    gid = RNG.integers(0, groups, n)
    k0 = pa.array((gid * 7919 - 3_000_000).astype(np.int64)); k1 = pa.array((8000 + gid % 2400).astype(np.int32)).cast(pa.date32()); k2 = pa.array((gid % 3 - 1).astype(np.int32))
    v = dec_array(RNG.integers(-10**14, 10**14, n)); w = dec_array(RNG.integers(0, 10**6, n), 38, 4)
    aggs = [("SUM", v), ("AVG", v), ("COUNT", None), ("SUM", w)]
    with forced(ctx) as f:
        gk, got, m = run_device_multi(ctx, [k0, k1, k2], aggs)
        ran = f.kernels()
    assert "pa_aggregate" in ran and "pa_pack" in ran, ran
    wk, want = run_oracle_multi([k0, k1, k2], aggs)
    compare_multi(gk, got, wk, want)
    assert m >= len(wk[0])


def test_nullable_key_columns_and_selection(ctx):
    """NULL is a group of its own in every key column (group_values/row.rs:94-146); a fused selection drops rows before they are grouped"""
    n = 200000
    a = RNG.integers(0, 5000, n).astype(np.int64); b = RNG.integers(-50, 50, n).astype(np.int32)
    k0 = pa.array(a, mask=RNG.random(n) < 0.05); k1 = pa.array(b, mask=RNG.random(n) < 0.1)
    v = pa.array(RNG.integers(-10**9, 10**9, n).astype(np.int64)); d = dec_array(RNG.integers(-10**12, 10**12, n))
    aggs = [("SUM", v), ("MIN", v), ("MAX", v), ("SUM", d), ("COUNT", None)]
    mask = RNG.random(n) < 0.7
    with forced(ctx) as f:
        gk, got, _ = run_device_multi(ctx, [k0, k1], aggs, mask=mask)
        assert "pa_aggregate" in f.kernels()
    wk, want = run_oracle_multi([k0, k1], aggs, mask=mask)
    compare_multi(gk, got, wk, want)


def test_one_nullable_key_column(ctx):
    n = 150000
    k = pa.array(RNG.integers(0, 9000, n).astype(np.int32), mask=RNG.random(n) < 0.03)
    v = pa.array(RNG.random(n)); c = pa.array(RNG.integers(0, 100, n).astype(np.int64))
    aggs = [("SUM", c), ("AVG", v), ("COUNT", None)]
    with forced(ctx) as f:
        gk, got, _ = run_device_multi(ctx, [k], aggs)
        assert "pa_aggregate" in f.kernels()
    wk, want = run_oracle_multi([k], aggs)
    assert gk[0].equals(wk[0])
    for g, w in zip(got, want):
        if pa.types.is_floating(g.type):
            assert np.allclose(g.to_numpy(zero_copy_only=False), w.to_numpy(zero_copy_only=False), rtol=FLOAT_RTOL, atol=0.0)
        else:
            assert g.equals(w)


def test_key_ranges_too_wide_to_pack_decline(ctx):
    """two Int64 columns spanning 2^62 each cannot share one 64-bit key: NotImplemented, the caller groups the ordinary way"""
    import dfgpu
    n = 100000
    k0 = pa.array(RNG.integers(-(1 << 61), 1 << 61, n).astype(np.int64)); k1 = pa.array(RNG.integers(-(1 << 61), 1 << 61, n).astype(np.int64))
    with forced(ctx):
        with pytest.raises(dfgpu.capi.DfgpuError) as e:
            run_device_multi(ctx, [k0, k1], [("COUNT", None)])
    assert "multiply beyond" in str(e.value)


@pytest.mark.parametrize("outlier", [False, True], ids=["ranges-hold", "outlier-forces-exact-ranges"])
def test_packed_key_ranges_from_a_sample(ctx, outlier):
    """large batches take the key columns' value ranges from every step-th row (widened) and check every row while packing; a value outside them makes the
    pack run again with exact ranges (sync:pa_pack_outside in the profile).  Either way groups, order and sums equal the oracle's."""
    n = 200000
    a = RNG.integers(1000, 6000, n).astype(np.int64); b = RNG.integers(0, 40, n).astype(np.int32)
    if outlier:
        a[12345] = 10**9; b[777] = -1000                  # odd rows (the sample takes every second row), far outside the widened estimate, exact ranges still pack (2^30 x 2^10)
    v = pa.array(RNG.integers(-1000, 1000, n).astype(np.int64))
    aggs = [("SUM", v), ("COUNT", None)]
    saved = ctx.get_option("agg_pack_estimate_min_rows"); ctx.set_option("agg_pack_estimate_min_rows", 1)
    try:
        with forced(ctx) as f:
            gk, got, _ = run_device_multi(ctx, [pa.array(a), pa.array(b)], aggs)
            ran = f.kernels()
    finally:
        ctx.set_option("agg_pack_estimate_min_rows", saved)
    assert ("sync:pa_pack_outside" in ran) == outlier, ran
    wk, want = run_oracle_multi([pa.array(a), pa.array(b)], aggs)
    compare_multi(gk, got, wk, want)


@pytest.mark.parametrize("two_level", [False, True], ids=["one-pass", "two-level"])
def test_nullable_aggregate_arguments(ctx, two_level):
    """Value columns with NULLs (accumulate.rs:126-233: a NULL takes no part; NullState::build, :328-356: a group that saw no value has a NULL state): SUM / MIN / MAX / AVG skip
    the NULL rows, COUNT(x) counts the values, a group whose argument is NULL in every row comes out NULL (COUNT 0).  The validity bits travel through the partition as one byte per
    row.  Keys, first-seen order and every state equal the oracle's row-by-row accumulation."""
    rng = np.random.default_rng(7 + two_level)
    n, card = (9_000_000, 2_600_000) if two_level else (600_000, 50_000)
    kv = rng.integers(0, card, n).astype(np.int64) * 7919 - 11
    key = pa.array(kv)
    all_null = (kv % 5) == 0                                  # every row of these groups carries a NULL in vi
    vi = pa.array(rng.integers(-10**9, 10**9, n).astype(np.int64), mask=all_null | (rng.random(n) < 0.3))
    vf = pa.array(rng.random(n) * 1000 - 300, mask=rng.random(n) < 0.2)
    vu = pa.array(rng.integers(0, 10**6, n).astype(np.int64))               # a column without NULLs beside them
    aggs = [("SUM", vi), ("COUNT", vi), ("MIN", vi), ("AVG", vf), ("COUNT", None), ("SUM", vu)]
    with forced(ctx, force=0 if two_level else 1) as f:
        gk, got, m = run_device(ctx, key, aggs)
        ran = f.kernels()
    assert "pa_aggregate" in ran and (("pa_scatter2" in ran) == two_level)
    wk, want = run_oracle(key, aggs)
    compare(gk, got, wk, want)
    assert got[0].null_count > 0 and got[0].null_count == want[0].null_count and got[2].null_count == want[2].null_count


def test_float_min_max_with_nans_and_infinities(ctx):
    """MIN / MAX over Float64 through the partitioned path: the reference's closures (`if *cur > new`, starting at f64::MAX / f64::MIN, min_max.rs:102-139) never take a NaN and
    never move past their starting value, so the result does not depend on the order rows are met in -- NaN inputs are ignored, a group of NaNs only keeps the starting value, an
    infinity beyond the starting value never enters.  Nullable too."""
    rng = np.random.default_rng(31)
    n, card = 500_000, 40_000
    kv = rng.integers(0, card, n).astype(np.int64) * 31 + 3
    x = rng.normal(size=n) * 1e6
    x[rng.random(n) < 0.02] = np.nan
    x[rng.random(n) < 0.01] = np.inf
    x[rng.random(n) < 0.01] = -np.inf
    x[(kv % 11) == 0] = np.nan                                  # whole groups of NaNs
    vf = pa.array(x, mask=rng.random(n) < 0.1)
    aggs = [("MIN", vf), ("MAX", vf), ("COUNT", vf), ("SUM", pa.array(rng.integers(0, 100, n).astype(np.int64)))]
    with forced(ctx) as f:
        gk, got, m = run_device(ctx, pa.array(kv), aggs)
        assert "pa_aggregate" in f.kernels()
    wk, want = run_oracle(pa.array(kv), aggs)
    assert gk.equals(wk)
    for g, w in zip(got, want):
        assert g.type == w.type and g.is_null().equals(w.is_null())
        a, b = g.fill_null(0).to_numpy(zero_copy_only=False), w.fill_null(0).to_numpy(zero_copy_only=False)
        assert np.array_equal(a, b)                               # MIN / MAX pick one of the inputs (or the starting value): exact


# ---------------------------------------------------------------------------------------------- round 4: partial rows in any order under a sort over the group columns
def test_any_order_flag_returns_the_same_partial_rows(ctx):
    """DFGPU_PREAGG_ANY_ORDER: the partial rows are the ordered call's rows in another order (partition order), nothing else changes; the ordering pass does not run."""
    import dfgpu
    n, card = 600_000, 90_000
    rng = np.random.default_rng(41)
    key = ctx.from_arrow(pa.array(rng.integers(0, card, n).astype(np.int64) * 7919 - 3))
    v = ctx.from_arrow(pa.array(rng.integers(-10**6, 10**6, n).astype(np.int64)))
    kinds = [KIND["SUM"], KIND["COUNT"], KIND["MIN"]]
    out = {}
    for any_order in (False, True):
        with forced(ctx, force=0) as f:
            pk, states = dfgpu.agg_preaggregate(ctx, key, kinds, [v, None, v], any_order=any_order)
            ran = f.kernels()
        assert ("pa_order" in ran) == (not any_order) and "pa_aggregate" in ran
        out[any_order] = (pk.to_numpy(), [s[0].to_numpy() for s in states])
    (k0, s0), (k1, s1) = out[False], out[True]
    assert len(k0) == len(k1) == len(np.unique(k0)) and not np.array_equal(k0, k1)
    o0, o1 = np.argsort(k0, kind="stable"), np.argsort(k1, kind="stable")
    assert np.array_equal(k0[o0], k1[o1])
    for a, b in zip(s0, s1):
        assert np.array_equal(a[o0], b[o1])


@pytest.mark.parametrize("shape", ["sort_over_all_group_columns", "having_and_projection_between", "sort_misses_a_group_column", "sort_on_an_aggregate_only"])
def test_sort_over_the_group_columns_lets_the_aggregation_emit_in_any_order(ctx, shape):
    """dfgpu_plan_sort marks the AggregateExec below it (through FilterExec / CoalesceBatchesExec / column-only ProjectionExec) when the sort keys hold every group column:
    the sorted result is the same rows in the same order as with the ordinary path, and the ordering pass of the pre-aggregation runs only when the sort could show the
    order of its input (a group column missing from the keys: ties keep input order, which must then be the first-seen order)."""
    from dfgpu import capi, physical_plan as ops
    import pyarrow as pa
    n = 500_000
    rng = np.random.default_rng(43)
    k1 = rng.integers(0, 300, n).astype(np.int64) * 1009; k2 = rng.integers(0, 200, n).astype(np.int32); v = rng.integers(0, 5, n).astype(np.int64)
    batch = ops.batch_from_arrow(ctx, pa.table({"k1": pa.array(k1), "k2": pa.array(k2), "v": pa.array(v)}))
    C, F, B, L = ops.Column, ops.Field, ops.BinaryExpr, ops.Literal
    tc = ops.TaskContext(ctx, batch_size=8192)

    def build():
        agg = ops.AggregateExec("Single", [(C("k1", 0), "k1"), (C("k2", 1), "k2")], [ops.AggregateFunctionExpr("SUM", C("v", 2), "s", input_field=F("v", capi.INT64)), ops.AggregateFunctionExpr("COUNT", None, "c")],
                                ops.MemoryExec([[batch]], batch.schema))
        S = ops.PhysicalSortExpr
        if shape == "sort_over_all_group_columns":
            return ops.SortExec([S(C("s", 2), True, True), S(C("k2", 1), False, False), S(C("k1", 0), True, False)], agg), True
        if shape == "having_and_projection_between":
            having = ops.CoalesceBatchesExec(ops.FilterExec(B(C("c", 3), ">", L(5, pa.int64())), agg), 8192)
            proj = ops.ProjectionExec([(C("c", 3), "c"), (C("k2", 1), "b"), (C("k1", 0), "a"), (C("s", 2), "s")], having)
            return ops.SortExec([S(C("c", 0), True, True), S(C("a", 2), False, False), S(C("b", 1), False, False)], proj, fetch=1000), True
        if shape == "sort_misses_a_group_column":
            return ops.SortExec([S(C("c", 3), True, True), S(C("k1", 0), False, False)], agg), False
        return ops.SortExec([S(C("s", 2), True, True)], agg), False

    res = []
    for on in (1, 0):
        with forced(ctx, force=0) as fz:
            ctx.set_option("agg_partitioned", on)
            plan, marked = build()
            cols = [[c.to_arrow() for c in b.materialize().columns] for b in plan.execute(0, tc)]
            ran = fz.kernels()
        assert ("pa_aggregate" in ran) == bool(on)
        if on:
            assert ("pa_order" in ran) == (not marked), (shape, sorted(ran))
        res.append([pa.concat_arrays([c[i] for c in cols]) for i in range(len(cols[0]))])
    for a, b in zip(*res):
        assert a.equals(b)


@pytest.mark.parametrize("dt,nullable", [(np.int32, False), (np.int64, True), (np.int32, True)], ids=["int32", "int64-nullable", "int32-nullable"])
def test_integer_argument_cast_to_float64_while_it_is_partitioned(ctx, dt, nullable):
    """value_casts = FLOAT64: AVG / SUM / MIN / MAX(CAST(x AS DOUBLE)) over an integer column handed over uncast == the same call over the cast column, state by state
    (the sums add the same doubles in the same order inside a partition, so they are compared exactly), NULLs of the column included."""
    import dfgpu
    from dfgpu import capi
    n, card = 500_000, 40_000
    rng = np.random.default_rng(47)
    key = ctx.from_arrow(pa.array(rng.integers(0, card, n).astype(np.int64) * 31 + 5))
    info = np.iinfo(dt)
    x = rng.integers(max(info.min, -10**12), min(info.max, 10**12), n).astype(dt)
    xa = pa.array(x, mask=(rng.random(n) < 0.1) if nullable else None)
    xi = ctx.from_arrow(xa); xf = ctx.from_arrow(xa.cast(pa.float64()))
    kinds = [KIND["AVG"], KIND["SUM"], KIND["MIN"], KIND["MAX"], KIND["COUNT"]]
    with forced(ctx, force=0) as f:
        pk0, st0 = dfgpu.agg_preaggregate(ctx, key, kinds, [xf] * 5)
        f.kernels()
        pk1, st1 = dfgpu.agg_preaggregate(ctx, key, kinds, [xi] * 5, casts=[capi.FLOAT64] * 5)
        assert "pa_aggregate" in f.kernels()
    assert pk0.to_arrow().equals(pk1.to_arrow())
    for a, b in zip(st0, st1):
        assert len(a) == len(b)
        for u, v in zip(a, b):
            assert u.to_arrow().equals(v.to_arrow())
    with forced(ctx, force=0):
        with pytest.raises(dfgpu.DfgpuError) as e:
            dfgpu.agg_preaggregate(ctx, key, [KIND["SUM"]], [xf], casts=[capi.FLOAT64])          # a cast of a column that is no Int32 / Int64 column: declined
        assert e.value.kind == "NotImplemented"
        with pytest.raises(dfgpu.DfgpuError) as e:
            dfgpu.agg_preaggregate(ctx, key, [KIND["SUM"]], [ctx.from_arrow(pa.array(rng.integers(0, 100, n).astype(np.uint16)))], casts=[capi.FLOAT64])
        assert e.value.kind == "NotImplemented"


def test_aggregate_exec_hands_cast_arguments_down_uncast(ctx):
    """AVG(CAST(len AS DOUBLE)) planned as a ProjectionExec computing the cast below AggregateExec (the ClickBench shape): the pre-aggregation takes the Int32 column with the
    cast named beside it -- no k_cast pass over the batch -- and the result equals the ordinary path's."""
    from dfgpu import capi, physical_plan as ops
    n = 400_000
    rng = np.random.default_rng(53)
    k = rng.integers(0, 50_000, n).astype(np.int64) * 7 + 1; ln = rng.integers(0, 500, n).astype(np.int32); w = rng.integers(0, 10**6, n).astype(np.int64)
    batch = ops.batch_from_arrow(ctx, pa.table({"k": pa.array(k), "len": pa.array(ln, mask=rng.random(n) < 0.05), "w": pa.array(w)}))
    C, F = ops.Column, ops.Field
    tc = ops.TaskContext(ctx, batch_size=8192)
    res = []
    for on in (1, 0):
        with forced(ctx, force=0) as fz:
            ctx.set_option("agg_partitioned", on)
            proj = ops.ProjectionExec([(C("k", 0), "k"), (ops.CastExpr(C("len", 1), capi.FLOAT64), "lenf"), (C("w", 2), "w")], ops.MemoryExec([[batch]], batch.schema))
            aggs = [ops.AggregateFunctionExpr("AVG", C("lenf", 1), "a", input_field=F("x", capi.FLOAT64)), ops.AggregateFunctionExpr("COUNT", None, "c"),
                    ops.AggregateFunctionExpr("MAX", C("w", 2), "m", input_field=F("x", capi.INT64)), ops.AggregateFunctionExpr("SUM", C("lenf", 1), "s", input_field=F("x", capi.FLOAT64))]
            plan = ops.AggregateExec("Single", [(C("k", 0), "k")], aggs, proj)
            cols = [[c.to_arrow() for c in b.materialize().columns] for b in plan.execute(0, tc)]
            ran = fz.kernels()
        assert ("pa_aggregate" in ran) == bool(on)
        if on:
            assert "k_cast" not in ran, sorted(ran)
        res.append([pa.concat_arrays([c[i] for c in cols]) for i in range(5)])
    a, b = res
    assert a[0].equals(b[0]) and a[2].equals(b[2]) and a[3].equals(b[3])
    for i in (1, 4):
        x, y = a[i], b[i]
        assert x.is_null().equals(y.is_null())
        assert np.allclose(x.fill_null(0).to_numpy(), y.fill_null(0).to_numpy(), rtol=FLOAT_RTOL, atol=0.0)


def test_the_mark_stays_on_the_sorts_own_copy_of_the_aggregation(ctx):
    """dfgpu_plan_sort marks its own copy of the plan below it: the caller's AggregateExec, executed by itself after a SortExec was built over it, still emits its groups in
    first-seen order (== the ordinary path), and the sort over it skips the ordering pass."""
    from dfgpu import capi, physical_plan as ops
    n = 300_000
    rng = np.random.default_rng(59)
    k = rng.integers(0, 40_000, n).astype(np.int64) * 13; v = rng.integers(-100, 100, n).astype(np.int64)
    batch = ops.batch_from_arrow(ctx, pa.table({"k": pa.array(k), "v": pa.array(v)}))
    C, F = ops.Column, ops.Field
    tc = ops.TaskContext(ctx, batch_size=8192)
    mk = lambda: ops.AggregateExec("Single", [(C("k", 0), "k")], [ops.AggregateFunctionExpr("SUM", C("v", 1), "s", input_field=F("v", capi.INT64))], ops.MemoryExec([[batch]], batch.schema))
    with forced(ctx, force=0) as fz:
        agg = mk()
        srt = ops.SortExec([ops.PhysicalSortExpr(C("s", 1), True, True), ops.PhysicalSortExpr(C("k", 0), False, False)], agg)
        sorted_cols = [[c.to_arrow() for c in b.materialize().columns] for b in srt.execute(0, tc)]
        ran = fz.kernels()
        assert "pa_aggregate" in ran and "pa_order" not in ran
        alone = [[c.to_arrow() for c in b.materialize().columns] for b in agg.execute(0, tc)]
        assert "pa_order" in fz.kernels()
        ctx.set_option("agg_partitioned", 0)
        plain = [[c.to_arrow() for c in b.materialize().columns] for b in mk().execute(0, tc)]
    cat = lambda cols, i: pa.concat_arrays([c[i] for c in cols])
    assert cat(alone, 0).equals(cat(plain, 0)) and cat(alone, 1).equals(cat(plain, 1))
    ks, ss = cat(sorted_cols, 0).to_numpy(), cat(sorted_cols, 1).to_numpy()
    order = np.lexsort((cat(plain, 0).to_numpy(), -cat(plain, 1).to_numpy()))
    assert np.array_equal(ks, cat(plain, 0).to_numpy()[order]) and np.array_equal(ss, cat(plain, 1).to_numpy()[order])


def test_many_partition_scatter_staged_in_two_rounds(ctx):
    """513 .. 2048 partitions: the 8192-row tile's columns staged in two rounds of half a tile (two workgroups per CU; option partition_two_round_staging, off by default --
    measured slower) give the same merged groups and states as the whole tile staged at once and as the oracle's row-by-row update; a Decimal128 argument (two 8-byte halves per row), a
    nullable Int64 one (flag byte column) and an Int32 one cast on the way."""
    import dfgpu
    from dfgpu import capi
    n, card = 3_000_000, 1_500_000
    rng = np.random.default_rng(67)
    key = pa.array(rng.integers(0, card, n).astype(np.int64) * 2654435761 + 11)
    vi = pa.array(rng.integers(-10**6, 10**6, n).astype(np.int64), mask=rng.random(n) < 0.1)
    vd = dec_array(rng.integers(-10**9, 10**9, n).tolist(), precision=30, scale=2)
    v32 = pa.array(rng.integers(0, 500, n).astype(np.int32))
    kd = ctx.from_arrow(key); cols = [ctx.from_arrow(vi), ctx.from_arrow(vd), ctx.from_arrow(v32)]
    kinds = [KIND["SUM"], KIND["MAX"], KIND["SUM"], KIND["AVG"], KIND["COUNT"]]
    vals = [cols[0], cols[0], cols[1], cols[2], None]
    casts = [0, 0, 0, capi.FLOAT64, 0]
    # six accumulator cells leave 2048 table slots per partition: partitions flush before their end, a key comes back in several partial rows, and where the cuts fall
    # depends on the order the LDS-atomic ranks gave the rows -- so the comparison is made after intern + merge_batch (what the plan layer does), not on the partial rows
    in_types = [(capi.INT64, 0, 0), (capi.INT64, 0, 0), (capi.DECIMAL128, 30, 2), (capi.FLOAT64, 0, 0), (capi.INT64, 0, 0)]
    out = []
    for on in (1, 0):
        ctx.set_option("partition_two_round_staging", on)
        try:
            with forced(ctx, force=1) as f:
                pk, states = dfgpu.agg_preaggregate(ctx, kd, kinds, vals, casts=casts)
                assert "pa_scatter" in f.kernels()
        finally:
            ctx.set_option("partition_two_round_staging", 0)
        gv = dfgpu.GroupValues(ctx, 1); gids = gv.intern([pk]); res = []
        for kind, (t, p_, s_), st in zip(kinds, in_types, states):
            acc = dfgpu.GroupsAccumulator(ctx, kind, t, p_, s_); acc.merge_batch(st, gids, None, len(gv)); res.append(acc.evaluate().to_arrow())
        out.append((gv.emit()[0].to_arrow(), res))
    (k1, r1), (k0, r0) = out
    o1, o0 = np.argsort(k1.to_numpy(), kind="stable"), np.argsort(k0.to_numpy(), kind="stable")          # group order follows the first partial row of a key: compare by key
    assert np.array_equal(k1.to_numpy()[o1], k0.to_numpy()[o0]) and len(k1) == len(np.unique(k1.to_numpy()))
    for a, b in zip(r1, r0):
        x, y = a.take(pa.array(o1)), b.take(pa.array(o0))
        if pa.types.is_floating(x.type):
            assert x.is_null().equals(y.is_null()) and np.allclose(x.fill_null(0).to_numpy(), y.fill_null(0).to_numpy(), rtol=FLOAT_RTOL, atol=0.0)
        else:
            assert x.equals(y)
    # against the oracle's row-by-row update: SUM / MAX of the nullable column and COUNT(*) exact
    wk, want = run_oracle(key, [("SUM", vi), ("MAX", vi), ("COUNT", None)])
    ow = np.argsort(wk.to_numpy(), kind="stable")
    assert np.array_equal(k1.to_numpy()[o1], wk.to_numpy()[ow])
    for got, w in zip((r1[0], r1[1], r1[4]), want):
        assert got.take(pa.array(o1)).cast(pa.int64()).equals(w.take(pa.array(ow)).cast(pa.int64()))
