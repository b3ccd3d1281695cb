"""-m gpu: SortExec (lexsort_to_indices) and RepartitionExec (hash partition) on device vs the CPU oracle -- index-exact."""
import numpy as np
import pyarrow as pa
import pytest

from oracle import pyoracle as po
from test_gpu_core import rand_array

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(11)

SORT_KINDS = ["int8", "int16", "int32", "int64", "uint8", "uint32", "uint64", "float32", "float64", "date32", "bool", "decimal", "dict_int"]


def sort_col(kind, n, nf):
    if kind == "dict_int":
        return pa.array(RNG.integers(0, 20, n).astype(np.int64), mask=RNG.random(n) < nf if nf else None).dictionary_encode()
    if kind in ("int64", "float64"):     # few distinct values => many ties (stability matters)
        a = rand_array(kind, n, nf, RNG)
        return pa.array([None if v is None else (v if i % 2 else (1.5 if kind == "float64" else 7)) for i, v in enumerate(a.to_pylist())], type=a.type)
    return rand_array(kind, n, nf, RNG)


@pytest.mark.parametrize("kind", SORT_KINDS)
@pytest.mark.parametrize("desc,nf_first", [(False, True), (True, True), (False, False), (True, False)])
def test_single_column_sort(ctx, kind, desc, nf_first):
    for n, nf in [(0, 0), (1, 0), (2, 0.5), (1000, 0.2), (5000, 0.0), (70000, 0.1)]:
        a = sort_col(kind, n, nf)
        got = ctx.sort_to_indices([ctx.from_arrow(a)], [desc], [nf_first]).to_numpy()
        assert np.array_equal(got, po.lexsort_to_indices([a], [desc], [nf_first])), f"n={n}"


def test_float_sort_nan_placement(ctx):
    """test_lex_sort_by_float (sorts/sort.rs:1290-1392): NaN sorts above every number, NULLs by nulls_first."""
    a = pa.array([float("nan"), None, None, float("nan"), 1.0, 2.0, 3.0, -0.0, 0.0, float("-inf"), float("inf")], type=pa.float32())
    b = pa.array([10.0, 20.0, 10.0, 100.0, float("nan"), None, None, float("nan"), 1.0, 2.0, 3.0], type=pa.float64())
    for da, na, db, nb in [(True, True, False, False), (False, False, True, True), (False, True, False, True)]:
        got = ctx.sort_to_indices([ctx.from_arrow(a), ctx.from_arrow(b)], [da, db], [na, nb]).to_numpy()
        assert np.array_equal(got, po.lexsort_to_indices([a, b], [da, db], [na, nb]))
    asc = ctx.sort_to_indices([ctx.from_arrow(a)], [False], [False]).to_numpy()
    vals = a.take(pa.array(asc)).to_pylist()
    assert vals[-2:] == [None, None] and all(v != v for v in vals[-4:-2]) and vals[0] == float("-inf")


def test_multi_column_sort_q3_shape_and_fetch(ctx):
    """Q3 sort keys: revenue Decimal128(38,4) DESC NULLS FIRST, o_orderdate Date32 ASC NULLS LAST."""
    import decimal
    n = 50000
    rev = pa.array([decimal.Decimal(int(v)).scaleb(-4) for v in RNG.integers(0, 3000, n) * 10**6], type=pa.decimal128(38, 4))
    date = pa.array(RNG.integers(8035, 9204, n).astype(np.int32)).cast(pa.date32())
    got = ctx.sort_to_indices([ctx.from_arrow(rev), ctx.from_arrow(date)], [True, False], [True, False]).to_numpy()
    assert np.array_equal(got, po.lexsort_to_indices([rev, date], [True, False], [True, False]))
    top = ctx.sort_to_indices([ctx.from_arrow(rev), ctx.from_arrow(date)], [True, False], [True, False], fetch=10).to_numpy()
    assert np.array_equal(top, got[:10])
    three = [rand_array("int8", 20000, 0.1, RNG), rand_array("bool", 20000, 0.1, RNG), rand_array("uint16", 20000, 0.0, RNG)]
    got3 = ctx.sort_to_indices([ctx.from_arrow(c) for c in three], [False, True, False], [False, True, True]).to_numpy()
    assert np.array_equal(got3, po.lexsort_to_indices(three, [False, True, False], [False, True, True]))


@pytest.mark.parametrize("kind", ["int64", "float64", "decimal", "int8", "utf8", "mixed", "ties"])
@pytest.mark.parametrize("desc,nf_first", [(False, True), (True, False)])
def test_topk_selection_equals_full_sort_then_slice(ctx, kind, desc, nf_first):
    """SortExec with fetch (≙ physical-plan/src/topk/mod.rs) on inputs large enough for the radix-select path (sort.hip k_topk_step):
    indices must equal the first `fetch` indices of the full (stable) sort -- ties at the boundary keep row order, NULL / NaN
    placement included; an input whose keys are all equal falls back to the full sort."""
    import decimal
    n = 200000
    if kind == "utf8":
        cols = [pa.array([None if RNG.random() < 0.05 else f"k{v:07d}" for v in RNG.integers(0, 10**6, n)], type=pa.utf8())]
    elif kind == "mixed":
        cols = [pa.array(RNG.integers(0, 50, n).astype(np.int32)), pa.array([None if RNG.random() < 0.1 else float(v) for v in RNG.integers(0, 1000, n)], type=pa.float64()),
                pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in RNG.integers(-10**6, 10**6, n)], type=pa.decimal128(15, 2))]
    elif kind == "ties":
        cols = [pa.array(np.full(n, 7, dtype=np.int64))]
    elif kind == "decimal":
        cols = [pa.array([None if RNG.random() < 0.1 else decimal.Decimal(int(v)).scaleb(-2) for v in RNG.integers(-10**13, 10**13, n)], type=pa.decimal128(15, 2))]
    elif kind == "float64":
        v = RNG.normal(size=n); v[RNG.random(n) < 0.01] = np.nan
        cols = [pa.array(v, mask=RNG.random(n) < 0.05)]
    else:
        cols = [rand_array(kind, n, 0.1, RNG)]
    d, f = [desc] * len(cols), [nf_first] * len(cols)
    if kind == "mixed":
        d, f = [desc, not desc, desc], [nf_first, not nf_first, nf_first]
    dev = [ctx.from_arrow(c) for c in cols]
    full = po.lexsort_to_indices(cols, d, f)
    for fetch in (1, 25, 1000, n // 16):
        got = ctx.sort_to_indices(dev, d, f, fetch=fetch).to_numpy()
        assert np.array_equal(got, full[:fetch]), f"fetch={fetch}"


@pytest.mark.parametrize("desc,nf_first", [(False, True), (True, False)])
def test_utf8_and_mixed_sort_keys(ctx, desc, nf_first):
    """Q1 sorts on two Utf8 columns (tpch/q1.slt.part SortExec [l_returnflag ASC, l_linestatus ASC]): byte-wise order, prefix first."""
    words = ["", "a", "ab", "abc", "b", "A", "R", "N", "O", "F", "zz", "日本", "a\x00", "ab\x00\x00"]
    for n, nf in [(0, 0), (1, 0), (300, 0.2), (20000, 0.05)]:
        a = pa.array([None if RNG.random() < nf else words[i] for i in RNG.integers(0, len(words), n)], type=pa.utf8())
        b = rand_array("int32", n, 0.1, RNG)
        c = pa.array([None if RNG.random() < nf else "x" * int(k) for k in RNG.integers(0, 40, n)], type=pa.utf8()).dictionary_encode()
        got = ctx.sort_to_indices([ctx.from_arrow(a)], [desc], [nf_first]).to_numpy()
        assert np.array_equal(got, po.lexsort_to_indices([a], [desc], [nf_first])), f"n={n}"
        got3 = ctx.sort_to_indices([ctx.from_arrow(c), ctx.from_arrow(a), ctx.from_arrow(b)], [desc, not desc, False], [nf_first, True, False]).to_numpy()
        assert np.array_equal(got3, po.lexsort_to_indices([c, a, b], [desc, not desc, False], [nf_first, True, False]))


def test_utf8_sort_key_too_long_reports_not_implemented(ctx):
    import dfgpu
    with pytest.raises(dfgpu.DfgpuError) as e:
        ctx.sort_to_indices([ctx.from_arrow(pa.array(["b" * 2000, "a"]))], [False], [True])
    assert e.value.kind == "NotImplemented"


@pytest.mark.parametrize("nparts", [1, 2, 3, 4, 8, 16, 200, 1000])
def test_hash_partition_matches_oracle(ctx, nparts):
    """BatchPartitioner::partition_iter (repartition/mod.rs:148-221): indices grouped by hash % n, input order kept; row conservation (:952-1031)."""
    for kinds, n in [(["int64"], 100000), (["int32", "utf8"], 5000), (["decimal"], 3000), (["int64"], 0), (["int64"], 1)]:
        from test_gpu_join import keycols
        cols = keycols(kinds, n, 0.1, 1000)
        idx, counts = ctx.hash_partition([ctx.from_arrow(c) for c in cols], nparts)
        oidx, ocounts = po.hash_partition(cols, nparts)
        assert counts == ocounts.tolist() and sum(counts) == n
        assert np.array_equal(idx.to_numpy(), oidx)


@pytest.mark.parametrize("shape", ["two_ints", "decimal_date_utf8", "many_ties"])
def test_large_input_sort_with_packed_moving_keys(ctx, shape):
    """From 2^20 rows on, the radix passes carry up to four key bytes with every row id instead of looking the digit up by row id; the
    row order must stay the oracle's stable lexsort (ties in input order) for multi-column keys, NULLs first / last and DESC columns,
    across several groups of four planes and a ragged last group."""
    rng = np.random.default_rng(3)
    n = (1 << 20) + 12345
    if shape == "two_ints":
        cols = [pa.array(rng.integers(0, 50, n).astype(np.int32), mask=rng.random(n) < 0.05), pa.array(rng.integers(-2**40, 2**40, n), mask=rng.random(n) < 0.05)]
        desc, nf = [False, True], [True, False]
    elif shape == "decimal_date_utf8":
        import decimal
        vals = [decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**7, 10**7, 5000)]
        cols = [pa.array([vals[i] for i in rng.integers(0, 5000, n)], type=pa.decimal128(15, 2)),
                pa.array(rng.integers(8000, 8060, n).astype(np.int32)).cast(pa.date32()),
                pa.array(np.array(["", "a", "ab", "b", "zz", "abc"], dtype=object)[rng.integers(0, 6, n)], type=pa.utf8())]
        desc, nf = [True, False, False], [True, True, False]
    else:
        cols = [pa.array(rng.integers(0, 3, n).astype(np.int8)), pa.array((rng.integers(0, 4, n) * 0.5).astype(np.float64))]
        desc, nf = [False, True], [True, True]
    got = ctx.sort_to_indices([ctx.from_arrow(c) for c in cols], desc, nf).to_numpy()
    assert np.array_equal(got, po.lexsort_to_indices(cols, desc, nf))


def test_large_sort_degenerate_keys(ctx):
    """2^20 rows exactly: all keys equal (no varying plane: the order is the input order), and a single varying byte."""
    n = 1 << 20
    same = pa.array(np.full(n, 7, dtype=np.int64))
    assert np.array_equal(ctx.sort_to_indices([ctx.from_arrow(same)], [True], [True]).to_numpy(), np.arange(n))
    one_byte = pa.array((np.arange(n) % 3).astype(np.int32))
    got = ctx.sort_to_indices([ctx.from_arrow(one_byte)], [True], [False]).to_numpy()
    assert np.array_equal(got, po.lexsort_to_indices([one_byte], [True], [False]))


@pytest.mark.parametrize("shape", ["decimal_desc_date", "ints_one_nullable", "floats_uint", "wide_range_falls_back"])
@pytest.mark.parametrize("fetch", [None, 700_000])
def test_sorted_key_columns_come_back_with_the_indices(ctx, shape, fetch):
    """dfgpu_sort_to_indices_keys: when packed key bits + row-number bits fit one 64-bit word, the passes move that word alone and the key columns are rebuilt
    from the sorted words.  The indices must stay the oracle's stable lexsort, and every returned column must equal take(column, indices) bit for bit
    (sort_batch, sorts/sort.rs:598-603); columns with NULLs and key sets too wide for the word come back as None and go through take()."""
    import decimal
    rng = np.random.default_rng(17)
    n = (1 << 20) + 4321
    if shape == "decimal_desc_date":
        cols = [pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-5 * 10**6, 5 * 10**6, n)], type=pa.decimal128(15, 2)), pa.array(rng.integers(8000, 10600, n).astype(np.int32)).cast(pa.date32())]
        desc, nf, produced = [True, False], [True, False], [True, True]
    elif shape == "ints_one_nullable":
        cols = [pa.array(rng.integers(-100, 100, n).astype(np.int16)), pa.array(rng.integers(0, 1000, n), mask=rng.random(n) < 0.1), pa.array(rng.integers(-2**10, 2**10, n).astype(np.int64))]
        desc, nf, produced = [False, True, True], [True, False, True], [True, False, True]
    elif shape == "floats_uint":
        cols = [pa.array((rng.integers(-8, 8, n) * 0.25).astype(np.float64)), pa.array(rng.integers(0, 60000, n).astype(np.uint16)), pa.array((rng.integers(-3, 3, n) * 1.5).astype(np.float32))]
        desc, nf, produced = [True, False, False], [True, True, True], [False, False, False]      # float bit patterns span more than 62 bits: the plane path, nothing produced
    else:
        cols = [pa.array(rng.integers(-2**45, 2**45, n)), pa.array(rng.integers(0, 2**20, n).astype(np.int32))]
        desc, nf, produced = [False, False], [True, True], [False, False]                         # 46 + 20 key bits + 21 row bits > 64: (key, row id) pairs move instead
    idx, sk = ctx.sort_to_indices_keys([ctx.from_arrow(c) for c in cols], desc, nf, fetch)
    want = po.lexsort_to_indices(cols, desc, nf)
    if fetch is not None:
        want = want[:fetch]
    got = idx.to_numpy()
    assert np.array_equal(got, want)
    assert [k is not None for k in sk] == produced
    for c, k in zip(cols, sk):
        if k is not None:
            assert k.to_arrow().equals(c.take(pa.array(want))), shape


def test_sort_exec_uses_sorted_key_columns(ctx):
    """SortExec over (payload, Decimal128 key DESC, Date32 key): the two key columns of the output are the rebuilt ones, the payload goes through take(); the batch
    equals pyarrow's take of the oracle's order."""
    import decimal
    from dfgpu import physical_plan as ops
    rng = np.random.default_rng(23)
    n = (1 << 20) + 99
    t = pa.table({"okey": pa.array(rng.integers(0, 10**9, n)), "price": pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(90000, 10**7, n)], type=pa.decimal128(15, 2)),
                  "d": pa.array(rng.integers(8035, 10560, n).astype(np.int32)).cast(pa.date32()), "tag": pa.array(rng.integers(0, 100, n).astype(np.int32), mask=rng.random(n) < 0.1)})
    batch = ops.batch_from_arrow(ctx, t)
    plan = ops.SortExec([ops.PhysicalSortExpr(ops.Column("price", 1), True, True), ops.PhysicalSortExpr(ops.Column("d", 2), False, False)], ops.MemoryExec([[batch]], batch.schema))
    out = pa.concat_tables([b.to_arrow() for b in plan.execute(0, ops.TaskContext(ctx, 8192))])
    order = po.lexsort_to_indices([t["price"].combine_chunks(), t["d"].combine_chunks()], [True, False], [True, False])
    assert out.equals(t.take(pa.array(order)))


@pytest.mark.parametrize("shape", ["ranges_hold", "outliers_outside_the_sample", "column_valid_only_off_the_sample", "estimate_off"])
def test_packed_sort_key_ranges_from_a_sample(ctx, shape):
    """From 2^22 rows on, the packed-key sort takes the value ranges from a sample (every n / 2^19-th row, widened), the encode pass checks every value against them and a
    miss costs the exact pass: a few extreme values on rows the sample does not visit; a column that is NULL wherever the sample looks.  The indices are numpy's stable
    lexsort either way, and the profile shows which passes ran."""
    rng = np.random.default_rng(17)
    n = (1 << 22) + 4321
    step = max(2, n >> 19)
    a = rng.integers(1000, 2000, n).astype(np.int64); b = rng.integers(0, 7, n).astype(np.int32)
    mask_b = None
    if shape == "outliers_outside_the_sample":
        a[step + 1] = -(10**12); a[5 * step + 3] = 10**12            # rows the strided sample does not visit
    if shape == "column_valid_only_off_the_sample":
        mask_b = (np.arange(n) % step) == 0                            # NULL on every sampled row
    if shape == "estimate_off":
        ctx.set_option("sort_estimate_ranges", 0)
    cols = [pa.array(a), pa.array(b, mask=mask_b)]
    ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read()
    try:
        got = ctx.sort_to_indices([ctx.from_arrow(c) for c in cols], [True, False], [True, False]).to_numpy()
        ks = set(ctx.profile_read())
    finally:
        ctx.profile_enable(False); ctx.set_option("sort_estimate_ranges", 1)
    assert np.array_equal(got, po.lexsort_to_indices(cols, [True, False], [True, False]))
    assert ("sort_key_sample" in ks) == (shape != "estimate_off")
    assert ("sort_key_ranges" in ks) == (shape != "ranges_hold")


@pytest.mark.parametrize("n", [1, 255, 4097, 300_001, 900_000])
def test_small_input_passes_in_one_launch_each_give_the_same_stable_order(ctx, n):
    """Below 2^20 rows every varying key byte is ONE launch (k_rs_plane_pass: offsets derived inside the scatter, the next pass's histogram counted by it) instead of
    histogram + scan + scatter.  Many equal keys (ties keep input order), a Decimal128 + Date32 + nullable Int32 key set (Q3's result shape plus NULLs): the indices equal the
    three-launch path's and numpy's stable lexsort."""
    import decimal
    rng = np.random.default_rng(n)
    rev = rng.integers(0, 50, n).astype(np.int64) * 10007                       # few distinct values: long runs of ties
    date = rng.integers(9000, 9040, n).astype(np.int32)
    prio = rng.integers(-3, 3, n).astype(np.int32); pmask = rng.random(n) < 0.1
    cols = [ctx.from_arrow(pa.array([decimal.Decimal(int(v)).scaleb(-4) for v in rev], type=pa.decimal128(38, 4))), ctx.from_arrow(pa.array(date, type=pa.int32()).cast(pa.date32())),
            ctx.from_arrow(pa.array(prio, mask=pmask))]
    desc, nf = [True, False, False], [True, False, True]
    got = ctx.sort_to_indices(cols, desc, nf).to_numpy()
    ctx.set_option("sort_fused_small_passes", 0)
    try:
        plain = ctx.sort_to_indices(cols, desc, nf).to_numpy()
    finally:
        ctx.set_option("sort_fused_small_passes", 1)
    assert np.array_equal(got, plain)
    pkey = np.where(pmask, np.int64(-(1 << 40)), prio.astype(np.int64))         # NULLs first in an ascending column
    want = np.lexsort((np.arange(n), pkey, date, -rev))                         # last key is the primary one; row number breaks ties = stable
    assert np.array_equal(got.astype(np.int64), want)


@pytest.mark.parametrize("rows_per_lane", [8, 16])
@pytest.mark.parametrize("shape", ["decimal_desc_date_nullable", "one_narrow_key", "skewed"])
def test_one_sweep_passes_equal_the_three_launch_passes(ctx, shape, rows_per_lane):
    """Word-mode sorts of 2^20 .. 2^30 rows run every LSD pass as one launch (k_os_pass: tiles by ticket, digit counts published per tile, look-back over the tiles in
    front) over histograms counted while the words are encoded.  The indices must be the three-launch passes' and numpy's stable lexsort: a ragged last tile, NULLs in both
    columns, a key of few bits (one short pass), and a skewed key whose top digits are constant over most waves (the one-add-per-wave branch of the histogram)."""
    rng = np.random.default_rng(29)
    n = (1 << 21) + 777 if shape != "one_narrow_key" else (1 << 20) + 1
    if shape == "decimal_desc_date_nullable":
        import decimal
        price = rng.integers(90000, 10494951, n); pm = rng.random(n) < 0.02
        date = rng.integers(8035, 10560, n).astype(np.int32); dm = rng.random(n) < 0.02
        cols = [pa.array([None if m else decimal.Decimal(int(v)).scaleb(-2) for v, m in zip(price, pm)], type=pa.decimal128(15, 2)), pa.array(date, mask=dm).cast(pa.date32())]
        desc, nf = [True, False], [True, False]
        pk = np.where(pm, np.int64(1 << 40), price); dk = np.where(dm, np.int64(1 << 40), date.astype(np.int64))
        want = np.lexsort((np.arange(n), dk, -pk))
    elif shape == "one_narrow_key":
        a = rng.integers(0, 11, n).astype(np.int32)
        cols = [pa.array(a)]; desc, nf = [False], [True]
        want = np.lexsort((np.arange(n), a))
    else:
        a = np.where(rng.random(n) < 0.999, 5, rng.integers(0, 1 << 30, n)).astype(np.int64); b = (rng.zipf(1.5, n) % 1000).astype(np.int32)
        cols = [pa.array(a), pa.array(b)]; desc, nf = [False, True], [True, True]
        want = np.lexsort((np.arange(n), -b.astype(np.int64), a))
    dcols = [ctx.from_arrow(c) for c in cols]
    ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read()
    try:
        ctx.set_option("sort_onesweep_rows", rows_per_lane)
        got = ctx.sort_to_indices(dcols, desc, nf).to_numpy()
        ks = set(ctx.profile_read())
        ctx.set_option("sort_onesweep_rows", 0)
        plain = ctx.sort_to_indices(dcols, desc, nf).to_numpy()
        kp = set(ctx.profile_read())
    finally:
        ctx.profile_enable(False); ctx.set_option("sort_onesweep_rows", 16)
    assert "sort_pass_onesweep" in ks and "sort_pass_scatter" not in ks and "sort_pass_scatter" in kp and "sort_pass_onesweep" not in kp
    assert np.array_equal(got, plain)
    assert np.array_equal(got.astype(np.int64), want)
    # the last pass writing row numbers and rebuilt key columns itself == the separate finishing pass, with and without a fetch
    for fetch in (None, 700_001):
        a_idx, a_keys = ctx.sort_to_indices_keys(dcols, desc, nf, fetch=fetch)
        ctx.set_option("sort_onesweep_fused_finish", 0)
        try:
            b_idx, b_keys = ctx.sort_to_indices_keys(dcols, desc, nf, fetch=fetch)
        finally:
            ctx.set_option("sort_onesweep_fused_finish", 1)
        assert np.array_equal(a_idx.to_numpy(), b_idx.to_numpy()) and np.array_equal(a_idx.to_numpy(), got[:fetch] if fetch else got)
        for x, y in zip(a_keys, b_keys):
            assert (x is None) == (y is None)
            if x is not None:
                assert x.to_arrow().equals(y.to_arrow())


@pytest.mark.parametrize("fetch", [1, 10, 5000, 60_000])
@pytest.mark.parametrize("shape", ["sum_desc_key", "many_ties", "nullable_float", "wide_keys_not_a_word"])
def test_topk_over_packed_words_equals_sort_then_slice(ctx, shape, fetch):
    """SortExec with fetch <= n / 16 over a large input whose keys pack with the row number into one word: radix select on the words (k_ws_hist per digit, the host picks the
    digit of the fetch-th word), the words up to the chosen prefix compacted and sorted on their own (by counting the words below each when they are at most 16384, by the
    LSD passes otherwise).  Result = the first `fetch` indices of the stable full sort; ties (equal keys) in row order; the option moves the threshold down to the test's size."""
    rng = np.random.default_rng(31 + fetch)
    n = (1 << 20) + 4099
    if shape == "sum_desc_key":
        a = rng.integers(0, 1 << 20, n).astype(np.int64); b = rng.integers(0, 1 << 18, n).astype(np.int64)              # 21 + 19 key bits + 21 row bits: one word
        cols = [pa.array(a), pa.array(b)]; desc, nf = [True, False], [True, False]
        want = np.lexsort((np.arange(n), b, -a))
    elif shape == "many_ties":
        a = rng.integers(0, 4, n).astype(np.int32)
        cols = [pa.array(a)]; desc, nf = [False], [True]
        want = np.lexsort((np.arange(n), a))
    elif shape == "wide_keys_not_a_word":             # 31 + 26 key bits + 21 row bits: keys and row numbers move apart, equal keys (about 20 rows each) keep row order
        a = rng.integers(0, 1 << 30, 1000)[rng.integers(0, 1000, n)].astype(np.int64); b = rng.integers(0, 1 << 25, 50)[rng.integers(0, 50, n)].astype(np.int64)
        cols = [pa.array(a), pa.array(b)]; desc, nf = [True, False], [True, False]
        want = np.lexsort((np.arange(n), b, -a))
    else:
        a = (rng.random(n) * 1000).astype(np.float32); am = rng.random(n) < 0.3
        cols = [pa.array(a, mask=am)]; desc, nf = [True], [False]
        bits = a.view(np.int32).astype(np.int64)                        # positive floats: the bit pattern orders like the value
        want = np.lexsort((np.arange(n), np.where(am, np.int64(-1), bits) * -1))      # DESC, NULLs last
    dcols = [ctx.from_arrow(c) for c in cols]
    ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read()
    try:
        ctx.set_option("sort_topk_words_min_rows", 1 << 20)
        got = ctx.sort_to_indices(dcols, desc, nf, fetch=fetch).to_numpy()
        ks = set(ctx.profile_read())
    finally:
        ctx.profile_enable(False); ctx.set_option("sort_topk_words_min_rows", 1 << 23)
    assert "sort_topk_words" in ks
    full = ctx.sort_to_indices(dcols, desc, nf).to_numpy()
    assert np.array_equal(got, full[:fetch])
    assert np.array_equal(got.astype(np.int64), want[:fetch])


@pytest.mark.parametrize("n", [2, 63, 255, 4097, 8192, 8193])
def test_sorts_of_one_workgroup_run_every_pass_in_one_launch(ctx, n):
    """Up to 8192 rows every varying key byte's pass runs inside ONE launch of one workgroup (k_rs_one_block: row ids in LDS, four barriers per pass): the indices are the
    two-launches-per-pass path's (option sort_one_block_max_rows = 0) and numpy's stable lexsort, for a Utf8 tie-break key with dozens of varying bytes behind a Float64 key with
    many ties, NULLs in both, DESC / NULLS LAST on the first; 8193 rows take the other path."""
    rng = np.random.default_rng(100 + n)
    avg = np.round(rng.random(n) * 3, 1); am = rng.random(n) < 0.1
    urls = np.array([f"https://site{int(k)}.example/{int(k) * 7919 % 1000}" for k in rng.integers(0, max(2, n // 3), n)], dtype=object); um = rng.random(n) < 0.05
    cols = [ctx.from_arrow(pa.array(avg, mask=am)), ctx.from_arrow(pa.array(urls, mask=um, type=pa.utf8()))]
    desc, nf = [True, False], [False, True]
    ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read()
    try:
        got = ctx.sort_to_indices(cols, desc, nf).to_numpy()
        ks = set(ctx.profile_read())
        ctx.set_option("sort_one_block_max_rows", 0)
        plain = ctx.sort_to_indices(cols, desc, nf).to_numpy()
        kp = set(ctx.profile_read())
    finally:
        ctx.profile_enable(False); ctx.set_option("sort_one_block_max_rows", 8192)
    assert ("radix_pass_one_block" in ks) == (n <= 8192) and "radix_pass_one_block" not in kp
    assert np.array_equal(got, plain)
    ukey = np.array([("" if m else u) for u, m in zip(urls, um)], dtype=object)
    order = sorted(range(n), key=lambda i: ((1 if am[i] else 0), -avg[i] if not am[i] else 0.0, (0 if um[i] else 1), ukey[i].encode(), i))
    assert np.array_equal(got.astype(np.int64), np.array(order, dtype=np.int64))


@pytest.mark.parametrize("fetch", [None, 900_000])
def test_sort_take_gathers_payload_columns_in_the_last_pass(ctx, fetch):
    """dfgpu_sort_take: fixed-width payload columns without NULLs (8, 4 and 16 bytes wide) come back in sorted order out of the sort's last pass; a nullable column, a Utf8 column
    and a fifth eligible column come back as None (the caller takes them); every returned column == take(column, indices), with the option off nothing is returned, and
    SortExec's output is the same either way."""
    import decimal
    from dfgpu import physical_plan as ops
    rng = np.random.default_rng(61)
    n = (1 << 20) + 555
    k1 = rng.integers(0, 5000, n).astype(np.int32); k2 = rng.integers(-2**27, 2**27, n).astype(np.int64)          # 13 + 29 key bits + 21 row bits: one word
    p8 = rng.integers(0, 2**60, n).astype(np.int64); p4 = rng.integers(0, 2**30, n).astype(np.int32)
    p16 = pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**9, 10**9, 2000)], type=pa.decimal128(20, 2)).take(pa.array(rng.integers(0, 2000, n)))
    pn = pa.array(rng.integers(0, 100, n).astype(np.int64), mask=rng.random(n) < 0.2)
    ps = pa.array(np.array(["a", "bb", "ccc"], dtype=object)[rng.integers(0, 3, n)], type=pa.utf8())
    f1 = rng.random(n); f2 = rng.random(n).astype(np.float32)
    keys = [ctx.from_arrow(pa.array(k1)), ctx.from_arrow(pa.array(k2))]
    pay_arrow = [pa.array(p8), pa.array(p4), p16, pn, ps, pa.array(f1), pa.array(f2)]
    pay = [ctx.from_arrow(a) for a in pay_arrow]
    desc, nf = [True, False], [True, True]
    ctx.set_option("sort_payload_in_last_pass", 1)                  # off by default (measured slower than the separate gather); the entry point's contract is tested with it on
    try:
        idx, sk, got = ctx.sort_take(keys, desc, nf, pay, fetch=fetch)
    finally:
        ctx.set_option("sort_payload_in_last_pass", 0)
    order = idx.to_numpy()
    want_order = np.lexsort((np.arange(n), k2, -k1.astype(np.int64)))[:fetch]
    assert np.array_equal(order.astype(np.int64), want_order)
    returned = [g is not None for g in got]
    assert returned == [True, True, True, False, False, True, False]          # the first four eligible columns; NULLs, strings and the fifth are the caller's
    for a, g in zip(pay_arrow, got):
        if g is not None:
            assert g.to_arrow().equals(a.take(pa.array(order)))
    idx0, _, got0 = ctx.sort_take(keys, desc, nf, pay, fetch=fetch)
    assert np.array_equal(idx0.to_numpy(), order) and all(g is None for g in got0)
    # through SortExec: the same batch either way
    t = pa.table({"k1": pa.array(k1), "k2": pa.array(k2), "p8": pa.array(p8), "pn": pn, "p16": p16, "ps": ps})
    batch = ops.batch_from_arrow(ctx, t)
    tc = ops.TaskContext(ctx, batch_size=8192)
    mk = lambda: ops.SortExec([ops.PhysicalSortExpr(ops.Column("k1", 0), True, True), ops.PhysicalSortExpr(ops.Column("k2", 1), False, True)], ops.MemoryExec([[batch]], batch.schema), fetch=fetch)
    outs = []
    for on in (1, 0):
        ctx.set_option("sort_payload_in_last_pass", on)
        try:
            outs.append(pa.concat_tables([b.to_arrow() for b in mk().execute(0, tc)]).combine_chunks())
        finally:
            ctx.set_option("sort_payload_in_last_pass", 0)
    assert outs[0].equals(outs[1]) and outs[0].equals(t.take(pa.array(want_order)).combine_chunks())


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_one_sweep_passes_over_random_sizes_and_key_widths(ctx, seed):
    """Random row counts between 2^20 and 2^22.2 (4096- and 8192-row tiles, ragged last tiles, a tile count that is and is not a multiple of 8), one to three key columns of
    random widths (1 .. 6 passes), random directions and NULL fractions: the one-launch passes give numpy's stable order and the three-launch passes' indices."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1 << 20, int(2 ** 22.2)))
    k = int(rng.integers(1, 4)); cols, desc, nf, keys_np = [], [], [], []
    budget = 64 - int(n - 1).bit_length()
    for c in range(k):
        bits = int(rng.integers(1, max(2, min(24, budget - 2 * (k - c)))));  budget -= bits + 1
        v = rng.integers(0, 1 << bits, n).astype(np.int64) - int(rng.integers(0, 1 << bits))
        m = rng.random(n) < float(rng.choice([0.0, 0.0, 0.05]))
        d, f = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        cols.append(pa.array(v, mask=m if m.any() else None)); desc.append(d); nf.append(f)
        sv = -v if d else v
        nullkey = np.where(m, 0 if f else 2, 1)                       # NULLs first or last whatever the direction
        keys_np.append((nullkey, np.where(m, 0, sv)))
    dcols = [ctx.from_arrow(c) for c in cols]
    ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read()
    try:
        got = ctx.sort_to_indices(dcols, desc, nf).to_numpy()
        ks = set(ctx.profile_read())
        ctx.set_option("sort_onesweep_rows", 0)
        plain = ctx.sort_to_indices(dcols, desc, nf).to_numpy()
    finally:
        ctx.profile_enable(False); ctx.set_option("sort_onesweep_rows", 16)
    assert "sort_pass_onesweep" in ks, (n, k)
    assert np.array_equal(got, plain)
    lex = [np.arange(n)]
    for nullkey, sv in reversed(keys_np):
        lex += [sv, nullkey]
    assert np.array_equal(got.astype(np.int64), np.lexsort(tuple(lex)))
