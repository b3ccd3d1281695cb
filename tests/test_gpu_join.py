"""-m gpu: HashJoinExec on device vs (1) the reference's own known-answer tables (tests/golden/hash_join.json) and
(2) the CPU oracle on seeded random inputs -- (build, probe) index pairs bit-exact INCLUDING ORDER."""
import numpy as np
import pyarrow as pa
import pytest

from helpers import load_golden, pa_types_for, rows_of, side_batches, sort_rows
from oracle import pyoracle as po
from test_gpu_core import rand_array

pytestmark = pytest.mark.gpu
CASES = load_golden("hash_join.json")["cases"]
RNG = np.random.default_rng(99)


def device_join(ctx, case, batch_size):
    import dfgpu
    from dfgpu import physical_plan as ops
    tc = ops.TaskContext(ctx, batch_size=batch_size)

    def side(name):
        names = case[name]["names"]
        batches = [ops.batch_from_arrow(ctx, pa.table(cols, names=names)) for cols in side_batches(case, name)]
        ts = pa_types_for(case, len(names))
        code = {pa.int32(): dfgpu.capi.INT32, pa.int64(): dfgpu.capi.INT64, pa.date32(): dfgpu.capi.DATE32}
        schema = ops.Schema([ops.Field(n, code[t]) for n, t in zip(names, ts)])
        parts = [[b] for b in batches] if "two_parts" in case["name"] else [batches]
        return ops.MemoryExec(parts, schema), schema

    left, ls = side("left")
    right, rs = side("right")
    on = [(ops.Column.new_with_schema(l, ls), ops.Column.new_with_schema(r, rs)) for l, r in case["on"]]
    filt = None
    if case["filter"]:
        spec = case["filter"]
        rhs = ops.Column("c", spec["rhs_column"]) if "rhs_column" in spec else ops.Literal(spec["rhs_literal"], pa.int32())
        filt = ops.JoinFilter(ops.BinaryExpr(ops.Column("x", 0), spec["op"], rhs), [tuple(ci) for ci in spec["column_indices"]],
                              ops.Schema([ops.Field("x", dfgpu.capi.INT32)] * len(spec["column_indices"])))
    join = ops.HashJoinExec(left, right, on, filt, case["join_type"], "CollectLeft", case["null_equals_null"])
    per_partition, n_batches = [], 0
    for p in range(join.output_partitioning().partition_count()):
        rows = []
        for b in join.execute(p, tc):
            rows += rows_of([c.to_arrow() for c in b.materialize().columns])
            n_batches += 1
        per_partition.append(rows)
    join.emitted_batches = n_batches
    return per_partition, join


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_hash_join_reference_golden(ctx, case):
    for bs in case["batch_sizes"]:
        parts, join = device_join(ctx, case, bs)
        if case["batch_count"]:         # the reference asserts the number of emitted RecordBatches per batch_size (hash_join.rs:3388-3412)
            assert join.emitted_batches == case["batch_count"][str(bs)], f"batch_size={bs}: {join.emitted_batches} batches"
        rows = [r for p in parts for r in p]
        if case["ordered"]:
            assert rows == case["expected"], f"batch_size={bs}"
        else:
            assert sort_rows(rows) == sort_rows(case["expected"]), f"batch_size={bs}"
        if case["per_partition_expected"]:
            assert parts == case["per_partition_expected"]
        want_names = case["left"]["names"] + case["right"]["names"]
        if case["join_type"] in ("LeftSemi", "LeftAnti"): want_names = case["left"]["names"]
        if case["join_type"] in ("RightSemi", "RightAnti"): want_names = case["right"]["names"]
        assert join.schema().names() == want_names


@pytest.mark.parametrize("case", [c for c in CASES if not c["name"].startswith("join_splitted")], ids=lambda c: c["name"])
def test_hash_join_golden_under_forced_collisions(ctx, case):
    """≙ CI job `cargo test --features force_hash_collisions` (.github/workflows/rust.yml:454-470)."""
    ctx.set_option("force_hash_collisions", 1)
    try:
        parts, _ = device_join(ctx, case, 8192)
        assert sort_rows([r for p in parts for r in p]) == sort_rows(case["expected"])
    finally:
        ctx.set_option("force_hash_collisions", 0)


def keycols(kinds, n, null_frac, card):
    cols = []
    for k in kinds:
        if k == "int64": cols.append(pa.array(RNG.integers(0, card, n).astype(np.int64), mask=RNG.random(n) < null_frac if null_frac else None))
        elif k == "int32": cols.append(pa.array(RNG.integers(-card // 2, card // 2, n).astype(np.int32), mask=RNG.random(n) < null_frac if null_frac else None))
        elif k == "utf8": cols.append(pa.array([None if RNG.random() < null_frac else f"k{v}" for v in RNG.integers(0, card, n)], type=pa.utf8()))
        elif k == "dict": cols.append(pa.array([None if RNG.random() < null_frac else f"k{v}" for v in RNG.integers(0, card, n)], type=pa.utf8()).dictionary_encode())
        elif k == "decimal":
            import decimal
            cols.append(pa.array([None if RNG.random() < null_frac else decimal.Decimal(int(v)).scaleb(-2) for v in RNG.integers(0, card, n)], type=pa.decimal128(15, 2)))
        elif k == "float64": cols.append(pa.array(RNG.integers(0, card, n).astype(np.float64), mask=RNG.random(n) < null_frac if null_frac else None))
    return cols


FUZZ = [(["int64"], 20000, 50000, 0.0, 30000), (["int64"], 5000, 20000, 0.1, 500), (["int32", "utf8"], 3000, 9000, 0.1, 40),
        (["decimal"], 4000, 8000, 0.05, 1000), (["dict", "int64"], 2000, 5000, 0.1, 30), (["float64"], 1000, 3000, 0.0, 100),
        (["int64"], 1, 1000, 0.0, 5), (["int64"], 1000, 1, 0.0, 5), (["int64"], 0, 100, 0.0, 5), (["int64"], 100, 0, 0.0, 5),
        (["int64"], 70000, 300000, 0.0, 10**9)]


@pytest.mark.parametrize("kinds,nb,npr,nf,card", FUZZ, ids=[f"{'+'.join(f[0])}-{f[1]}x{f[2]}" for f in FUZZ])
@pytest.mark.parametrize("nen", [False, True])
def test_probe_pairs_match_oracle_exactly(ctx, kinds, nb, npr, nf, card, nen):
    """≙ core/tests/fuzz_cases/join_fuzz.rs, but stricter: pair ORDER must equal the reference's (probe order, then build order)."""
    import dfgpu
    b, p = keycols(kinds, nb, nf, card), keycols(kinds, npr, nf, card)
    table = dfgpu.JoinTable(ctx, [ctx.from_arrow(c) for c in b], null_equals_null=nen)
    bi, pi = table.probe([ctx.from_arrow(c) for c in p])
    want = po.hash_join([b], [p], "Inner", nen, batch_size=1 << 40)
    assert np.array_equal(bi.to_numpy().astype(np.int64), want.build_idx)
    assert np.array_equal(pi.to_numpy().astype(np.int64), want.probe_idx)
    assert table.num_rows == nb and table.memory > 0


@pytest.mark.parametrize("kinds,nb,npr,nf,card", [(["int64", "int32"], 20000, 60000, 0.0, 300), (["int32", "int64", "int32"], 5000, 20000, 0.1, 40), (["int64", "int64"], 3000, 9000, 0.05, 10**9)],
                         ids=["2keys", "3keys-nulls", "2keys-wide-range"])
def test_multi_integer_key_packing_equals_unpacked(ctx, kinds, nb, npr, nf, card):
    """2..4 integer key columns with small ranges are packed into one Int64 key (join.hip k_pack_keys): pairs (order included) must equal
    the oracle's and the unpacked hash path's (option join_key_packing=0); probe tuples outside the build ranges and NULL components never
    match; ranges too wide to pack (10^9 x 10^9) keep the multi-column table."""
    import dfgpu
    b, p = keycols(kinds, nb, nf, card), keycols(kinds, npr, nf, card)
    p = [pa.concat_arrays([c, pa.array([10**6 + 7, -10**6 - 7], type=c.type)]) for c in p]          # components far outside the build ranges
    want = po.hash_join([b], [p], "Inner", False, batch_size=1 << 40)
    for packing in (1, 0):
        ctx.set_option("join_key_packing", packing)
        try:
            table = dfgpu.JoinTable(ctx, [ctx.from_arrow(c) for c in b])
            bi, pi = table.probe([ctx.from_arrow(c) for c in p])
        finally:
            ctx.set_option("join_key_packing", 1)
        assert np.array_equal(bi.to_numpy().astype(np.int64), want.build_idx), f"packing={packing}"
        assert np.array_equal(pi.to_numpy().astype(np.int64), want.probe_idx), f"packing={packing}"


def test_probe_with_fused_masks_equals_filtered_inputs(ctx):
    """FilterExec fused into build and probe: same pairs as compacting first (indices mapped back)."""
    import dfgpu
    nb, npr = 8000, 30000
    b, p = keycols(["int64"], nb, 0.0, 3000), keycols(["int64"], npr, 0.0, 3000)
    bm, pm = RNG.random(nb) < 0.6, RNG.random(npr) < 0.4
    table = dfgpu.JoinTable(ctx, [ctx.from_arrow(b[0])], mask=ctx.from_arrow(pa.array(bm)))
    bi, pi = table.probe([ctx.from_arrow(p[0])], mask=ctx.from_arrow(pa.array(pm)))
    want = po.hash_join([[b[0].filter(pa.array(bm))]], [[p[0].filter(pa.array(pm))]], "Inner", batch_size=1 << 40)
    bmap, pmap = np.flatnonzero(bm), np.flatnonzero(pm)
    assert np.array_equal(bi.to_numpy().astype(np.int64), bmap[want.build_idx])
    assert np.array_equal(pi.to_numpy().astype(np.int64), pmap[want.probe_idx])
    # unmatched build rows exclude rows dropped by the build-side filter
    table.mark_visited(bi)
    fin = table.final_indices(dfgpu.capi.JOIN_LEFT).to_numpy()
    visited = np.zeros(nb, bool); visited[bmap[want.build_idx]] = True
    assert np.array_equal(fin, np.flatnonzero(bm & ~visited))


def sorted_unique_keys(n, dtype, shape):
    """strictly increasing build keys: 'dense' = k0 + row (identity), 'sparse' = TPC-H style gaps, 'wide' = too sparse for a bitmap."""
    if shape == "dups": k = np.sort(RNG.integers(0, n // 3, n)) * 3 - 700          # non-decreasing with repeats (a sorted foreign key)
    elif shape == "dense": k = np.arange(n, dtype=np.int64) - 1000
    elif shape == "sparse": k = np.cumsum(RNG.integers(1, 9, n)) - 5000
    else: k = np.cumsum(RNG.integers(1, 10**7 if dtype == np.int64 else 50000, n))
    return k.astype(dtype)


@pytest.mark.parametrize("dtype", [np.int32, np.int64], ids=["int32", "int64"])
@pytest.mark.parametrize("shape", ["dense", "sparse", "wide", "dups"])
@pytest.mark.parametrize("masked", [False, True])
def test_rank_index_build_equals_hash_build_and_oracle(ctx, dtype, shape, masked):
    """Sorted integer build keys (strictly increasing, or -- "dups" -- non-decreasing with repeats, whose equal keys form contiguous
    runs) take the bitmap rank index (join.hip build_rank_index) instead of the hash table:
    pairs must equal the oracle's and the hash-table path's (option join_rank_index=0), probe NULLs / misses / selection included."""
    import dfgpu
    nb, npr = 40000, 150000
    bk = sorted_unique_keys(nb, dtype, shape)
    lo, hi = int(bk[0]) - 500, int(bk[-1]) + 500
    pk = RNG.integers(lo, hi, npr).astype(dtype)
    hit = RNG.random(npr) < 0.4
    pk[hit] = RNG.choice(bk, int(hit.sum()))
    b = pa.array(bk); p = pa.array(pk, mask=RNG.random(npr) < 0.05)
    bm = RNG.random(nb) < 0.5 if masked else None
    pm = RNG.random(npr) < 0.7 if masked else None
    got = []
    for rank in (1, 0):
        ctx.set_option("join_rank_index", rank)
        try:
            table = dfgpu.JoinTable(ctx, [ctx.from_arrow(b)], mask=ctx.from_arrow(pa.array(bm)) if masked else None)
            bi, pi = table.probe([ctx.from_arrow(p)], mask=ctx.from_arrow(pa.array(pm)) if masked else None)
            got.append((bi.to_numpy().astype(np.int64), pi.to_numpy().astype(np.int64)))
            table.mark_visited(bi)
            got[-1] += (table.final_indices(dfgpu.capi.JOIN_LEFT).to_numpy(),)
        finally:
            ctx.set_option("join_rank_index", 1)
    bsel, psel = (b.filter(pa.array(bm)), p.filter(pa.array(pm))) if masked else (b, p)
    want = po.hash_join([[bsel]], [[psel]], "Inner", batch_size=1 << 40)
    bmap = np.flatnonzero(bm) if masked else np.arange(nb)
    pmap = np.flatnonzero(pm) if masked else np.arange(npr)
    for bi, pi, fin in got:
        assert np.array_equal(bi, bmap[want.build_idx])
        assert np.array_equal(pi, pmap[want.probe_idx])
        visited = np.zeros(nb, bool); visited[bmap[want.build_idx]] = True
        assert np.array_equal(fin, np.flatnonzero((bm if masked else np.ones(nb, bool)) & ~visited))


def test_rank_index_falls_back_to_hash_for_dictionary_probe_keys(ctx):
    """A rank-indexed build probed by a dictionary-encoded column of the same logical type builds its hash table lazily."""
    import dfgpu
    bk = np.arange(0, 3000, 3, dtype=np.int64)
    table = dfgpu.JoinTable(ctx, [ctx.from_arrow(pa.array(bk))])
    plain = pa.array(RNG.integers(-10, 3100, 20000).astype(np.int64), mask=RNG.random(20000) < 0.1)
    bi0, pi0 = table.probe([ctx.from_arrow(plain)])
    bi1, pi1 = table.probe([ctx.from_arrow(plain.dictionary_encode())])
    bi2, pi2 = table.probe([ctx.from_arrow(plain)])                       # rank path again after the fallback
    want = po.hash_join([[pa.array(bk)]], [[plain]], "Inner", batch_size=1 << 40)
    for bi, pi in ((bi0, pi0), (bi1, pi1), (bi2, pi2)):
        assert np.array_equal(bi.to_numpy().astype(np.int64), want.build_idx)
        assert np.array_equal(pi.to_numpy().astype(np.int64), want.probe_idx)


@pytest.mark.parametrize("jt", ["Inner", "Left", "Right", "Full", "LeftSemi", "RightSemi", "LeftAnti", "RightAnti"])
def test_join_types_fuzz_vs_oracle(ctx, task_ctx, jt):
    """All 8 join types over multi-batch inputs with NULL keys and duplicate keys; multiset of output rows equals the oracle's."""
    import dfgpu
    from dfgpu import physical_plan as ops
    lb = [pa.table({"k": keycols(["int64"], n, 0.1, 50)[0], "v": pa.array(RNG.integers(0, 10**6, n))}) for n in (300, 1, 500)]
    rb = [pa.table({"k": keycols(["int64"], n, 0.1, 70)[0], "w": pa.array(RNG.integers(0, 10**6, n))}) for n in (700, 64, 200)]
    mk = lambda tabs: ops.MemoryExec([[ops.batch_from_arrow(ctx, t) for t in tabs]], ops.batch_from_arrow(ctx, tabs[0]).schema)
    left, right = mk(lb), mk(rb)
    join = ops.HashJoinExec(left, right, [(ops.Column("k", 0), ops.Column("k", 0))], None, jt)
    got = []
    for b in ops.collect(join, task_ctx):
        got += rows_of([c.to_arrow() for c in b.columns])
    res = po.hash_join([[t["k"]] for t in lb], [[t["k"]] for t in rb], jt, batch_size=8192)
    lcat = pa.concat_tables(lb[::-1]); want = []
    for bi, pi, pb in zip(res.build_idx, res.probe_idx, res.probe_batch):
        l = [None, None] if bi < 0 else [lcat["k"][int(bi)].as_py(), lcat["v"][int(bi)].as_py()]
        r = [None, None] if pi < 0 else [rb[pb]["k"][int(pi)].as_py(), rb[pb]["w"][int(pi)].as_py()]
        want.append(l if jt in ("LeftSemi", "LeftAnti") else r if jt in ("RightSemi", "RightAnti") else l + r)
    assert sort_rows(got) == sort_rows(want)
    if jt in ("Inner", "RightSemi", "RightAnti"):        # probe-order preserving join types: exact order
        assert got == want


def test_join_key_type_mismatch_is_rejected(ctx):
    import dfgpu
    t = dfgpu.JoinTable(ctx, [ctx.from_arrow(pa.array([1, 2, 3], type=pa.int64()))])
    with pytest.raises(dfgpu.DfgpuError):
        t.probe([ctx.from_arrow(pa.array([1, 2, 3], type=pa.int32()))])


def test_join_requires_on_columns(ctx):
    import dfgpu
    from dfgpu import physical_plan as ops
    s = ops.Schema([ops.Field("a", dfgpu.capi.INT32)])
    with pytest.raises(dfgpu.DfgpuError):     # hash_join.rs:303-305
        ops.HashJoinExec(ops.MemoryExec([[]], s), ops.MemoryExec([[]], s), [], None, "Inner")


@pytest.mark.parametrize("mode", ["CollectLeft", "Partitioned"])
@pytest.mark.parametrize("with_filter", [False, True])
@pytest.mark.parametrize("jt", ["LeftSemi", "LeftAnti"])
def test_left_semi_anti_with_small_right_side_keeps_reference_order(ctx, task_ctx, jt, with_filter, mode):
    """A LeftSemi / LeftAnti join whose right input is much smaller than its left one indexes the RIGHT side on the device and probes
    with the collected left batch; rows and their order (ascending index of the reversed-batch concatenation, hash_join.rs:746,764;
    joins/utils.rs:1119-1141) must stay the reference's, with NULL keys, duplicate keys, a join filter and a multi-batch left input."""
    import dfgpu
    from dfgpu import physical_plan as ops
    rng = np.random.default_rng(5)
    lb = [pa.table({"k": pa.array(rng.integers(0, 400, n), mask=rng.random(n) < 0.05), "v": pa.array(rng.integers(0, 1000, n))}) for n in (4000, 1, 6000)]
    rb = [pa.table({"k": pa.array(rng.integers(0, 400, n), mask=rng.random(n) < 0.2), "w": pa.array(rng.integers(0, 1000, n))}) for n in (20, 5)]
    mk = lambda tabs: ops.MemoryExec([[ops.batch_from_arrow(ctx, t) for t in tabs]], ops.batch_from_arrow(ctx, tabs[0]).schema)
    filt = None
    if with_filter:
        filt = ops.JoinFilter(ops.BinaryExpr(ops.Column("x", 0), ">", ops.Column("y", 1)), [("left", 1), ("right", 1)],
                              ops.Schema([ops.Field("x", dfgpu.capi.INT64), ops.Field("y", dfgpu.capi.INT64)]))
    join = ops.HashJoinExec(mk(lb), mk(rb), [(ops.Column("k", 0), ops.Column("k", 0))], filt, jt, mode)
    got = []
    for b in ops.collect(join, task_ctx):
        got += rows_of([c.to_arrow() for c in b.columns])
    lcat = pa.concat_tables(lb[::-1])
    lv = lcat["v"].to_numpy()
    fn = (lambda pb, bi, pi: (lv[bi] > rb[pb]["w"].to_numpy()[pi]).astype(np.uint8)) if with_filter else None
    res = po.hash_join([[t["k"]] for t in lb], [[t["k"]] for t in rb], jt, batch_size=8192, filter_fn=fn)
    want = [[lcat["k"][int(bi)].as_py(), lcat["v"][int(bi)].as_py()] for bi in res.build_idx]
    assert len(want) > 0 and got == want
    # the same plan with the swap disabled is the build-on-left path: identical rows in identical order
    ctx.set_option("join_swap_small_semi", 0)
    try:
        plain = []
        for b in ops.collect(ops.HashJoinExec(mk(lb), mk(rb), [(ops.Column("k", 0), ops.Column("k", 0))], filt, jt, mode), task_ctx):
            plain += rows_of([c.to_arrow() for c in b.columns])
    finally:
        ctx.set_option("join_swap_small_semi", 1)
    assert plain == want


@pytest.mark.parametrize("dups", [False, True])
def test_sparse_small_build_gets_its_bitmap_from_the_first_large_probe(ctx, dups):
    """A build side tiny against its key range (here 60 keys over a 3 M range) starts as a plain hash table; a probe batch of at least
    range / 16 rows builds the membership bitmap on arrival.  Results before, at and after that probe equal the oracle's."""
    import dfgpu
    rng = np.random.default_rng(11)
    bk = np.sort(rng.choice(3_000_000, 60, replace=False)).astype(np.int64)[::-1].copy()          # not sorted ascending: no rank index
    if dups:
        bk = np.concatenate([bk, bk[:7]])
    bkeys = pa.array(bk, mask=np.arange(len(bk)) % 13 == 5)
    table = dfgpu.JoinTable(ctx, [ctx.from_arrow(bkeys)])
    def probe(n):
        pk = rng.integers(0, 3_000_000, n)
        pk[rng.random(n) < 0.3] = rng.choice(bk, 1)[0]
        pk[::7] = rng.choice(bk, len(pk[::7]))
        parr = pa.array(pk.astype(np.int64), mask=rng.random(n) < 0.05)
        bi, pi = table.probe([ctx.from_arrow(parr)])
        want = po.hash_join([[bkeys]], [[parr]], "Inner", batch_size=1 << 40)
        assert len(want.build_idx) > 0
        assert np.array_equal(bi.to_numpy().astype(np.int64), want.build_idx) and np.array_equal(pi.to_numpy().astype(np.int64), want.probe_idx)
    probe(1000)
    probe(400_000)
    probe(1000)


def test_identity_index_arrays_are_recognised_and_skipped(ctx, task_ctx):
    """A compaction that keeps every row and the probe indices of a join whose probe rows all match once are 0 .. n-1: the producer marks
    them (dfgpu_array_is_identity), dfgpu_take through them hands back the values array, and the plan layer passes probe-side columns
    through instead of gathering them.  Results are unchanged (oracle), a selective probe is not marked."""
    import dfgpu
    from dfgpu import physical_plan as ops
    lib = ctx.lib
    n = 5000
    all_true = ctx.mask_to_indices(ctx.from_arrow(pa.array(np.ones(n, dtype=bool))))
    some = ctx.mask_to_indices(ctx.from_arrow(pa.array(np.arange(n) % 7 != 0)))
    assert lib.dfgpu_array_is_identity(all_true.h) == 1 and lib.dfgpu_array_is_identity(some.h) == 0
    vals = pa.array(RNG.integers(0, 10**6, n), mask=RNG.random(n) < 0.1)
    assert ctx.take(ctx.from_arrow(vals), all_true).to_arrow().equals(vals)
    bk = np.arange(0, 3 * 400, 3, dtype=np.int64)
    table = dfgpu.JoinTable(ctx, [ctx.from_arrow(pa.array(bk))])
    pk_all, pk_some = RNG.choice(bk, n), RNG.integers(0, 1200, n).astype(np.int64)
    _, pi = table.probe([ctx.from_arrow(pa.array(pk_all))])
    assert lib.dfgpu_array_is_identity(pi.h) == 1 and np.array_equal(pi.to_numpy(), np.arange(n))
    _, pi = table.probe([ctx.from_arrow(pa.array(pk_some))])
    assert lib.dfgpu_array_is_identity(pi.h) == 0
    lt = pa.table({"k": pa.array(bk), "v": pa.array(RNG.integers(0, 100, len(bk)))})
    rt = pa.table({"k": pa.array(pk_all), "w": pa.array(RNG.integers(0, 100, n)), "s": pa.array([f"row{i}" for i in range(n)])})
    mk = lambda t: ops.MemoryExec([[ops.batch_from_arrow(ctx, t)]], ops.batch_from_arrow(ctx, t).schema)
    join = ops.HashJoinExec(mk(lt), mk(rt), [(ops.Column("k", 0), ops.Column("k", 0))], None, "Inner")
    got = []
    for b in ops.collect(ops.FilterExec(ops.BinaryExpr(ops.Column("w", 3), ">=", ops.Literal(0, pa.int64())), join), task_ctx):      # a filter that keeps every row on top
        got += rows_of([c.to_arrow() for c in b.columns])
    res = po.hash_join([[lt["k"]]], [[rt["k"]]], "Inner", batch_size=1 << 40)
    want = [[lt["k"][int(bi)].as_py(), lt["v"][int(bi)].as_py(), rt["k"][int(p)].as_py(), rt["w"][int(p)].as_py(), rt["s"][int(p)].as_py()] for bi, p in zip(res.build_idx, res.probe_idx)]
    assert got == want


@pytest.mark.parametrize("jt", ["Right", "Full", "RightSemi", "RightAnti", "Inner", "LeftSemi"])
def test_joins_over_a_filtered_probe_side_above_batch_size(ctx, jt):
    """Filter -> CoalesceBatches -> HashJoin with >= batch_size probe rows keeps the filter as a fused selection; the join types that emit
    unmatched probe rows (adjust_indices_by_join_type, joins/utils.rs:1234-1279) must not bring dropped rows back as unmatched ones.
    Rows must equal the same join over a probe side filtered beforehand (and the oracle's pair count for Inner)."""
    from dfgpu import physical_plan as ops
    nb, npr = 3000, 40000
    left = pa.table({"k": pa.array(RNG.integers(0, 2500, nb)), "a": pa.array(np.arange(nb, dtype=np.int64))})
    right = pa.table({"k": pa.array(RNG.integers(0, 5000, npr)), "v": pa.array(np.arange(npr, dtype=np.int64))})
    keep = np.asarray(right["v"]) < 30000
    tc = ops.TaskContext(ctx, batch_size=8192)
    C, L, B = ops.Column, ops.Literal, ops.BinaryExpr

    def rows(plan):
        out = []
        for p in range(plan.output_partitioning().partition_count()):
            for b in plan.execute(p, tc):
                out += rows_of([c.to_arrow() for c in b.materialize().columns])
        return sort_rows(out)
    lb = ops.batch_from_arrow(ctx, left); rb = ops.batch_from_arrow(ctx, right); rf = ops.batch_from_arrow(ctx, right.filter(pa.array(keep)))
    on = [(C("k", 0), C("k", 0))]
    fused = ops.HashJoinExec(ops.MemoryExec([[lb]], lb.schema), ops.CoalesceBatchesExec(ops.FilterExec(B(C("v", 1), "<", L(30000, pa.int64())), ops.MemoryExec([[rb]], rb.schema)), 8192),
                             on, None, jt, "CollectLeft")
    plain = ops.HashJoinExec(ops.MemoryExec([[lb]], lb.schema), ops.MemoryExec([[rf]], rf.schema), on, None, jt, "CollectLeft")
    got, want = rows(fused), rows(plain)
    assert got == want and len(got) > 0
    if jt == "Inner":
        assert len(got) == len(po.hash_join([[left["k"].combine_chunks()]], [[right["k"].combine_chunks().filter(pa.array(keep))]], "Inner", False, batch_size=1 << 40).probe_idx)


def test_probe_key_column_sliced_at_an_odd_row_offset(ctx):
    """RecordBatch::slice of a non-null fixed-width column is a pointer-offset view (dfgpu_array_slice): an Int64 key column starting at an
    odd row is 8-byte but not 16-byte aligned, and the bitmap probe reads two keys per load -- pairs must still equal the oracle's."""
    import dfgpu
    nb, npr = 4000, 60001
    b = np.arange(nb, dtype=np.int64) * 3 + 5                      # strictly increasing dense keys: rank index + bitmap probe
    p = RNG.integers(0, 3 * nb + 10, npr).astype(np.int64)
    table = dfgpu.JoinTable(ctx, [ctx.from_arrow(pa.array(b))])
    for off in (1, 3, 64, 65):
        sliced = ctx.from_arrow(pa.array(p)).slice(off, npr - off)
        bi, pi = table.probe([sliced])
        want = po.hash_join([[pa.array(b)]], [[pa.array(p[off:])]], "Inner", False, batch_size=1 << 40)
        assert np.array_equal(bi.to_numpy().astype(np.int64), want.build_idx) and np.array_equal(pi.to_numpy().astype(np.int64), want.probe_idx)


@pytest.mark.parametrize("shape", ["sorted_keys", "unsorted_unique_keys", "masked_build", "repeated_keys", "nullable_probe"])
def test_deferred_probe_and_lookup_at_the_abi(ctx, shape):
    """dfgpu_join_probe_deferred / dfgpu_join_lookup: a unique rank-indexed build leaves the build indices out and gives them later for any subset of the matched probe
    rows; every other table (repeated keys -> CSR, a nullable probe column) answers at once, exactly as dfgpu_join_probe.  Both ways equal the plain probe."""
    import dfgpu
    rng = np.random.default_rng(len(shape))
    nb, npr = 30_000, 100_000
    b = np.arange(nb, dtype=np.int64) * 2 + 10
    if shape == "unsorted_unique_keys":
        b = rng.permutation(b)
    if shape == "repeated_keys":
        b[5] = b[4]
    bmask = ctx.from_arrow(pa.array(rng.random(nb) < 0.8)) if shape == "masked_build" else None
    p = rng.integers(0, 2 * nb + 40, npr).astype(np.int64)
    pa_p = pa.array(p, mask=(rng.random(npr) < 0.05) if shape == "nullable_probe" else None)
    table = dfgpu.JoinTable(ctx, [ctx.from_arrow(pa.array(b))], mask=bmask)
    probe = [ctx.from_arrow(pa_p)]
    bi, pi = table.probe(probe)
    dbi, dpi = table.probe_deferred(probe)
    assert np.array_equal(dpi.to_numpy(), pi.to_numpy()) if dbi is None or shape != "repeated_keys" else True
    if shape in ("repeated_keys", "nullable_probe"):
        assert dbi is not None and np.array_equal(dbi.to_numpy(), bi.to_numpy()) and np.array_equal(dpi.to_numpy(), pi.to_numpy())
        return
    assert dbi is None
    assert np.array_equal(table.lookup(probe, dpi).to_numpy(), bi.to_numpy())                       # every matched row
    sub = np.sort(rng.choice(len(pi), 777, replace=False))
    rows = ctx.from_arrow(pa.array(pi.to_numpy()[sub].astype(np.uint32)))
    assert np.array_equal(table.lookup(probe, rows).to_numpy(), bi.to_numpy()[sub])                # a subset, in the subset's order
    with pytest.raises(dfgpu.DfgpuError):
        dfgpu.JoinTable(ctx, [ctx.from_arrow(pa.array(["a", "b"]))]).lookup([ctx.from_arrow(pa.array(["a"]))])      # a table that cannot locate a row from the key alone


def test_lookup_keeps_the_nulls_of_its_row_list(ctx):
    """dfgpu_join_lookup over a `rows` array with NULL entries (the NULL side of an outer join's indices above a deferred Inner join): the answer is NULL there and
    the build row everywhere else; a NULL entry's value (anything, even beyond the probe column) is never used as a row."""
    import dfgpu
    rng = np.random.default_rng(11)
    nb, npr = 20_000, 70_000
    b = np.arange(nb, dtype=np.int64) * 2 + 10
    p = b[rng.integers(0, nb, npr)]
    table = dfgpu.JoinTable(ctx, [ctx.from_arrow(pa.array(b))])
    probe = [ctx.from_arrow(pa.array(p))]
    bi, pi = table.probe(probe)
    dbi, dpi = table.probe_deferred(probe)
    assert dbi is None
    want = bi.to_numpy()
    pick = rng.integers(0, npr, 5000)
    null = rng.random(5000) < 0.3
    vals = pick.astype(np.uint32)
    vals[null] = 0xFFFFFFF0                                   # garbage under the NULLs
    got = table.lookup(probe, ctx.from_arrow(pa.array(vals, mask=null))).to_arrow()
    assert got.null_count == int(null.sum())
    assert got.to_pylist() == [None if nl else int(want[r]) for r, nl in zip(pick, null)]


@pytest.mark.parametrize("jt", ["Inner", "Right", "Left"])
def test_chunked_probe_reads_back_once_per_probe_batch(ctx, jt):
    """A reference-sized probe batch whose rows each match many build rows comes out in `batch_size`-pair chunks (process_probe_batch, hash_join.rs:1238-1343); the
    probe row each chunk ends on is read back once for the whole probe batch (the reference resumes from an in-memory offset, :1332-1340), not once per chunk.
    Pairs, their order and the batch boundaries equal the oracle's at the same batch_size."""
    from dfgpu import physical_plan as ops
    rng = np.random.default_rng(21)
    nkeys, per, npr, bsz = 40, 150, 6000, 1024
    bk = np.repeat(np.arange(nkeys, dtype=np.int64), per); rng.shuffle(bk)
    pk = rng.integers(0, nkeys + 8, npr).astype(np.int64)                     # a few probe keys without a match
    lb = ops.batch_from_arrow(ctx, pa.table({"k": pa.array(bk), "b": pa.array(np.arange(len(bk), dtype=np.int64))}))
    rb = ops.batch_from_arrow(ctx, pa.table({"k": pa.array(pk), "p": pa.array(np.arange(npr, dtype=np.int64))}))
    tc = ops.TaskContext(ctx, batch_size=bsz)
    join = ops.HashJoinExec(ops.MemoryExec([[lb]], lb.schema), ops.MemoryExec([[rb]], rb.schema), [(ops.Column("k", 0), ops.Column("k", 0))], None, jt, "CollectLeft")
    ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read()
    try:
        out = [b for b in join.execute(0, tc)]
        ctx.synchronize()
        prof = ctx.profile_read()
    finally:
        ctx.profile_enable(False)
    want = po.hash_join([[pa.array(bk)]], [[pa.array(pk)]], jt, False, batch_size=bsz)
    got_b = np.concatenate([np.asarray(b.columns[1].to_arrow().fill_null(-1)) for b in out])
    got_p = np.concatenate([np.asarray(b.columns[3].to_arrow().fill_null(-1)) for b in out])
    assert np.array_equal(got_b, want.build_idx) and np.array_equal(got_p, want.probe_idx)
    assert [b.num_rows for b in out] == np.diff(want.batch_offsets).tolist()    # the reference's batch boundaries
    pairs = int((pk < nkeys).sum()) * per
    assert len(out) >= pairs // bsz                                            # hundreds of chunks ...
    syncs = sum(v[0] for k, v in prof.items() if k.startswith("sync:"))
    by_cause = {k: v[0] for k, v in prof.items() if k.startswith("sync:")}
    if jt == "Inner":
        assert syncs <= 12, (syncs, by_cause)       # ... and a handful of read-backs (build, probe, the chunk ends), not one per chunk
    else:                   # Left marks the visited build rows of every chunk (one flag check per emitted batch), Right counts the unmatched probe rows of every chunk's row range
        assert syncs <= len(out) + 12, (syncs, by_cause)


@pytest.mark.parametrize("shape", ["unclustered", "unclustered_masked_nullable", "clustered", "many_matches"])
def test_bitmap_probe_by_key_range_of_unclustered_keys(ctx, shape):
    """Probe keys in no order against a build whose membership bitmap is larger than an L2 (pjoin.hip bp_probe): the batch is split by key range and every partition tests its
    slice of the bitmap; clustered keys and batches with many matches keep the streaming probe (the sample decides).  Pairs equal the oracle's, in order, either way."""
    import dfgpu
    rng = np.random.default_rng(len(shape))
    nb, step, npr = 4_300_000, 16, 3_000_000
    b = np.arange(nb, dtype=np.int64) * step + 5                                  # sorted unique keys over a 68.8 M domain: 8.6 MB of bitmap
    if shape == "many_matches":
        p = b[rng.integers(0, nb, npr)]
    else:
        p = rng.integers(0, nb * step + 100, npr).astype(np.int64)
    if shape == "clustered":
        p = np.sort(p)
    mask = valid = None
    if shape == "unclustered_masked_nullable":
        mask = rng.random(npr) < 0.6; valid = rng.random(npr) < 0.9
    table = dfgpu.JoinTable(ctx, [ctx.from_arrow(pa.array(b))])
    pa_p = pa.array(p, mask=None if valid is None else ~valid)
    ctx.set_option("join_bitmap_partitioned", 1); ctx.set_option("join_bitmap_partitioned_min_rows", 1 << 20)       # off by default (measured slower than the random probes at SF100)
    ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read()
    try:
        bi, pi = table.probe([ctx.from_arrow(pa_p)], mask=None if mask is None else ctx.from_arrow(pa.array(mask)))
        prof = ctx.profile_read()
    finally:
        ctx.profile_enable(False); ctx.set_option("join_bitmap_partitioned_min_rows", 1 << 24); ctx.set_option("join_bitmap_partitioned", 0)
    assert ("bp_probe" in prof) == shape.startswith("unclustered"), sorted(prof)
    ok = np.ones(npr, bool) if mask is None else (mask & valid)
    pos = np.searchsorted(b, p); pos[pos >= nb] = nb - 1
    hit = ok & (b[pos] == p)
    assert np.array_equal(pi.to_numpy().astype(np.int64), np.nonzero(hit)[0]) and np.array_equal(bi.to_numpy().astype(np.int64), pos[hit])


@pytest.mark.parametrize("shape", ["sorted_keys", "masked_build", "repeated_keys", "nullable_probe", "masked_probe"])
def test_probe_selection_is_the_match_bitmap_of_the_plain_probe(ctx, shape):
    """dfgpu_join_probe_selection: for a unique rank-indexed build the Inner join's probe side as a Boolean column (selected AND matching), nothing read back; any other table
    (repeated keys, a nullable probe column) declines before doing work.  Its set bits are the probe indices of dfgpu_join_probe."""
    import dfgpu
    rng = np.random.default_rng(len(shape) + 40)
    nb, npr = 30_000, 100_003
    b = np.arange(nb, dtype=np.int64) * 3 + 10
    if shape == "repeated_keys":
        b[7] = b[6]
    bmask = ctx.from_arrow(pa.array(rng.random(nb) < 0.7)) if shape == "masked_build" else None
    p = rng.integers(0, 3 * nb + 40, npr).astype(np.int64)
    pa_p = pa.array(p, mask=(rng.random(npr) < 0.05) if shape == "nullable_probe" else None)
    pmask = ctx.from_arrow(pa.array(rng.random(npr) < 0.5)) if shape == "masked_probe" else None
    table = dfgpu.JoinTable(ctx, [ctx.from_arrow(pa.array(b))], mask=bmask)
    probe = [ctx.from_arrow(pa_p)]
    sel = table.probe_selection(probe, mask=pmask)
    if shape in ("repeated_keys", "nullable_probe"):
        assert sel is None
        return
    bi, pi = table.probe(probe, mask=pmask)
    got = sel.to_arrow()
    assert got.null_count == 0 and len(got) == npr
    assert np.array_equal(np.nonzero(np.asarray(got))[0], pi.to_numpy().astype(np.int64))
