"""-m gpu: the radix-partitioned hash join (csrc/pjoin.hip + radix_partition.h) against the CPU oracle.

The (build, probe) index pairs must equal the oracle's bit for bit INCLUDING ORDER (probe order, then build order: hash_join.rs:161-197,
asserted :1593-1594), whatever the number of partitions, with NULL keys, fused selections on either side and every integer key width.
Thresholds are lowered through the ctx options so that small inputs take the partitioned path; each test also checks that the path really
ran (its kernels show up in the profile) or really fell back (repeated build keys, an over-full partition)."""
import numpy as np
import pyarrow as pa
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(4242)
M64 = (1 << 64) - 1


def mix64(x: int) -> int:
    x = (x + 0x9e3779b97f4a7c15) & M64
    x = ((x ^ (x >> 30)) * 0xbf58476d1ce4e5b9) & M64
    x = ((x ^ (x >> 27)) * 0x94d049bb133111eb) & M64
    return x ^ (x >> 31)


class forced:
    """ctx options that send small inputs down the partitioned path"""

    def __init__(self, ctx, rows_per_partition=12800, on=1):
        self.ctx, self.opts = ctx, {"join_partitioned": on, "join_partitioned_min_build": 1, "join_partitioned_min_probe": 1, "join_partition_rows": rows_per_partition,
                                    "join_rank_index_unsorted": 0}          # unique keys over a dense domain (the narrow integer types here) would take the rank index first

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


def unique_keys(n, dtype=np.int64, lo=-(1 << 62), hi=1 << 62):
    k = np.unique(RNG.integers(lo, hi, int(n * 1.1) + 8, dtype=np.int64))
    RNG.shuffle(k)
    assert len(k) >= n
    return k[:n].astype(dtype)


def probe_keys(b, n, frac, dtype=np.int64, lo=-(1 << 62), hi=1 << 62):
    p = RNG.integers(lo, hi, n, dtype=np.int64).astype(dtype)
    if len(b) and n:
        hit = RNG.random(n) < frac
        p[hit] = b[RNG.integers(0, len(b), int(hit.sum()))]
    return p


def check(ctx, b, p, bmask=None, pmask=None, bnull=None, pnull=None, typ=None):
    import dfgpu
    ba = pa.array(b, type=typ, mask=bnull); pa_ = pa.array(p, type=typ, mask=pnull)
    table = dfgpu.JoinTable(ctx, [ctx.from_arrow(ba)], mask=ctx.from_arrow(pa.array(bmask)) if bmask is not None else None)
    bi, pi = table.probe([ctx.from_arrow(pa_)], mask=ctx.from_arrow(pa.array(pmask)) if pmask is not None else None)
    # oracle over the compacted inputs, indices mapped back
    bsel = np.arange(len(b)) if bmask is None else np.flatnonzero(bmask)
    psel = np.arange(len(p)) if pmask is None else np.flatnonzero(pmask)
    want = po.hash_join([[ba.take(pa.array(bsel))]], [[pa_.take(pa.array(psel))]], "Inner", False, batch_size=1 << 40)
    assert np.array_equal(bi.to_numpy().astype(np.int64), bsel[want.build_idx])
    assert np.array_equal(pi.to_numpy().astype(np.int64), psel[want.probe_idx])
    return len(want.probe_idx)


SHAPES = [(50000, 200000, 0.2, 12800), (50000, 200000, 0.2, 64), (20000, 70001, 1.0, 100), (3000, 100000, 0.0, 16), (1, 5000, 0.5, 12800), (2, 5000, 0.5, 12800),
          (4097, 4096, 0.3, 33), (130000, 400000, 0.25, 64), (60000, 1, 1.0, 500), (60000, 0, 0.0, 500),
          (250000, 600000, 0.3, 64), (150000, 300001, 0.5, 40)]          # the last two: more than 2048 partitions (3907 -> 4096, 3750 -> 3840: the four-partitions-per-thread instantiations)


@pytest.mark.parametrize("nb,npr,frac,per", SHAPES, ids=[f"{s[0]}x{s[1]}-m{s[2]}-r{s[3]}" for s in SHAPES])
def test_partitioned_pairs_match_oracle_exactly(ctx, nb, npr, frac, per):
    b = unique_keys(nb); p = probe_keys(b, npr, frac)
    with forced(ctx, per) as f:
        check(ctx, b, p)
        ran = f.kernels()
    if nb == 1:
        return                      # one key is a dense domain: the bitmap path keeps it
    assert "pj_build_check" in ran
    if npr:
        assert "pj_join" in ran and "k_probe_match_hash" not in ran, ran


def test_partitioned_equals_general_path(ctx):
    """A/B: option join_partitioned = 0 gives the same arrays."""
    import dfgpu
    b = unique_keys(40000); p = probe_keys(b, 150000, 0.3)
    out = []
    for on in (1, 0):
        with forced(ctx, 256, on) as f:
            t = dfgpu.JoinTable(ctx, [ctx.from_arrow(pa.array(b))])
            bi, pi = t.probe([ctx.from_arrow(pa.array(p))])
            out.append((bi.to_numpy(), pi.to_numpy()))
            assert ("pj_join" in f.kernels()) == bool(on)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("nb,npr,per", [(30000, 120000, 200), (240000, 300000, 64)], ids=["150-partitions", "3840-partitions"])
def test_partitioned_nulls_and_fused_selections(ctx, nb, npr, per):
    b = unique_keys(nb); p = probe_keys(b, npr, 0.5)
    with forced(ctx, per) as f:
        m = check(ctx, b, p, bmask=RNG.random(nb) < 0.7, pmask=RNG.random(npr) < 0.6, bnull=RNG.random(nb) < 0.1, pnull=RNG.random(npr) < 0.1)
        assert m > 1000 and "pj_join" in f.kernels()


@pytest.mark.parametrize("typ,lo,hi", [(pa.int32(), -(1 << 31), 1 << 31), (pa.uint32(), 0, 1 << 32), (pa.date32(), -(1 << 31), 1 << 31), (pa.int16(), -(1 << 15), 1 << 15),
                                       (pa.uint64(), 0, 1 << 63)], ids=["int32", "uint32", "date32", "int16", "uint64"])
def test_partitioned_integer_key_widths(ctx, typ, lo, hi):
    nb = 20000 if typ != pa.int16() else 200
    dt = {pa.int32(): np.int32, pa.uint32(): np.uint32, pa.date32(): np.int32, pa.int16(): np.int16, pa.uint64(): np.uint64}[typ]
    b = unique_keys(nb, dt, lo, hi); p = probe_keys(b, 90000, 0.4, dt, lo, hi)
    with forced(ctx, 50) as f:
        check(ctx, b, p, typ=typ)
        assert "pj_join" in f.kernels()


def test_dense_domain_keeps_the_bitmap_path(ctx):
    """keys 0 .. 4n shuffled: the membership bitmap prefilter stays in charge (range <= 256 x rows)"""
    nb = 30000
    b = RNG.permutation(4 * nb)[:nb].astype(np.int64); p = RNG.integers(0, 4 * nb, 100000).astype(np.int64)
    with forced(ctx, 100) as f:
        check(ctx, b, p)
        assert "pj_join" not in f.kernels()


def test_repeated_build_keys_take_the_partitioned_path(ctx):
    """one repeated key: the distinct keys of every partition become groups, the rows of a key a CSR (k_pj_groups); pairs in probe order, then build order"""
    b = unique_keys(30000); b[1234] = b[77]; p = probe_keys(b, 80000, 0.5)
    with forced(ctx, 100) as f:
        check(ctx, b, p)
        ran = f.kernels()
        assert "pj_build_check" in ran and "pj_build_groups" in ran and "pj_join" in ran and "k_probe_match_hash" not in ran, ran


@pytest.mark.parametrize("nb,npr,distinct,per", [(60000, 200000, 9000, 128), (40000, 100000, 39000, 64), (50000, 150000, 500, 12800), (3000, 50000, 1, 12800), (200000, 300000, 40000, 256), (300000, 400000, 60000, 90)],
                         ids=["fk7", "few-dups", "100-per-key", "one-key", "fk5-2048-partitions", "fk5-3584-partitions"])
def test_foreign_key_builds(ctx, nb, npr, distinct, per):
    """a build side whose keys repeat (a foreign key): every match emits its key's rows in build input order; masks and NULLs on both sides"""
    keys = unique_keys(distinct)
    b = keys[RNG.integers(0, distinct, nb)]
    p = probe_keys(keys, npr, 0.4)
    with forced(ctx, per) as f:
        m = check(ctx, b, p, bmask=RNG.random(nb) < 0.8, pmask=RNG.random(npr) < 0.7, bnull=RNG.random(nb) < 0.05, pnull=RNG.random(npr) < 0.05)
        ran = f.kernels()
    assert m > 0
    if distinct >= 500:                 # few rows per key: the partitioned CSR; a key with more rows than PJ_MAX_GROUP leaves the build to the general path
        assert "pj_join" in ran and "pj_build_groups" in ran, ran
    else:
        assert "pj_join" not in ran


def test_overfull_partition_falls_back(ctx):
    """20 000 distinct keys whose hash lands in partition 0 of 3 (7300 rows per partition are planned): beyond the 16 382 rows an LDS table indexes -> general path, same pairs"""
    ks = []
    x = 1
    while len(ks) < 20000:
        x += 1
        if (mix64(x * 7919) >> 32) * 3 >> 32 == 0:
            ks.append(x * 7919)
    b = np.array(ks, dtype=np.int64)[RNG.permutation(len(ks))]          # range / rows ~ 16 000: too sparse for the bitmap path
    p = probe_keys(b, 60000, 0.5)
    with forced(ctx, 12800) as f:
        check(ctx, b, p)
        ran = f.kernels()
        assert "pj_join" not in ran


def test_two_packed_key_columns_take_the_partitioned_path(ctx):
    """(a, b) with ranges 10^4 x 10^5 packs into one sparse Int64 key (join.hip k_pack_keys) -> partitioned probe; pairs equal the oracle's"""
    import dfgpu
    nb, npr = 20000, 90000
    flat = unique_keys(nb, np.int64, 0, 10**9)
    b0, b1 = (flat // 10**5).astype(np.int64), (flat % 10**5).astype(np.int64)
    pf = probe_keys(flat, npr, 0.4, np.int64, 0, 10**9)
    p0, p1 = (pf // 10**5).astype(np.int64), (pf % 10**5).astype(np.int64)
    with forced(ctx, 128) as f:
        t = dfgpu.JoinTable(ctx, [ctx.from_arrow(pa.array(b0)), ctx.from_arrow(pa.array(b1))])
        bi, pi = t.probe([ctx.from_arrow(pa.array(p0)), ctx.from_arrow(pa.array(p1))])
        ran = f.kernels()
    want = po.hash_join([[pa.array(b0), pa.array(b1)]], [[pa.array(p0), pa.array(p1)]], "Inner", False, batch_size=1 << 40)
    assert np.array_equal(bi.to_numpy().astype(np.int64), want.build_idx) and np.array_equal(pi.to_numpy().astype(np.int64), want.probe_idx)
    assert "pj_join" in ran


def test_packed_join_keys_with_a_date32_column(ctx):
    """(Int64, Date32) join keys packed into one Int64 (join.hip k_pack_keys): a Date32 column is 4 bytes per row.  Regression for round 3: key_at() had no
    Date32 case and read 8 bytes per row from it (out of bounds, garbage ranges -- the pack was then refused for the wrong reason)."""
    import dfgpu
    nb, npr = 40000, 120000
    bk = unique_keys(nb, np.int64, 0, 10**6); bd = (9000 + (bk % 1500)).astype(np.int32)
    hit = RNG.random(npr) < 0.5; pick = RNG.integers(0, nb, npr)
    pk = np.where(hit, bk[pick], RNG.integers(0, 10**6, npr)).astype(np.int64); pd = np.where(hit, bd[pick], RNG.integers(9000, 10500, npr)).astype(np.int32)
    arr = lambda k, d: [pa.array(k), pa.array(d).cast(pa.date32())]
    with forced(ctx, 256) as f:
        t = dfgpu.JoinTable(ctx, [ctx.from_arrow(a) for a in arr(bk, bd)])
        bi, pi = t.probe([ctx.from_arrow(a) for a in arr(pk, pd)])
        ran = f.kernels()
    want = po.hash_join([arr(bk, bd)], [arr(pk, pd)], "Inner", False, batch_size=1 << 40)
    assert np.array_equal(bi.to_numpy().astype(np.int64), want.build_idx) and np.array_equal(pi.to_numpy().astype(np.int64), want.probe_idx)
    assert "k_pack_keys" in ran, ran              # the ranges (10^6 x 1500) fit: the packed single-key paths serve the join


# ------------------------------------------------------------------ hashed mode: key columns the integer mode does not take travel as 64-bit keyset hashes, pairs are verified in the columns
def _hashed_case(kind, nb, npr, rng):
    """-> (build key columns, probe key columns) as pyarrow arrays; about a third of the probe rows match, keys repeat on the build side in some cases"""
    words = np.array([f"k{i:07d}" + "x" * (i % 5) for i in range(nb)], dtype=object)
    if kind == "utf8":
        b = [pa.array(words[rng.permutation(nb)], pa.utf8())]
        p = [pa.array(np.where(rng.random(npr) < 0.3, words[rng.integers(0, nb, npr)], np.array([f"q{i}" for i in range(npr)], dtype=object)), pa.utf8())]
    elif kind == "utf8_nullable_dups":                # every key twice on the build side, NULLs on both sides (never a match)
        base = words[: nb // 2]
        b = [pa.array(np.concatenate([base, base])[rng.permutation(2 * (nb // 2))], pa.utf8(), mask=rng.random(2 * (nb // 2)) < 0.05)]
        p = [pa.array(np.where(rng.random(npr) < 0.3, base[rng.integers(0, len(base), npr)], "none"), pa.utf8(), mask=rng.random(npr) < 0.05)]
    elif kind == "two_wide_ints":                     # ranges multiply far beyond 2^40: no packing
        k0 = rng.integers(-(1 << 60), 1 << 60, nb); k1 = rng.integers(-(1 << 60), 1 << 60, nb)
        pick = rng.integers(0, nb, npr); hit = rng.random(npr) < 0.3
        b = [pa.array(k0), pa.array(k1)]
        p = [pa.array(np.where(hit, k0[pick], rng.integers(0, 1 << 60, npr))), pa.array(np.where(rng.random(npr) < 0.9, k1[pick], 7))]      # some rows match in the first column only
    elif kind == "int_and_utf8":
        k0 = rng.integers(0, 50, nb); pick = rng.integers(0, nb, npr); hit = rng.random(npr) < 0.3
        b = [pa.array(k0), pa.array(words, pa.utf8())]
        p = [pa.array(np.where(hit, k0[pick], -1)), pa.array(words[pick], pa.utf8())]
    elif kind == "dictionary":
        d = pa.array([f"d{i}" for i in range(4000)], pa.utf8())
        b = [pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 4000, nb).astype(np.int32)), d)]
        p = [pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 4000, npr).astype(np.int32)), d)]      # ~nb / 4000 build rows per key
    else:
        raise AssertionError(kind)
    return b, p


def _check_hashed(ctx, b, p, nen=False, bmask=None, pmask=None):
    import dfgpu
    table = dfgpu.JoinTable(ctx, [ctx.from_arrow(c) for c in b], mask=ctx.from_arrow(pa.array(bmask)) if bmask is not None else None, null_equals_null=nen)
    bi, pi = table.probe([ctx.from_arrow(c) for c in p], mask=ctx.from_arrow(pa.array(pmask)) if pmask is not None else None)
    bsel = np.arange(len(b[0])) if bmask is None else np.flatnonzero(bmask)
    psel = np.arange(len(p[0])) if pmask is None else np.flatnonzero(pmask)
    plain = lambda c: c.cast(c.type.value_type) if pa.types.is_dictionary(c.type) else c
    want = po.hash_join([[plain(c).take(pa.array(bsel)) for c in b]], [[plain(c).take(pa.array(psel)) for c in p]], "Inner", nen, batch_size=1 << 40)
    assert np.array_equal(bi.to_numpy().astype(np.int64), bsel[want.build_idx])
    assert np.array_equal(pi.to_numpy().astype(np.int64), psel[want.probe_idx])
    return len(want.probe_idx)


@pytest.mark.parametrize("per", [64, 12800])
@pytest.mark.parametrize("kind", ["utf8", "utf8_nullable_dups", "two_wide_ints", "int_and_utf8", "dictionary"])
def test_hashed_keys_take_the_partitioned_path(ctx, kind, per):
    rng = np.random.default_rng(len(kind) * 1000 + per)
    b, p = _hashed_case(kind, 40_000, 150_000, rng)
    if kind == "dictionary" and per == 64:
        pytest.skip("groups of ~10 rows per key in partitions of 64 rows: the build declines (covered by the fall-back test)")
    with forced(ctx, per) as f:
        m = _check_hashed(ctx, b, p)
        ks = f.kernels()
    assert m > 0 and "pj_join" in ks and "pj_verify" in ks, ks


def test_hashed_keys_with_masks_and_null_equals_null(ctx):
    """null_equals_null: NULL keys hash (keyset_hash leaves the running hash unchanged for a NULL cell) and match each other; one Int64 key column with NULLs on both
    sides, which the integer mode refuses; fused selections on both sides."""
    rng = np.random.default_rng(77)
    nb, npr = 30_000, 100_000
    k = unique_keys(nb); bnull = rng.random(nb) < 0.001            # a few NULL build keys: they form one group of ~30 rows
    pk = probe_keys(k, npr, 0.3); pnull = rng.random(npr) < 0.01
    b = [pa.array(k, mask=bnull)]; p = [pa.array(pk, mask=pnull)]
    with forced(ctx, 12800) as f:
        m = _check_hashed(ctx, b, p, nen=True, bmask=rng.random(nb) < 0.9, pmask=rng.random(npr) < 0.8)
        ks = f.kernels()
    assert m > 0 and "pj_join" in ks and "pj_verify" in ks and "pj_build_groups" in ks, ks


@pytest.mark.parametrize("bits", [12, 16])
def test_hashed_keys_that_share_a_hash_are_told_apart(ctx, bits):
    """`join_partitioned_hash_mask` leaves `bits` bits of every key hash, so thousands of different keys share one: the tables, the groups and the expansion treat them as
    one key, the verification against the columns drops the wrong pairs -- the result is still the oracle's, in order."""
    rng = np.random.default_rng(bits)
    b, p = _hashed_case("utf8", 30_000, 100_000, rng)
    ctx.set_option("join_partitioned_hash_mask", (1 << bits) - 1)
    try:
        with forced(ctx, 12800) as f:
            m = _check_hashed(ctx, b, p)
            ks = f.kernels()
    finally:
        ctx.set_option("join_partitioned_hash_mask", 0)
    assert m > 0 and "pj_verify" in ks and "pj_build_groups" in ks, ks


def test_hashed_mode_can_be_switched_off_and_small_builds_do_not_take_it(ctx):
    rng = np.random.default_rng(5)
    b, p = _hashed_case("utf8", 20_000, 60_000, rng)
    with forced(ctx, 12800) as f:
        ctx.set_option("join_partitioned_hashed", 0)
        try:
            _check_hashed(ctx, b, p)
            assert "pj_join" not in f.kernels()
        finally:
            ctx.set_option("join_partitioned_hashed", 1)
    _check_hashed(ctx, b, p)                                        # default thresholds: 20 000 rows stay with the general table


# ------------------------------------------------------------------ rank index over UNSORTED unique keys (join.hip build_rank_index_unsorted)
@pytest.mark.parametrize("shape", ["permutation", "one_in_five_masked", "int32_negative_range", "a_key_repeats", "domain_too_wide"])
def test_rank_index_over_unsorted_unique_keys(ctx, shape):
    """Unique integer build keys over a dense domain in any order (a primary-key column after a hash repartition): the membership bitmap ranks the keys and one array maps
    rank -> build row; no hash table is built.  A repeated key (found while the bits are set) or a domain of more than 256 slots per key leaves the build to the hash paths.
    Pairs are the oracle's, in order, either way."""
    rng = np.random.default_rng(len(shape))
    nb, npr = 200_000, 600_000
    typ, bmask = None, None
    if shape == "permutation":
        b = rng.permutation(nb).astype(np.int64) + 1000
    elif shape == "one_in_five_masked":
        b = (rng.permutation(nb * 5)[:nb]).astype(np.int64); bmask = rng.random(nb) < 0.7
    elif shape == "int32_negative_range":
        b = (rng.permutation(nb * 2)[:nb] - nb).astype(np.int32); typ = pa.int32()
    elif shape == "a_key_repeats":
        b = rng.permutation(nb).astype(np.int64); b[777] = b[5]
    else:
        b = (rng.permutation(nb) * 1000).astype(np.int64)
    lo, hi = int(b.min()), int(b.max())
    p = rng.integers(lo - 50, hi + 50, npr).astype(b.dtype)
    ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read()
    try:
        m = check(ctx, b, p, bmask=bmask, pmask=rng.random(npr) < 0.9, typ=typ)
        ks = set(ctx.profile_read())
    finally:
        ctx.profile_enable(False)
    assert m > 0
    taken = shape in ("permutation", "one_in_five_masked", "int32_negative_range")
    assert ("k_probe_lookup_rank" in ks) == taken and ("k_join_build" in ks) == (not taken), ks
    if taken:                                   # switched off, the same build takes a hash table and gives the same pairs
        ctx.set_option("join_rank_index_unsorted", 0)
        try:
            ctx.profile_enable(True); ctx.profile_read()
            check(ctx, b, p, bmask=bmask, typ=typ)
            assert "k_join_build" in set(ctx.profile_read())
        finally:
            ctx.profile_enable(False); ctx.set_option("join_rank_index_unsorted", 1)


def test_builds_beyond_2048_partitions_take_up_to_4096(ctx):
    """A build of more than 2048 x join_partition_rows rows used to decline to the global table; it now takes up to 4096 partitions (k_pj_scatter / k_pj_restore / k_pj_pstart
    with four partitions per thread).  Same pairs as the global table (option join_partitioned_big = 0: declined), and a build that needs more than 4096 still declines."""
    import dfgpu
    b = unique_keys(260000); p = probe_keys(b, 500000, 0.4)
    out = []
    for big in (1, 0):
        with forced(ctx, 64) as f:
            ctx.set_option("join_partitioned_big", big)
            try:
                t = dfgpu.JoinTable(ctx, [ctx.from_arrow(pa.array(b))])
                bi, pi = t.probe([ctx.from_arrow(pa.array(p))])
                out.append((bi.to_numpy(), pi.to_numpy()))
                assert ("pj_join" in f.kernels()) == bool(big)
            finally:
                ctx.set_option("join_partitioned_big", 1)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    b2 = unique_keys(300000); p2 = probe_keys(b2, 100000, 0.5)
    with forced(ctx, 64) as f:                       # 300 000 / 64 = 4688 partitions: beyond 4096
        check(ctx, b2, p2)
        assert "pj_join" not in f.kernels()
