"""-m gpu: arrays, create_hashes, take / filter / concat / slice through the C ABI vs the CPU oracle (bit-exact)."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(20241024)


def rand_array(kind, n, null_frac=0.2, rng=RNG):
    mask = rng.random(n) < null_frac if null_frac else None
    if kind == "int8": a = pa.array(rng.integers(-128, 127, n).astype(np.int8), mask=mask)
    elif kind == "int16": a = pa.array(rng.integers(-2**15, 2**15 - 1, n).astype(np.int16), mask=mask)
    elif kind == "int32": a = pa.array(rng.integers(-2**31, 2**31 - 1, n).astype(np.int32), mask=mask)
    elif kind == "int64": a = pa.array(rng.integers(-2**62, 2**62, n).astype(np.int64), mask=mask)
    elif kind == "uint8": a = pa.array(rng.integers(0, 255, n).astype(np.uint8), mask=mask)
    elif kind == "uint16": a = pa.array(rng.integers(0, 2**16 - 1, n).astype(np.uint16), mask=mask)
    elif kind == "uint32": a = pa.array(rng.integers(0, 2**32 - 1, n).astype(np.uint32), mask=mask)
    elif kind == "uint64": a = pa.array(rng.integers(0, 2**63, n).astype(np.uint64) * 2, mask=mask)
    elif kind == "float32": a = pa.array(rng.normal(size=n).astype(np.float32), mask=mask)
    elif kind == "float64": a = pa.array(rng.normal(size=n) * 1e6, mask=mask)
    elif kind == "date32": a = pa.array(rng.integers(8000, 11000, n).astype(np.int32), mask=mask).cast(pa.date32())
    elif kind == "bool": a = pa.array(rng.random(n) < 0.5, mask=mask)
    elif kind == "decimal":
        import decimal
        vals = [decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**12, 10**12, n)]
        a = pa.array([None if (mask is not None and m) else v for v, m in zip(vals, mask if mask is not None else [False] * n)], type=pa.decimal128(15, 2))
    elif kind == "utf8":
        words = ["", "a", "BUILDING", "AUTOMOBILE", "MACHINERY", "HOUSEHOLD", "FURNITURE", "x" * 37, "日本語", "ASIA"]
        a = pa.array([None if (mask is not None and m) else words[i] + str(i % 7) for i, m in zip(rng.integers(0, len(words), n), mask if mask is not None else [False] * n)], type=pa.utf8())
    elif kind == "dict":
        a = rand_array("utf8", n, null_frac, rng).dictionary_encode()
    else: raise ValueError(kind)
    return a


KINDS = ["int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float32", "float64", "date32", "bool", "decimal", "utf8", "dict"]


def plain(a):
    return a.dictionary_decode() if pa.types.is_dictionary(a.type) else a


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000, 4097])
def test_import_export_roundtrip(ctx, kind, n):
    a = rand_array(kind, n)
    back = ctx.from_arrow(a).to_arrow()
    assert back.type == a.type and back.equals(a)


@pytest.mark.parametrize("kind", KINDS)
def test_create_hashes_matches_oracle(ctx, kind):
    a = rand_array(kind, 5000)
    got = ctx.hash_columns([ctx.from_arrow(a)]).to_numpy()
    assert np.array_equal(got, po.create_hashes([a]))


def test_create_hashes_multi_column_and_relations(ctx):
    cols = [rand_array(k, 3000) for k in ["int64", "utf8", "decimal", "dict", "float64", "bool", "date32"]]
    got = ctx.hash_columns([ctx.from_arrow(c) for c in cols]).to_numpy()
    assert np.array_equal(got, po.create_hashes(cols))
    # relations the reference asserts (hash_utils.rs:523-586): NULL in the only column leaves hash 0; dictionary == plain values
    a = rand_array("utf8", 2000, 0.3)
    h = ctx.hash_columns([ctx.from_arrow(a)]).to_numpy()
    assert (h[np.asarray(a.is_null())] == 0).all()
    hd = ctx.hash_columns([ctx.from_arrow(a.dictionary_encode())]).to_numpy()
    assert np.array_equal(h, hd)


def test_force_hash_collisions_option(ctx):
    a = rand_array("int64", 100)
    ctx.set_option("force_hash_collisions", 1)
    try:
        assert (ctx.hash_columns([ctx.from_arrow(a)]).to_numpy() == 0).all()
    finally:
        ctx.set_option("force_hash_collisions", 0)


@pytest.mark.parametrize("kind", KINDS)
def test_take_matches_arrow(ctx, kind):
    a = rand_array(kind, 3000)
    idx = RNG.integers(0, 3000, 5000)
    imask = RNG.random(5000) < 0.1
    for idx_arr in [pa.array(idx.astype(np.uint32)), pa.array(idx.astype(np.uint64), mask=imask)]:
        got = ctx.take(ctx.from_arrow(a), ctx.from_arrow(idx_arr)).to_arrow()
        assert plain(got).equals(plain(a).take(idx_arr))
        oidx = np.where(np.asarray(idx_arr.is_null()), -1, idx)
        assert plain(got).equals(po.take(a, oidx))


def test_take_out_of_bounds_is_an_error(ctx):
    import dfgpu
    with pytest.raises(dfgpu.DfgpuError) as e:
        ctx.take(ctx.from_arrow(pa.array([1, 2, 3])), ctx.from_arrow(pa.array([0, 7], type=pa.uint32())))
    assert e.value.kind == "Execution"


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("n", [0, 1, 64, 2047, 2048, 2049, 10000])
def test_filter_matches_arrow(ctx, kind, n):
    a = rand_array(kind, n)
    m = rand_array("bool", n, 0.15)
    got = ctx.filter(ctx.from_arrow(a), ctx.from_arrow(m)).to_arrow()
    assert plain(got).equals(plain(a).filter(m, null_selection_behavior="drop"))
    assert plain(got).equals(po.filter_(a, m))
    sel = ctx.mask_to_indices(ctx.from_arrow(m)).to_numpy()
    assert np.array_equal(sel, np.flatnonzero(np.asarray(m.fill_null(False))))


@pytest.mark.parametrize("kind", [k for k in KINDS if k != "dict"])
def test_concat_and_slice(ctx, kind):
    parts = [rand_array(kind, n, nf) for n, nf in [(100, 0.2), (0, 0.0), (37, 0.0), (1000, 0.5), (1, 0.0)]]
    cat = ctx.concat([ctx.from_arrow(p) for p in parts])
    want = pa.concat_arrays(parts)
    assert cat.to_arrow().equals(want)
    for off, ln in [(0, 64), (64, 500), (3, 70), (1000, 138), (1137, 1)]:
        assert cat.slice(off, ln).to_arrow().equals(want.slice(off, ln))
    assert cat.null_count == want.null_count


def test_new_null(ctx):
    import dfgpu
    a = ctx.new_null(dfgpu.capi.DECIMAL128, 70, 15, 2).to_arrow()
    assert a.type == pa.decimal128(15, 2) and a.null_count == 70
    u = ctx.new_null(dfgpu.capi.UTF8, 5).to_arrow()
    assert u.null_count == 5
