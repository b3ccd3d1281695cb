"""-m gpu: Parquet pages -> Arrow columns in HBM (csrc/parquet.hip) against pyarrow reading the same file.

The decoding algorithm lives in the `parquet` crate (arrow-rs 50), which is not part of the reference tree; the checker here is the Arrow C++
reader behind pyarrow -- an independent implementation of the same published format -- and the bar is bit-exact columns (values, validity,
offsets; Float columns compared by bit pattern).  Files: the committed fixtures of tests/golden/parquet (written by pyarrow under seven writer
configurations, plus the reference's own clickbench_hits_10.parquet written by DuckDB), and larger files written at test time."""
import glob
import os

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pyarrow.parquet as pq
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parquet")
FILES = sorted(f for f in glob.glob(os.path.join(HERE, "*.parquet")) if "unsupported" not in f)


def same_column(got: pa.Array, want: pa.ChunkedArray, name=""):
    want = want.combine_chunks() if isinstance(want, pa.ChunkedArray) else want
    if pa.types.is_dictionary(got.type):
        got = got.dictionary_decode()
    if pa.types.is_large_string(want.type) or pa.types.is_dictionary(want.type):
        want = want.cast(pa.string())
    assert got.type == want.type, (name, got.type, want.type)
    assert len(got) == len(want), name
    assert got.null_count == want.null_count, name
    if pa.types.is_floating(got.type):
        it = np.uint32 if got.type == pa.float32() else np.uint64
        a = got.fill_null(0).to_numpy(zero_copy_only=False).view(it); b = want.fill_null(0).to_numpy(zero_copy_only=False).view(it)
        assert np.array_equal(a, b), name
        assert got.is_valid().equals(want.is_valid()), name
    else:
        assert got.equals(want), name


def supported(ref: pq.ParquetFile, f):
    return [i for i in range(f.num_columns) if f.column_type(i)[0] != 0]


@pytest.mark.parametrize("staged", [False, True], ids=["host-image", "device-image"])
@pytest.mark.parametrize("as_dict", [True, False], ids=["utf8-dictionary", "utf8-plain"])
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-8] for f in FILES])
def test_fixture_columns_equal_pyarrow(ctx, path, as_dict, staged):
    from dfgpu.parquet import ParquetFile
    from dfgpu import capi
    want = pq.read_table(path)
    f = ParquetFile(ctx, path=path, stage_on_device=staged, utf8_dictionary=as_dict)
    cols = supported(None, f)
    assert len(cols) == want.num_columns
    got = f.read(columns=cols)
    for i, a in zip(cols, got):
        name = f.column_names()[i]
        t = f.column_type(i)[0]
        assert a.type == t, name                      # the type the schema promised, in every batch
        same_column(a.to_arrow(), want[name], name)
    # row group by row group == the matching slice
    off = 0
    for g in range(f.num_row_groups):
        part = f.read(g, 1, cols[:4])
        for i, a in zip(cols[:4], part):
            same_column(a.to_arrow(), want[f.column_names()[i]].slice(off, f.row_group_rows(g)))
        off += f.row_group_rows(g)
    ctx.synchronize()


def test_unsupported_columns_say_not_implemented_and_the_rest_reads(ctx):
    import dfgpu
    from dfgpu.parquet import ParquetFile
    path = os.path.join(HERE, "unsupported_columns.parquet")
    f = ParquetFile(ctx, path=path)
    names = f.column_names()
    for bad in ("ts", "bin"):
        assert f.column_type(names.index(bad))[0] == 0
        with pytest.raises(dfgpu.DfgpuError) as e:
            f.read(columns=[bad])
        assert e.value.kind == "NotImplemented"
    same_column(f.read(columns=["ok"])[0].to_arrow(), pq.read_table(path)["ok"])


def big_table(n, seed=5):
    rng = np.random.default_rng(seed)
    return pa.table({
        "l_orderkey": pa.array(np.sort(rng.integers(0, n // 4, n)).astype(np.int64)),
        "l_quantity": pa.array((rng.integers(1, 51, n) * 100).astype(np.int64)).cast(pa.decimal128(15, 2), safe=False) if False else pa.array(rng.integers(1, 51, n).astype(np.int32)),
        "l_extendedprice": pa.array(rng.random(n) * 1e5),
        "l_shipdate": pa.array(rng.integers(8036, 10592, n).astype(np.int32), type=pa.date32()),
        "l_returnflag": pa.array(np.array(["A", "N", "R"])[rng.integers(0, 3, n)], mask=rng.random(n) < 0.01),
        "l_shipmode": pa.array(np.array(["AIR", "FOB", "MAIL", "RAIL", "REG AIR", "SHIP", "TRUCK"])[rng.integers(0, 7, n)]),
        "l_comment": pa.array([f"c{v:x}" * (1 + v % 5) for v in rng.integers(0, 1 << 40, n)]),
        "l_nullable": pa.array(rng.integers(-10**12, 10**12, n), mask=rng.random(n) < 0.3),
    })


@pytest.mark.parametrize("kw", [dict(compression="snappy", use_dictionary=True), dict(compression="none", use_dictionary=False), dict(compression="snappy", use_dictionary=True, data_page_version="2.0", data_page_size=1 << 16)],
                         ids=["snappy-dict", "plain", "snappy-dict-v2-small-pages"])
def test_larger_file_many_pages_and_row_groups(ctx, tmp_path, kw):
    from dfgpu.parquet import ParquetFile
    t = big_table(600000)
    path = str(tmp_path / "big.parquet")
    pq.write_table(t, path, row_group_size=150000, **kw)
    want = pq.read_table(path)
    f = ParquetFile(ctx, path=path, stage_on_device=True)
    assert f.num_row_groups == 4
    got = f.read()
    for name, a in zip(f.column_names(), got):
        same_column(a.to_arrow(), want[name], name)
    got = f.read(1, 2, ["l_comment", "l_nullable", "l_returnflag"])           # the middle: dictionary bases of two row groups
    for name, a in zip(["l_comment", "l_nullable", "l_returnflag"], got):
        same_column(a.to_arrow(), want[name].slice(150000, 300000), name)


def test_parquet_exec_feeds_filter_and_aggregate(ctx, tmp_path):
    """ParquetExec -> FilterExec(l_shipdate <= d AND l_shipmode = 'MAIL') -> AggregateExec(GROUP BY l_returnflag: COUNT, SUM): the dictionary columns go
    into the predicate and the group-by as they come off the pages; == pyarrow's group_by on the decoded table.  Row-group pruning on the sorted key."""
    from dfgpu import capi, physical_plan as ops
    from dfgpu.parquet import ParquetFile
    t = big_table(400000, seed=9)
    path = str(tmp_path / "li.parquet")
    pq.write_table(t, path, row_group_size=50000, compression="snappy")
    f = ParquetFile(ctx, path=path, stage_on_device=True)
    C, F = ops.Column, ops.Field
    scan = ops.ParquetExec(f, ["l_orderkey", "l_quantity", "l_shipdate", "l_returnflag", "l_shipmode"], partitions=2, row_groups_per_batch=2, prune=[("l_orderkey", 0, 30000)])
    assert scan.schema().names() == ["l_orderkey", "l_quantity", "l_shipdate", "l_returnflag", "l_shipmode"]
    lit = ops.Literal
    pred = ops.BinaryExpr(ops.BinaryExpr(ops.BinaryExpr(C("l_shipdate", 2), "<=", lit(9500, pa.date32())), "AND", ops.BinaryExpr(C("l_shipmode", 4), "=", lit("MAIL", pa.string()))),
                          "AND", ops.BinaryExpr(C("l_orderkey", 0), "<=", lit(30000, pa.int64())))
    agg = ops.AggregateExec("Single", [(C("l_returnflag", 3), "l_returnflag")],
                            [ops.AggregateFunctionExpr("COUNT", None, "n"), ops.AggregateFunctionExpr("SUM", ops.CastExpr(C("l_quantity", 1), capi.INT64), "q", input_field=F("l_quantity", capi.INT64))],
                            ops.CoalescePartitionsExec(ops.FilterExec(pred, scan)))
    tc = ops.TaskContext(ctx, 8192)
    out = pa.concat_tables([b.to_arrow() for b in agg.execute(0, tc)])
    sel = t.filter(pc.and_(pc.and_(pc.less_equal(t["l_shipdate"], pa.scalar(9500, pa.int32()).cast(pa.date32())), pc.equal(t["l_shipmode"], "MAIL")), pc.less_equal(t["l_orderkey"], 30000)))
    want = sel.group_by("l_returnflag").aggregate([([], "count_all"), ("l_quantity", "sum")])
    g = {(r["l_returnflag"]): (r["n"], r["q"]) for r in out.to_pylist()}
    w = {(r["l_returnflag"]): (r["count_all"], r["l_quantity_sum"]) for r in want.to_pylist()}
    assert g == w and len(g) == 4                            # A, N, R and NULL
    assert scan.row_groups_pruned(tc) >= 4                    # keys are sorted: the later row groups lie above 30000


def test_malformed_pages_raise_instead_of_faulting(ctx, tmp_path):
    """Bytes inside the data pages overwritten: the kernels bound every read by the page and flag the page (Execution error), or the damage lands in
    value bytes and decodes to different values -- never an out-of-bounds access."""
    import dfgpu
    from dfgpu.parquet import ParquetFile
    good = bytearray(open(os.path.join(HERE, "dict_snappy_v1.parquet"), "rb").read())
    rng = np.random.default_rng(1)
    md = pq.ParquetFile(os.path.join(HERE, "dict_snappy_v1.parquet")).metadata
    raised = 0
    for trial in range(6):
        bad = bytearray(good)
        cc = md.row_group(0).column(int(rng.integers(0, md.num_columns)))
        start = cc.dictionary_page_offset or cc.data_page_offset
        for _ in range(40):
            bad[start + 30 + int(rng.integers(0, max(1, cc.total_compressed_size - 40)))] = int(rng.integers(0, 256))
        try:
            f = ParquetFile(ctx, data=bytes(bad))
            f.read()
            ctx.synchronize()
        except dfgpu.DfgpuError as e:
            assert e.kind in ("Execution", "NotImplemented"), str(e)
            raised += 1
    assert raised >= 1


def test_snappy_element_patterns_across_blocks(ctx, tmp_path):
    """PLAIN pages of several hundred KB under Snappy, chosen for what the decompressor sees: runs (copies of offset 8 / 1 that overlap their own output), incompressible
    bytes (literals of up to 64 KB), a pattern that repeats every 50 000 bytes (references further back than the 32 KB ring: read from HBM), short repeats inside
    strings, and a sorted column (two elements per value).  Multi-block pages: every 64 KB output block is decoded by its own wave."""
    from dfgpu.parquet import ParquetFile
    n = 400000
    rng = np.random.default_rng(17)
    period = rng.integers(0, 1 << 62, 6250).astype(np.int64)             # 50 000 bytes
    words = np.array(["alpha", "beta", "gamma", "delta-delta-delta", "", "x" * 70, "épsilon"])
    t = pa.table({
        "constant": pa.array(np.full(n, 123456789012345, dtype=np.int64)),
        "random": pa.array(rng.integers(-(1 << 62), 1 << 62, n).astype(np.int64)),
        "sorted": pa.array(np.cumsum(rng.integers(0, 9, n)).astype(np.int64)),
        "sawtooth": pa.array((np.arange(n) % 1000).astype(np.int32)),
        "far_repeat": pa.array(np.tile(period, n // len(period) + 1)[:n]),
        "zeros_then_noise": pa.array(np.where(np.arange(n) % 20000 < 15000, 0, rng.integers(0, 1 << 40, n)).astype(np.int64)),
        "text": pa.array(words[rng.integers(0, len(words), n)]),
        "runs_text": pa.array(["ab" * int(k) for k in rng.integers(0, 40, n)]),
        "nullable": pa.array(rng.integers(0, 3, n).astype(np.int64), mask=rng.random(n) < 0.5),
    })
    path = str(tmp_path / "patterns.parquet")
    pq.write_table(t, path, compression="snappy", use_dictionary=False, row_group_size=n, data_page_size=1 << 20)
    want = pq.read_table(path)
    for staged in (True, False):
        f = ParquetFile(ctx, path=path, stage_on_device=staged, utf8_dictionary=False)
        for name, a in zip(f.column_names(), f.read()):
            same_column(a.to_arrow(), want[name], name)


def _pattern_table(n, seed=17):
    rng = np.random.default_rng(seed)
    period = rng.integers(0, 1 << 62, 6250).astype(np.int64)             # 50 000 bytes
    words = np.array(["alpha", "beta", "gamma", "delta-delta-delta", "", "x" * 70, "épsilon"])
    return pa.table({
        "constant": pa.array(np.full(n, 123456789012345, dtype=np.int64)),
        "random": pa.array(rng.integers(-(1 << 62), 1 << 62, n).astype(np.int64)),
        "sorted": pa.array(np.cumsum(rng.integers(0, 9, n)).astype(np.int64)),
        "sawtooth": pa.array((np.arange(n) % 1000).astype(np.int32)),
        "far_repeat": pa.array(np.tile(period, n // len(period) + 1)[:n]),
        "zeros_then_noise": pa.array(np.where(np.arange(n) % 20000 < 15000, 0, rng.integers(0, 1 << 40, n)).astype(np.int64)),
        "text": pa.array(words[rng.integers(0, len(words), n)]),
        "runs_text": pa.array(["ab" * int(k) for k in rng.integers(0, 40, n)]),
        "nullable": pa.array(rng.integers(0, 3, n).astype(np.int64), mask=rng.random(n) < 0.5),
        "floats": pa.array(np.round(rng.normal(0, 100, n), 2)),
    })


@pytest.mark.parametrize("kw", [dict(compression_level=1, use_dictionary=False), dict(compression_level=3, use_dictionary=True), dict(compression_level=9, use_dictionary=False, data_page_version="2.0"),
                                dict(compression_level=19, use_dictionary=["text", "sawtooth"], data_page_size=1 << 16)], ids=["level1-plain", "level3-dict", "level9-v2", "level19-small-pages"])
def test_zstd_pages_equal_pyarrow(ctx, tmp_path, kw):
    """ZSTD (the codec DataFusion's Parquet writer defaults to): pages of up to 1 MB through every block and literals type the compressor picks for these columns -- RLE blocks
    (constant), raw blocks and raw literals (random), Huffman literals with one and four streams, FSE-coded weights, predefined / RLE / coded / repeated sequence tables, repeat
    offsets, matches that overlap their own output, and references further back than the 32 KB LDS ring -- bit-exact against pyarrow (libzstd) on the same file."""
    from dfgpu.parquet import ParquetFile
    n = 300000
    t = _pattern_table(n)
    path = str(tmp_path / "z.parquet")
    pq.write_table(t, path, compression="zstd", row_group_size=200000, **{"data_page_size": 1 << 20, **kw})
    want = pq.read_table(path)
    for staged in (True, False):
        f = ParquetFile(ctx, path=path, stage_on_device=staged, utf8_dictionary=False)
        for name, a in zip(f.column_names(), f.read()):
            same_column(a.to_arrow(), want[name], name)


def test_zstd_lineitem_shape_and_corrupt_frames(ctx, tmp_path):
    from dfgpu.parquet import ParquetFile
    import dfgpu
    t = big_table(400000)
    path = str(tmp_path / "li.parquet")
    pq.write_table(t, path, compression="zstd", row_group_size=100000)
    want = pq.read_table(path)
    f = ParquetFile(ctx, path=path, stage_on_device=True)
    for name, a in zip(f.column_names(), f.read()):
        same_column(a.to_arrow(), want[name], name)
    f.close()
    # flip bytes inside the first column chunk's compressed pages: an error (or, when the damage happens to decode, different values), never a fault
    raw = bytearray(open(path, "rb").read())
    md = pq.ParquetFile(path).metadata.row_group(0).column(0)
    start = md.dictionary_page_offset or md.data_page_offset
    for k in range(60, min(4000, md.total_compressed_size), 97):
        raw[start + k] ^= 0x5A
    bad = str(tmp_path / "bad.parquet"); open(bad, "wb").write(bytes(raw))
    g = ParquetFile(ctx, path=bad, stage_on_device=True)
    try:
        g.read(columns=[0])
    except dfgpu.DfgpuError:
        pass


@pytest.mark.parametrize("kw", [dict(use_dictionary=False), dict(use_dictionary=True, data_page_version="2.0")], ids=["plain", "dict-v2"])
def test_lz4_raw_pages_equal_pyarrow(ctx, tmp_path, kw):
    """LZ4_RAW (what pyarrow writes for compression="lz4"): tokens with extended literal and match lengths (incompressible 1 MB pages, constant columns), overlapping matches,
    references beyond the LDS ring -- bit-exact against pyarrow on the same file."""
    from dfgpu.parquet import ParquetFile
    t = _pattern_table(300000)
    path = str(tmp_path / "l.parquet")
    pq.write_table(t, path, compression="lz4", row_group_size=200000, data_page_size=1 << 20, **kw)
    want = pq.read_table(path)
    for staged in (True, False):
        f = ParquetFile(ctx, path=path, stage_on_device=staged, utf8_dictionary=False)
        for name, a in zip(f.column_names(), f.read()):
            same_column(a.to_arrow(), want[name], name)


def _plain_strings_case(kind, n, rng):
    if kind == "short":                      # the TPC-H shape: the byte before a length prefix reads as a small length too
        words = ["MAIL", "RAIL", "SHIP", "TRUCK", "AIR", "REG AIR", "FOB", "DELIVER IN PERSON", "COLLECT COD", "NONE", "TAKE BACK RETURN", ""]
        return [words[i] for i in rng.integers(0, len(words), n)]
    if kind == "looks_like_prefixes":        # values made of bytes that read as plausible length prefixes: every guess of the parallel walk can be wrong, the result may not be
        alphabet = ["\x01\x00\x00\x00", "\x04\x00\x00\x00\x00", "\x00\x00\x00\x00", "\x02\x00\x00\x00ab", "\x00", "\x03\x00\x00"]
        return ["".join(alphabet[j] for j in rng.integers(0, len(alphabet), int(k))) for k in rng.integers(0, 9, n)]
    if kind == "long":                       # values longer than a segment, longer than a window (64 KB), and tiny ones between them
        lens = rng.choice([0, 1, 3, 200, 300, 5000, 70_000, 140_000], n, p=[.2, .2, .2, .15, .15, .07, .02, .01])
        return [chr(97 + int(i) % 26) * int(L) for i, L in enumerate(lens)]
    raise AssertionError(kind)


@pytest.mark.parametrize("version", ["1.0", "2.0"], ids=["v1-pages", "v2-pages"])
@pytest.mark.parametrize("nulls", [False, True], ids=["required", "nullable"])
@pytest.mark.parametrize("kind", ["short", "looks_like_prefixes", "long"])
def test_plain_byte_arrays_walked_in_parallel(ctx, tmp_path, kind, nulls, version):
    """PLAIN BYTE_ARRAY pages (k_pq_str_walk): each lane walks a segment of the page from a guessed value boundary, the guesses are checked against the true path.  Short
    strings (guesses right), strings whose bytes look like length prefixes (guesses wrong), strings longer than a segment / the LDS window (segments jumped over), with and
    without NULLs, both page versions, pages of 64 KB..1 MB, host and device image."""
    from dfgpu.parquet import ParquetFile
    rng = np.random.default_rng(hash((kind, nulls, version)) % 2**32)
    n = 6_000 if kind == "long" else 250_000
    vals = _plain_strings_case(kind, n, rng)
    mask = rng.random(n) < 0.3 if nulls else None
    t = pa.table({"s": pa.array(vals, type=pa.string(), mask=mask), "i": pa.array(np.arange(n))})
    for page in (1 << 16, 1 << 20):
        path = str(tmp_path / f"{kind}_{page}.parquet")
        pq.write_table(t, path, compression="none", use_dictionary=False, data_page_version=version, data_page_size=page, row_group_size=100_000)
        for staged in (False, True):
            f = ParquetFile(ctx, path=path, stage_on_device=staged, utf8_dictionary=False)
            got = f.read()
            same_column(got[0].to_arrow(), t["s"], f"{kind} page={page} staged={staged}")
            same_column(got[1].to_arrow(), t["i"])
            f.close()
    ctx.synchronize()
