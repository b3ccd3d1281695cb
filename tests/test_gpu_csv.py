"""-m gpu: delimited text -> Arrow columns in HBM (csrc/csv.hip) against pyarrow.csv reading the same bytes under the same schema.

The CPU algorithm is the `arrow-csv` reader (arrow-rs 50) that CsvExec drives (core/src/datasource/physical_plan/csv.rs:CsvOpener) -- a dependency outside the
reference tree; the checker is the Arrow C++ CSV reader behind pyarrow, an independent implementation of the same format, with the conventions of
arrow-csv pinned in ConvertOptions (an empty field of a non-string column is NULL, an empty string field is the empty string).  Files: the data files the
reference's own tests read (core/tests/tpch-csv/*.csv and core/tests/data/*.csv, committed under tests/golden/csv as data), plus generated files with
quoted fields, doubled quotes, embedded delimiters and line feeds, CRLF, NULLs, no trailing newline.  Bar: bit-exact (Float64 by bit pattern)."""
import io
import os

import numpy as np
import pyarrow as pa
import pyarrow.csv as pcsv
import pytest

from test_gpu_parquet import same_column

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "csv")

I32, I64, F64, D32, DEC, S, B = pa.int32(), pa.int64(), pa.float64(), pa.date32(), pa.decimal128(15, 2), pa.string(), pa.bool_()
TPCH = {
    "customer": [("c_custkey", I64), ("c_name", S), ("c_address", S), ("c_nationkey", I64), ("c_phone", S), ("c_acctbal", DEC), ("c_mktsegment", S), ("c_comment", S)],
    "lineitem": [("l_orderkey", I64), ("l_partkey", I64), ("l_suppkey", I64), ("l_linenumber", I32), ("l_quantity", DEC), ("l_extendedprice", DEC), ("l_discount", DEC), ("l_tax", DEC),
                 ("l_returnflag", S), ("l_linestatus", S), ("l_shipdate", D32), ("l_commitdate", D32), ("l_receiptdate", D32), ("l_shipinstruct", S), ("l_shipmode", S), ("l_comment", S)],
    "nation": [("n_nationkey", I64), ("n_name", S), ("n_regionkey", I64), ("n_comment", S)],
    "orders": [("o_orderkey", I64), ("o_custkey", I64), ("o_orderstatus", S), ("o_totalprice", DEC), ("o_orderdate", D32), ("o_orderpriority", S), ("o_clerk", S), ("o_shippriority", I32), ("o_comment", S)],
    "part": [("p_partkey", I64), ("p_name", S), ("p_mfgr", S), ("p_brand", S), ("p_type", S), ("p_size", I32), ("p_container", S), ("p_retailprice", DEC), ("p_comment", S)],
    "partsupp": [("ps_partkey", I64), ("ps_suppkey", I64), ("ps_availqty", I32), ("ps_supplycost", DEC), ("ps_comment", S)],
    "region": [("r_regionkey", I64), ("r_name", S), ("r_comment", S)],
    "supplier": [("s_suppkey", I64), ("s_name", S), ("s_address", S), ("s_nationkey", I64), ("s_phone", S), ("s_acctbal", DEC), ("s_comment", S)],
}
DATA = {
    "data_aggregate_simple": ([("c1", F64), ("c2", F64), ("c3", B)], ","),
    "data_aggregate_simple_pipe": ([("c1", F64), ("c2", F64), ("c3", B)], "|"),
    "data_cars": ([("car", S), ("speed", F64), ("time", S)], ","),
    "data_decimal_data": ([("c1", pa.decimal128(10, 6)), ("c2", pa.decimal128(20, 12)), ("c3", I64), ("c4", B), ("c5", pa.decimal128(12, 7))], ","),
    "data_null_cases": ([("c1", I64), ("c2", F64), ("c3", I64)], ","),
    "data_wide_rows": (None, ","),
    "data_one_col": ([("c1", I32)], ","),
    "data_empty": ([("c1", I64), ("c2", S), ("c3", F64)], ","),
}


def dfgpu_schema(fields):
    from dfgpu import capi
    out = []
    for name, t in fields:
        code = {pa.int8(): 2, pa.int16(): 3, pa.int32(): 4, pa.int64(): 5, pa.uint8(): 6, pa.uint16(): 7, pa.uint32(): 8, pa.uint64(): 9, pa.float64(): 11, pa.date32(): 12, pa.string(): 14, pa.bool_(): 1}.get(t)
        if pa.types.is_decimal(t):
            out.append((name, 13, t.precision, t.scale))
        else:
            out.append((name, code, 0, 0))
    return out


def reference_read(data: bytes, fields, delimiter=",", has_header=True, include=None, escape=None):
    ro = pcsv.ReadOptions(column_names=None if has_header else [n for n, _ in fields])
    po = pcsv.ParseOptions(delimiter=delimiter, newlines_in_values=True, escape_char=escape or False)
    co = pcsv.ConvertOptions(column_types=dict(fields), null_values=[""], strings_can_be_null=False, quoted_strings_can_be_null=False, true_values=["true", "TRUE", "True"], false_values=["false", "FALSE", "False"],
                             include_columns=include)
    return pcsv.read_csv(io.BytesIO(data), read_options=ro, parse_options=po, convert_options=co)


def check(ctx, data: bytes, fields, delimiter=",", has_header=True, projection=None, on_device=False, escape=None):
    from dfgpu.csv import read_csv
    import torch
    names = [n for n, _ in fields]
    want = reference_read(data, fields, delimiter, has_header, include=None if projection is None else sorted(projection, key=names.index), escape=escape)
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda() if on_device else data
    got = read_csv(ctx, src, dfgpu_schema(fields), projection=projection, delimiter=delimiter, has_header=has_header, on_device=on_device, escape=escape)
    assert len(got) == want.num_columns
    for a, name in zip(got, want.column_names):
        same_column(a.to_arrow(), want[name], name)
    if len(data) <= 65536:                  # small inputs: also against the oracle's restatement of the record rules (plain Python; pinned on pyarrow in test_oracle_csv.py)
        from oracle import pyoracle as po
        recs = po.csv_records(data, delimiter, has_header=has_header, escape=escape)
        for a, name in zip(got, want.column_names):
            same_column(a.to_arrow(), po.csv_column(recs, names.index(name), dict(fields)[name]), name + " (oracle)")
    return want.num_rows


@pytest.mark.parametrize("table", sorted(TPCH))
def test_reference_tpch_csv_files(ctx, table):
    data = open(os.path.join(HERE, table + ".csv"), "rb").read()
    assert check(ctx, data, TPCH[table]) > 0
    # the projection lineitem / orders queries push into CsvExec: a few columns out of the record
    fields = TPCH[table]
    check(ctx, data, fields, projection=[fields[-1][0], fields[0][0]])
    check(ctx, data, fields, on_device=True)


@pytest.mark.parametrize("name", sorted(DATA))
def test_reference_data_csv_files(ctx, name):
    fields, delim = DATA[name]
    data = open(os.path.join(HERE, name + ".csv"), "rb").read()
    if fields is None:
        fields = [("column_%d" % (i + 1), I64) for i in range(len(data.split(b"\n")[0].split(b",")))]
    n = check(ctx, data, fields, delimiter=delim, has_header=name not in ("data_one_col", "data_wide_rows"))      # no header line in those two (file_format/csv.rs:1123)
    if name == "data_empty":
        assert n == 0


def test_quoting_crlf_nulls_and_missing_final_newline(ctx):
    fields = [("k", I64), ("s", S), ("f", F64), ("d", D32), ("m", pa.decimal128(12, 3)), ("b", B), ("u", pa.uint8())]
    rows = [
        b'1,plain,1.5,2020-02-29,12.345,true,0',
        b'-2,"with, comma",-0.25,1969-12-31,-0.5,FALSE,255',
        b',"say ""hi"" twice",,,,,',
        b'4,"line\nfeed inside",1e3,1600-01-01,+7,True,7',
        b'5,"",2.5E-3,9999-12-31,000.100,false,12',
        b'9223372036854775807,,123456789012345,0001-01-01,999999999.999,true,1',
        b'-9223372036854775808,"""",0.000,2000-03-01,.5,false,2',
        b'8,last,1,2001-01-01,1,true,"3"',
    ]
    for eol in (b"\n", b"\r\n"):
        for tail in (eol, b""):
            data = b"k,s,f,d,m,b,u" + eol + eol.join(rows) + tail
            assert check(ctx, data, fields) == len(rows)
            check(ctx, data, fields, projection=["s", "b"])
            check(ctx, eol.join(rows) + tail, fields, has_header=False)
            check(ctx, eol + data + eol + eol, fields)                         # blank lines are not records


def test_other_delimiters(ctx):
    fields = [("a", I32), ("b", S), ("c", F64)]
    for delim in ("|", "\t", ";"):
        d = delim.encode()
        data = d.join([b"a", b"b", b"c"]) + b"\n" + b"\n".join(d.join([str(i).encode(), b'"x' + d + b'y"' if i % 3 == 0 else b"v%d" % i, b"%d.%02d" % (i, i % 100)]) for i in range(1000)) + b"\n"
        assert check(ctx, data, fields, delimiter=delim) == 1000


def test_large_generated_file(ctx):
    rng = np.random.default_rng(5)
    n = 300_000
    k = rng.integers(-2**40, 2**40, n); q = rng.integers(0, 10**9, n); dt = rng.integers(-20000, 40000, n).astype("datetime64[D]")
    words = np.array(["", "a", "quoted, text", 'he said "no"', "multi\nline", "plain words here", "x" * 40], dtype=object)
    s = words[rng.integers(0, len(words), n)]
    f = np.round(rng.normal(0, 1e4, n), 4)
    nullk = rng.random(n) < 0.05
    tbl = pa.table({"k": pa.array(k, mask=nullk), "q": pa.array([None if i % 17 == 0 else int(v) for i, v in enumerate(q)], pa.int64()), "dt": pa.array(dt), "s": pa.array(s, pa.string()),
                    "f": pa.array(f), "b": pa.array(rng.random(n) < 0.5)})
    buf = io.BytesIO(); pcsv.write_csv(tbl, buf); data = buf.getvalue()
    fields = [("k", I64), ("q", I64), ("dt", D32), ("s", S), ("f", F64), ("b", B)]
    assert check(ctx, data, fields) == n
    assert check(ctx, data, fields, projection=["q", "f"], on_device=True) == n


@pytest.mark.parametrize("bad,what", [(b"a,b\n1,x\n", "not a number"), (b"a,b\n1\n", "short record"), (b'a,b\n1,"open\n', "unterminated quote"), (b"a,b\n300,1\n", "out of range")])
def test_malformed_input_is_an_error(ctx, bad, what):
    from dfgpu.csv import read_csv
    from dfgpu import DfgpuError
    t = 6 if what == "out of range" else 5
    with pytest.raises(DfgpuError):
        read_csv(ctx, bad, [("a", t, 0, 0), ("b", 5, 0, 0)])


def test_empty_inputs(ctx):
    from dfgpu.csv import read_csv
    for data in (b"", b"a,b\n"):
        got = read_csv(ctx, data, [("a", 5, 0, 0), ("b", 14, 0, 0)])
        assert [len(a.to_arrow()) for a in got] == [0, 0]


def test_csv_exec_plan_filter_aggregate(ctx):
    """CsvExec -> FilterExec -> AggregateExec over lineitem-shaped text == pyarrow's group_by over the table pyarrow reads; the cut into pieces and the number
    of partitions do not change the result.  Then every column back, batch by batch == the file."""
    import pyarrow.compute as pc
    from dfgpu import capi, physical_plan as ops
    rng = np.random.default_rng(11)
    n = 50_000
    tbl = pa.table({"k": pa.array(rng.integers(0, 50, n)), "flag": pa.array(np.array(["A", "N", "R", "has,comma", "two\nlines"], dtype=object)[rng.integers(0, 5, n)], pa.string()),
                    "qty": pa.array(rng.integers(1, 51, n)), "d": pa.array(rng.integers(8000, 10600, n).astype("datetime64[D]"))})
    buf = io.BytesIO(); pcsv.write_csv(tbl, buf); data = buf.getvalue()
    file_schema = [("k", capi.INT64, 0, 0), ("flag", capi.UTF8, 0, 0), ("qty", capi.INT64, 0, 0), ("d", capi.DATE32, 0, 0)]
    C, F = ops.Column, ops.Field
    tc = ops.TaskContext(ctx, 8192)
    sel = tbl.filter(pc.less(tbl["qty"], 25))
    want = {r["k"]: (r["qty_sum"], r["qty_count"]) for r in sel.group_by("k").aggregate([("qty", "sum"), ("qty", "count")]).to_pylist()}
    for parts, piece in ((1, 0), (1, 64 << 10), (3, 100 << 10), (4, 1 << 10)):
        scan = ops.CsvExec(data, file_schema, projection=["k", "qty"], partitions=parts, batch_bytes=piece)
        assert scan.schema().names() == ["k", "qty"]
        pred = ops.BinaryExpr(C("qty", 1), "<", ops.Literal(25, pa.int64()))
        agg = ops.AggregateExec("Single", [(C("k", 0), "k")], [ops.AggregateFunctionExpr("SUM", C("qty", 1), "s", input_field=F("qty", capi.INT64)), ops.AggregateFunctionExpr("COUNT", C("qty", 1), "c")],
                                ops.CoalescePartitionsExec(ops.FilterExec(pred, scan)))
        out = pa.concat_tables([b.to_arrow() for b in agg.execute(0, tc)])
        assert {r["k"]: (r["s"], r["c"]) for r in out.to_pylist()} == want, (parts, piece)
    scan = ops.CoalescePartitionsExec(ops.CsvExec(data, file_schema, partitions=2, batch_bytes=200 << 10))
    batches = [b.to_arrow() for b in scan.execute(0, tc)]
    assert len(batches) > 4
    key = [("k", "ascending"), ("qty", "ascending"), ("d", "ascending"), ("flag", "ascending")]
    rows = pa.concat_tables(batches).sort_by(key)
    assert rows.equals(reference_read(data, [("k", I64), ("flag", S), ("qty", I64), ("d", D32)]).sort_by(key))


def test_csv_exec_edge_files(ctx):
    """Header only, empty image, one record without a line feed, more partitions than pieces: CsvExec yields the rows once, or none."""
    from dfgpu import capi, physical_plan as ops
    sch = [("a", capi.INT64, 0, 0), ("b", capi.UTF8, 0, 0)]
    tc = ops.TaskContext(ctx, 8192)
    for data, rows in ((b"a,b\n", 0), (b"", 0), (b"a,b\n1,x", 1), (b"a,b\n1,x\n2,y\n", 2)):
        for parts in (1, 3):
            scan = ops.CsvExec(data, sch, partitions=parts, batch_bytes=4)
            out = [b.to_arrow() for b in ops.CoalescePartitionsExec(scan).execute(0, tc)] if parts > 1 else [b.to_arrow() for b in scan.execute(0, tc)]
            assert sum(b.num_rows for b in out) == rows, (data, parts)
    got = pa.concat_tables([b.to_arrow() for b in ops.CsvExec(b"a,b\n1,x\n2,y\n", sch, batch_bytes=4).execute(0, tc)])
    assert got["a"].to_pylist() == [1, 2] and got["b"].to_pylist() == ["x", "y"]


def test_escape_character(ctx):
    """CsvExec::escape (csv.rs:59, :304): the reference's escape.csv (core/tests/data, quotes escaped by a backslash) and generated records with escaped quotes, escaped
    backslashes, an escaped quote in front of the closing quote, escapes next to doubled quotes, and escapes across the 1 KB blocks of the record pass."""
    from dfgpu import physical_plan as ops, capi
    data = open(os.path.join(HERE, "data_escape.csv"), "rb").read()
    fields = [("c1", S), ("c2", S)]
    assert check(ctx, data, fields, escape="\\") == 10
    rows = [b'1,"a\\"b",x', b'2,"back\\\\slash",y', b'3,"ends with quote\\"",z', b'4,"both \\" and """,w', b'5,"line\nfeed \\" inside",v', b'6,plain,u']
    rows += [b'%d,"%s\\"%s",t' % (i, b"p" * (i % 700), b"q" * (i % 13)) for i in range(7, 400)]
    data = b"k,s,t\n" + b"\n".join(rows) + b"\n"
    fields = [("k", I64), ("s", S), ("t", S)]
    assert check(ctx, data, fields, escape="\\") == len(rows)
    check(ctx, data, fields, escape="\\", on_device=True, projection=["s"])
    # through CsvExec, cut into small pieces
    sch = [("k", capi.INT64, 0, 0), ("s", capi.UTF8, 0, 0), ("t", capi.UTF8, 0, 0)]
    tc = ops.TaskContext(ctx, 8192)
    out = pa.concat_tables([b.to_arrow() for b in ops.CsvExec(data, sch, batch_bytes=4096, escape="\\").execute(0, tc)])
    want = reference_read(data, fields, escape="\\")
    assert out["s"].combine_chunks().equals(want["s"].combine_chunks()) and out["k"].combine_chunks().equals(want["k"].combine_chunks())
