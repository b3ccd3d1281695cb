"""CPU: the oracle's restatement of arrow-csv's record rules (oracle/pyoracle.py csv_records / csv_column) against pyarrow.csv on the data files the reference's own tests
read (core/tests/tpch-csv/*.csv, core/tests/data/*.csv; committed under tests/golden/csv) and on hand-written edge cases -- the pin the device tests of test_gpu_csv.py rest on."""
import os

import pyarrow as pa
import pytest

from oracle import pyoracle as po
from test_gpu_csv import DATA, HERE, I64, TPCH, reference_read


def columns_equal(data, fields, delimiter=",", has_header=True):
    want = reference_read(data, fields, delimiter, has_header)
    recs = po.csv_records(data, delimiter, has_header=has_header)
    assert len(recs) == want.num_rows
    for i, (name, t) in enumerate(fields):
        got = po.csv_column(recs, i, t)
        w = want[name].combine_chunks() if has_header else want.column(i).combine_chunks()
        assert got.equals(w), name


@pytest.mark.parametrize("table", sorted(TPCH))
def test_tpch_csv_files(table):
    columns_equal(open(os.path.join(HERE, table + ".csv"), "rb").read(), TPCH[table])


@pytest.mark.parametrize("name", sorted(DATA))
def test_data_csv_files(name):
    fields, delim = DATA[name]
    data = open(os.path.join(HERE, name + ".csv"), "rb").read()
    if fields is None:
        fields = [("column_%d" % (i + 1), I64) for i in range(len(data.split(b"\n")[0].split(b",")))]
    columns_equal(data, fields, delim, has_header=name not in ("data_one_col", "data_wide_rows"))


def test_quoting_crlf_blank_lines_and_missing_final_newline():
    fields = [("k", I64), ("s", pa.string()), ("f", pa.float64())]
    rows = [b'1,plain,1.5', b'-2,"with, comma",-0.25', b',"say ""hi"" twice",', b'4,"line\nfeed inside",1e3', b'5,"",2.5E-3', b'6,"""",0.000']
    for eol in (b"\n", b"\r\n"):
        for tail in (eol, b""):
            data = b"k,s,f" + eol + eol.join(rows) + tail
            columns_equal(data, fields)
            columns_equal(eol + data + eol + eol, fields)


def test_escape_csv_file():
    data = open(os.path.join(HERE, "data_escape.csv"), "rb").read()
    want = reference_read(data, [("c1", pa.string()), ("c2", pa.string())], escape="\\")
    recs = po.csv_records(data, escape="\\")
    assert po.csv_column(recs, 0, pa.string()).equals(want["c1"].combine_chunks()) and po.csv_column(recs, 1, pa.string()).equals(want["c2"].combine_chunks())
    assert recs[0][1][0] == 'value"0'
