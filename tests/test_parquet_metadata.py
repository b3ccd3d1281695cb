"""CPU: the host side of the Parquet scan (csrc/parquet.hip: Thrift compact footer, schema -> Arrow types, row-group statistics) against what
pyarrow reports for the same files.  No device call: the footer is parsed without a ctx."""
import glob
import os

import pyarrow as pa
import pyarrow.parquet as pq
import pytest

import dfgpu
from dfgpu import capi
from dfgpu.parquet import ParquetFile

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parquet")
FILES = sorted(glob.glob(os.path.join(HERE, "*.parquet")))

ARROW = {pa.int8(): capi.INT8, pa.uint8(): capi.UINT8, pa.int16(): capi.INT16, pa.uint16(): capi.UINT16, pa.int32(): capi.INT32, pa.uint32(): capi.UINT32,
         pa.int64(): capi.INT64, pa.uint64(): capi.UINT64, pa.float32(): capi.FLOAT32, pa.float64(): capi.FLOAT64, pa.bool_(): capi.BOOL, pa.date32(): capi.DATE32,
         pa.string(): capi.UTF8}


def want_type(t):
    if pa.types.is_decimal128(t):
        return capi.DECIMAL128, t.precision, t.scale
    return ARROW.get(t, 0), 0, 0


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-8] for f in FILES])
def test_footer_schema_and_row_groups_match_pyarrow(path):
    ref = pq.ParquetFile(path)
    f = ParquetFile(None, path=path, utf8_dictionary=False)
    md = ref.metadata
    assert f.num_rows == md.num_rows and f.num_row_groups == md.num_row_groups
    leaves = [ref.schema.column(i) for i in range(len(ref.schema.names))]
    assert f.num_columns == md.num_columns
    flat = all(len(c.path.split(".")) == 1 for c in leaves)
    for i, c in enumerate(leaves):
        t, vt, prec, scale, nullable = f.column_type(i)
        if len(c.path.split(".")) > 1:
            assert t == 0                                   # nested leaf: outside the device scan
            continue
        assert f.column_names()[i] == c.name
        w = want_type(ref.schema_arrow.field(c.name).type)
        assert (vt, prec, scale) == w, (c.name, vt, w)
        assert nullable == (c.max_definition_level == 1)
    for g in range(md.num_row_groups):
        rg = md.row_group(g)
        assert f.row_group_rows(g) == rg.num_rows
        for i in range(rg.num_columns):
            cc = rg.column(i)
            assert f.chunk_bytes(g, i) == cc.total_compressed_size and f.chunk_bytes(g, i, True) == cc.total_uncompressed_size
            st = cc.statistics
            mn, mx, nc = f.column_stats(g, i)
            if st is not None and st.has_null_count:
                assert nc == st.null_count
            at = ref.schema_arrow.field(leaves[i].name).type if flat else None
            if flat and st is not None and st.has_min_max and cc.physical_type in ("INT32", "INT64") and not pa.types.is_decimal(at) and at not in (pa.uint32(), pa.uint64()):
                lo, hi = st.min, st.max
                if pa.types.is_date32(at):
                    import datetime
                    lo, hi = (lo - datetime.date(1970, 1, 1)).days, (hi - datetime.date(1970, 1, 1)).days
                assert (mn, mx) == (lo, hi), (leaves[i].name, mn, mx, lo, hi)
    f.close()


def test_dictionary_option_changes_the_reported_type_only_for_utf8():
    f = ParquetFile(None, path=os.path.join(HERE, "dict_snappy_v1.parquet"))
    names = f.column_names()
    assert f.column_type(names.index("low_card"))[:2] == (capi.DICTIONARY, capi.UTF8)
    assert f.column_type(names.index("i64"))[:2] == (capi.INT64, capi.INT64)


def test_corrupt_files_are_execution_errors(tmp_path):
    good = open(os.path.join(HERE, "dict_snappy_v1.parquet"), "rb").read()
    for name, data in (("short", good[:8]), ("magic", b"XXXX" + good[4:-4] + b"XXXX"), ("length", good[:-8] + (len(good) * 2).to_bytes(4, "little") + b"PAR1"),
                       ("thrift", good[:-200] + b"\xff" * 192 + good[-8:])):
        with pytest.raises(dfgpu.DfgpuError) as e:
            ParquetFile(None, data=data)
        assert e.value.kind == "Execution", name
    with pytest.raises(dfgpu.DfgpuError):
        ParquetFile(None, path=str(tmp_path / "missing.parquet"))
