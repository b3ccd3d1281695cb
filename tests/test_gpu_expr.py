"""-m gpu: PhysicalExpr kernels (BinaryExpr / Cast / Not / IsNull / InList / Negative) vs the CPU oracle, bit-exact
for integer / decimal / boolean results and for IEEE float arithmetic (same operations, same order)."""
import decimal

import numpy as np
import pyarrow as pa
import pytest

from oracle import pyoracle as po
from test_gpu_core import rand_array

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(7)
CMP = ["=", "!=", "<", "<=", ">", ">=", "IS DISTINCT FROM", "IS NOT DISTINCT FROM"]
OPCODE = {"+": 0, "-": 1, "*": 2, "/": 3, "%": 4, "=": 10, "!=": 11, "<": 12, "<=": 13, ">": 14, ">=": 15, "IS DISTINCT FROM": 16, "IS NOT DISTINCT FROM": 17, "AND": 20, "OR": 21}


def dev_binary(ctx, op, l, r, ls=False, rs=False):
    return ctx.binary(OPCODE[op], ctx.from_arrow(l), ctx.from_arrow(r), ls, rs).to_arrow()


def same(a, b):
    if pa.types.is_floating(a.type):
        return a.type == b.type and np.array_equal(np.asarray(a.is_null()), np.asarray(b.is_null())) and \
            np.array_equal(np.asarray(a.fill_null(0)).view(np.uint64 if a.type == pa.float64() else np.uint32),
                           np.asarray(b.fill_null(0)).view(np.uint64 if a.type == pa.float64() else np.uint32))
    return a.type == b.type and a.equals(b)


@pytest.mark.parametrize("kind", ["int8", "int32", "int64", "uint16", "uint64", "float32", "float64", "date32", "decimal", "utf8", "bool", "dict"])
@pytest.mark.parametrize("op", CMP)
def test_comparisons(ctx, kind, op):
    n = 3000
    l, r = rand_array(kind, n), rand_array(kind, n)
    if kind in ("utf8", "dict", "bool", "int8"):      # make equal pairs likely
        pass
    else:
        r = pa.array([lv if i % 3 == 0 else rv for i, (lv, rv) in enumerate(zip(l.to_pylist(), r.to_pylist()))], type=r.type)
    assert same(dev_binary(ctx, op, l, r), po.binary(op, l, r))
    s = r.slice(5, 1) if r.slice(5, 1).null_count == 0 else r.drop_null().slice(0, 1)
    assert same(dev_binary(ctx, op, l, s, rs=True), po.binary(op, l, s, r_scalar=True))
    assert same(dev_binary(ctx, op, s, l, ls=True), po.binary(op, s, l, l_scalar=True))


def test_float_total_order_nan_and_zero(ctx):
    l = pa.array([float("nan"), 0.0, -0.0, 1.0, float("inf"), None, float("nan")], type=pa.float64())
    r = pa.array([float("nan"), -0.0, 0.0, float("nan"), float("nan"), 1.0, 1.0], type=pa.float64())
    for op in CMP:
        assert same(dev_binary(ctx, op, l, r), po.binary(op, l, r))


@pytest.mark.parametrize("kind", ["int8", "int16", "int32", "int64", "uint8", "uint32", "uint64", "float32", "float64"])
@pytest.mark.parametrize("op", ["+", "-", "*"])
def test_wrapping_arithmetic(ctx, kind, op):
    l, r = rand_array(kind, 4000), rand_array(kind, 4000)
    assert same(dev_binary(ctx, op, l, r), po.binary(op, l, r))
    s = r.drop_null().slice(0, 1)
    assert same(dev_binary(ctx, op, l, s, rs=True), po.binary(op, l, s, r_scalar=True))


@pytest.mark.parametrize("kind", ["int32", "int64", "uint32", "float64"])
@pytest.mark.parametrize("op", ["/", "%"])
def test_division(ctx, kind, op):
    import dfgpu
    l = rand_array(kind, 3000)
    r = rand_array(kind, 3000)
    if kind != "float64":
        r = pa.array([None if v is None else (v if v != 0 else 1) for v in r.to_pylist()], type=r.type)
    assert same(dev_binary(ctx, op, l, r), po.binary(op, l, r))
    if kind != "float64":
        z = pa.array([0], type=r.type)
        with pytest.raises(dfgpu.DfgpuError) as e:
            dev_binary(ctx, op, l.drop_null(), z, rs=True)
        assert "Divide by zero" in str(e.value)
        with pytest.raises(po.OracleError):
            po.binary(op, l.drop_null(), z, r_scalar=True)


def dec_array(n, p, s, lim, null_frac=0.1):
    vals = RNG.integers(-lim, lim, n)
    mask = RNG.random(n) < null_frac
    return pa.array([None if m else decimal.Decimal(int(v)).scaleb(-s) for v, m in zip(vals, mask)], type=pa.decimal128(p, s))


@pytest.mark.parametrize("op", ["+", "-", "*", "/", "%"])
@pytest.mark.parametrize("shape", [((15, 2), (15, 2)), ((20, 0), (15, 2)), ((10, 4), (18, 1)), ((38, 10), (38, 10))])
def test_decimal_arithmetic_types_and_values(ctx, op, shape):
    (p1, s1), (p2, s2) = shape
    l, r = dec_array(2000, p1, s1, 10**9), dec_array(2000, p2, s2, 10**9)
    if op in ("/", "%"):
        r = pa.array([None if v is None else (v if v != 0 else decimal.Decimal(1)) for v in r.to_pylist()], type=r.type)
    want = po.binary(op, l, r)
    got = dev_binary(ctx, op, l, r)
    assert got.type == want.type, f"result dtype {got.type} vs {want.type}"
    assert got.equals(want)


def test_q1_q3_projection_dtype_pins(ctx):
    """l_extendedprice * (1 - l_discount): Decimal128(15,2) * (Decimal128(20,0) - Decimal128(15,2)) = Decimal128(38,4)
    (plan dtypes of sqllogictest/test_files/tpch/q1.slt.part / q3.slt.part; decimal.slt:209-211,:262-264)."""
    ext, disc = dec_array(1000, 15, 2, 10**9, 0), pa.array([decimal.Decimal(int(d)).scaleb(-2) for d in RNG.integers(0, 11, 1000)], type=pa.decimal128(15, 2))
    one = pa.array([decimal.Decimal(1)], type=pa.decimal128(20, 0))
    sub = dev_binary(ctx, "-", one, disc, ls=True)
    assert sub.type == pa.decimal128(23, 2)
    rev = ctx.binary(OPCODE["*"], ctx.from_arrow(ext), ctx.from_arrow(sub)).to_arrow()
    assert rev.type == pa.decimal128(38, 4)
    want = [e * (1 - d) for e, d in zip(ext.to_pylist(), disc.to_pylist())]
    assert rev.to_pylist() == want


def test_decimal_overflow_is_an_error(ctx):
    import dfgpu
    big = pa.array([decimal.Decimal(10**37)], type=pa.decimal128(38, 0))
    with pytest.raises(dfgpu.DfgpuError) as e:
        dev_binary(ctx, "*", big, big)
    assert e.value.kind == "Execution" and "overflow" in str(e.value).lower()
    with pytest.raises(po.OracleError):
        po.binary("*", big, big)


def test_deferred_flag_region_raises_when_left_and_before_export(ctx):
    """Option defer_flag_checks (include/dfgpu.h): inside a region the kernel error surfaces when the region is left, and
    never later than an export of device data; afterwards the context is clean again."""
    import dfgpu
    big = pa.array([decimal.Decimal(10**37)], type=pa.decimal128(38, 0))
    ok = pa.array([decimal.Decimal(3)], type=pa.decimal128(38, 0))
    ctx.set_option("defer_flag_checks", 1)
    ctx.set_option("defer_flag_checks", 1)           # regions nest
    dev_binary(ctx, "*", ok, ok)
    bad = ctx.from_arrow(big)
    ctx.binary(OPCODE["*"], bad, bad)                # no error yet
    ctx.set_option("defer_flag_checks", 0)           # inner region left: still deferred
    with pytest.raises(dfgpu.DfgpuError) as e:
        ctx.set_option("defer_flag_checks", 0)
    assert e.value.kind == "Execution" and "overflow" in str(e.value).lower()
    ctx.set_option("defer_flag_checks", 1)
    try:
        out = ctx.binary(OPCODE["*"], bad, bad)
        with pytest.raises(dfgpu.DfgpuError):
            out.to_arrow()
    finally:
        ctx.set_option("defer_flag_checks", 0)
    assert dev_binary(ctx, "*", ok, ok).to_pylist() == [decimal.Decimal(9)]


@pytest.mark.parametrize("op", CMP[:6])
def test_dictionary_column_vs_plain_scalar_uses_dictionary_predicate(ctx, op):
    """`c_mktsegment = 'BUILDING'` shape: dictionary-encoded column against a plain Utf8 / Int64 literal (either side).
    The device evaluates the predicate on the dictionary values and maps the codes; same answer as on the decoded column,
    NULL codes and a NULL dictionary VALUE included."""
    words = pa.array(["AUTOMOBILE", "BUILDING", None, "FURNITURE", "MACHINERY", "HOUSEHOLD"], type=pa.utf8())
    codes = pa.array(RNG.integers(0, 6, 5000).astype(np.int8), mask=RNG.random(5000) < 0.1)
    col = pa.DictionaryArray.from_arrays(codes, words)
    lit = pa.array(["BUILDING"], type=pa.utf8())
    plain = col.cast(pa.utf8())
    assert same(dev_binary(ctx, op, col, lit, rs=True), po.binary(op, plain, lit, r_scalar=True))
    assert same(dev_binary(ctx, op, lit, col, ls=True), po.binary(op, lit, plain, l_scalar=True))
    nums = pa.DictionaryArray.from_arrays(pa.array(RNG.integers(0, 4, 3000).astype(np.int32)), pa.array([10, 20, 30, 40], type=pa.int64()))
    k = pa.array([30], type=pa.int64())
    assert same(dev_binary(ctx, op, nums, k, rs=True), po.binary(op, nums.cast(pa.int64()), k, r_scalar=True))


@pytest.mark.parametrize("op", ["AND", "OR"])
def test_kleene_logic(ctx, op):
    l, r = rand_array("bool", 5000, 0.3), rand_array("bool", 5000, 0.3)
    assert same(dev_binary(ctx, op, l, r), po.binary(op, l, r))
    for s in [pa.array([True]), pa.array([False]), pa.array([None], type=pa.bool_())]:
        assert same(dev_binary(ctx, op, l, s, rs=True), po.binary(op, l, s, r_scalar=True))


def test_not_isnull_negative_inlist(ctx):
    b = rand_array("bool", 3000, 0.2)
    assert same(ctx.not_(ctx.from_arrow(b)).to_arrow(), po.not_(b))
    for kind in ["int64", "utf8", "dict", "decimal"]:
        a = rand_array(kind, 3000, 0.3)
        assert same(ctx.is_null(ctx.from_arrow(a)).to_arrow(), po.is_null(a))
        assert same(ctx.is_null(ctx.from_arrow(a), True).to_arrow(), po.is_null(a, True))
    for kind in ["int32", "int64", "float64", "decimal"]:
        a = rand_array(kind, 3000, 0.2)
        assert same(ctx.negative(ctx.from_arrow(a)).to_arrow(), po.negative(a))
    a = pa.array(RNG.integers(0, 20, 3000), mask=RNG.random(3000) < 0.1)
    for lst in [pa.array([1, 5, 7]), pa.array([1, None, 7]), pa.array([], type=pa.int64())]:
        for neg in (False, True):
            assert same(ctx.in_list(ctx.from_arrow(a), ctx.from_arrow(lst), neg).to_arrow(), po.in_list(a, lst, neg))
    u = rand_array("utf8", 2000, 0.2)
    lst = pa.array(["BUILDING0", "ASIA3", "a1"])
    assert same(ctx.in_list(ctx.from_arrow(u), ctx.from_arrow(lst)).to_arrow(), po.in_list(u, lst))


CASTS = [("int32", pa.int64()), ("int64", pa.float64()), ("int8", pa.decimal128(10, 2)), ("int64", pa.decimal128(38, 4)), ("decimal", pa.decimal128(20, 4)),
         ("decimal", pa.decimal128(15, 0)), ("decimal", pa.float64()), ("decimal", pa.int64()), ("float32", pa.float64()), ("date32", pa.int32()),
         ("int32", pa.date32()), ("uint8", pa.int32()), ("bool", pa.int32()), ("int16", pa.float32()), ("dict", None)]


@pytest.mark.parametrize("kind,to", [c for c in CASTS if c[1] is not None])
def test_cast(ctx, kind, to):
    import dfgpu
    a = rand_array(kind, 3000, 0.15)
    code = {pa.int32(): 4, pa.int64(): 5, pa.float32(): 10, pa.float64(): 11, pa.date32(): 12}.get(to)
    if pa.types.is_decimal(to):
        got = ctx.cast(ctx.from_arrow(a), dfgpu.capi.DECIMAL128, to.precision, to.scale).to_arrow()
    else:
        got = ctx.cast(ctx.from_arrow(a), code).to_arrow()
    assert same(got, po.cast(a, to))


def test_cast_overflow_is_an_error(ctx):
    import dfgpu
    a = pa.array([1, 2, 300], type=pa.int32())
    with pytest.raises(dfgpu.DfgpuError) as e:
        ctx.cast(ctx.from_arrow(a), dfgpu.capi.INT8)
    assert e.value.kind == "Execution"
    with pytest.raises(po.OracleError):
        po.cast(a, pa.int8())
