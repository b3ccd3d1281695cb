"""One interpreter for the transcribed known-answer fixtures (tests/golden/{decimal_slt,aggregates}.json), two back ends:
OracleEngine (oracle/pyoracle.py, CPU, pins the oracle) and DeviceEngine (libdfgpu through the C ABI, -m gpu).
Every operation takes / returns pyarrow arrays so that both back ends are checked by the same assertions."""
import decimal
import math

import numpy as np
import pyarrow as pa

OPCODE = {"+": 0, "-": 1, "*": 2, "/": 3, "%": 4, "=": 10, "!=": 11, "<": 12, "<=": 13, ">": 14, ">=": 15, "IS DISTINCT FROM": 16, "IS NOT DISTINCT FROM": 17, "AND": 20, "OR": 21}
AGG = {"SUM": 0, "AVG": 1, "COUNT": 2, "MIN": 3, "MAX": 4}


def pa_type(t):
    if isinstance(t, dict):
        p, s = t["decimal128"]
        return pa.decimal128(p, s)
    return {"int8": pa.int8(), "int16": pa.int16(), "int32": pa.int32(), "int64": pa.int64(), "uint8": pa.uint8(), "uint16": pa.uint16(), "uint32": pa.uint32(), "uint64": pa.uint64(), "float32": pa.float32(), "float64": pa.float64(),
            "bool": pa.bool_(), "utf8": pa.utf8(), "date32": pa.date32()}[t]


def make_array(t, values):
    typ = pa_type(t)
    if pa.types.is_decimal(typ):
        return pa.array([None if v is None else decimal.Decimal(int(v)).scaleb(-typ.scale) for v in values], type=typ)
    if pa.types.is_floating(typ):
        neg_nan = np.frombuffer(np.array([0xFFF8000000000000], dtype=np.uint64).tobytes(), dtype=np.float64)[0]      # -f64::NAN: sign bit set
        return pa.array([None if v is None else (math.nan if v == "NaN" else neg_nan if v == "-NaN" else float(v)) for v in values], type=typ)
    return pa.array(values, type=typ)


def type_string(typ):
    """arrow_typeof spelling used by the .slt files"""
    if pa.types.is_decimal(typ):
        return f"Decimal128({typ.precision}, {typ.scale})"
    return {pa.float64(): "Float64", pa.int64(): "Int64", pa.bool_(): "Boolean"}[typ]


class OracleEngine:
    name = "oracle"

    def __init__(self):
        from oracle import pyoracle as po
        self.po = po

    def binary(self, op, l, r, ls=False, rs=False):
        # arrow-arith / arrow-ord unpack dictionary operands before computing (the reference's expected arrays are plain): the restatement takes the values
        l, r = (x.dictionary_decode() if pa.types.is_dictionary(x.type) else x for x in (l, r))
        return self.po.binary(op, l, r, l_scalar=ls, r_scalar=rs)

    def cast(self, a, typ):
        return self.po.cast(a, typ)

    def in_list(self, a, lst, negated):
        return self.po.in_list(a, lst, negated)

    def filter(self, a, mask):
        return self.po.filter_(a, mask)

    def take(self, a, idx):
        return self.po.take(a, np.asarray(idx, dtype=np.int64))

    def sort_indices(self, cols, desc, nulls_first, fetch=None):
        return np.asarray(self.po.lexsort_to_indices(cols, desc, nulls_first, fetch))

    def group_ids(self, cols):
        g = self.po.Groups([c.type for c in cols])
        ids = np.asarray(g.intern(cols))
        return ids, len(g), g.emit()

    def accumulate(self, func, values, in_type, gids, total, filt=None, merge_states=None):
        """returns (evaluate(), state()) of a fresh accumulator after one update_batch (or one merge_batch of `merge_states`)"""
        a = self.po.Acc(func, in_type)
        if merge_states is not None:
            a.merge_batch(merge_states, np.asarray(gids, dtype=np.int64), filt, total)
        else:
            a.update_batch(values, np.asarray(gids, dtype=np.int64), filt, total)
        st = a.state()
        a2 = self.po.Acc(func, in_type)
        if merge_states is not None:
            a2.merge_batch(merge_states, np.asarray(gids, dtype=np.int64), filt, total)
        else:
            a2.update_batch(values, np.asarray(gids, dtype=np.int64), filt, total)
        return a2.evaluate(), st


class DeviceEngine:
    name = "device"

    def __init__(self, ctx):
        import dfgpu
        self.ctx, self.dfgpu = ctx, dfgpu

    def _code(self, typ):
        c = self.dfgpu.capi
        if pa.types.is_decimal(typ):
            return c.DECIMAL128, typ.precision, typ.scale
        return {pa.int16(): c.INT16, pa.int32(): c.INT32, pa.int64(): c.INT64, pa.uint32(): c.UINT32, pa.uint64(): c.UINT64, pa.float32(): c.FLOAT32, pa.float64(): c.FLOAT64,
                pa.bool_(): c.BOOL, pa.utf8(): c.UTF8, pa.date32(): c.DATE32}[typ], 0, 0

    def binary(self, op, l, r, ls=False, rs=False):
        return self.ctx.binary(OPCODE[op], self.ctx.from_arrow(l), self.ctx.from_arrow(r), ls, rs).to_arrow()

    def in_list(self, a, lst, negated):
        return self.ctx.in_list(self.ctx.from_arrow(a), self.ctx.from_arrow(lst), negated).to_arrow()

    def cast(self, a, typ):
        t, p, s = self._code(typ)
        return self.ctx.cast(self.ctx.from_arrow(a), t, p, s).to_arrow()

    def filter(self, a, mask):
        return self.ctx.filter(self.ctx.from_arrow(a), self.ctx.from_arrow(mask)).to_arrow()

    def take(self, a, idx):
        return self.ctx.take(self.ctx.from_arrow(a), self.ctx.from_arrow(pa.array(np.asarray(idx, dtype=np.uint32)))).to_arrow()

    def sort_indices(self, cols, desc, nulls_first, fetch=None):
        return self.ctx.sort_to_indices([self.ctx.from_arrow(c) for c in cols], desc, nulls_first, fetch).to_numpy().astype(np.int64)

    def group_ids(self, cols):
        g = self.dfgpu.GroupValues(self.ctx, len(cols))
        ids = g.intern([self.ctx.from_arrow(c) for c in cols]).to_numpy().astype(np.int64)
        return ids, len(g), [c.to_arrow() for c in g.emit()] if len(g) else []

    def accumulate(self, func, values, in_type, gids, total, filt=None, merge_states=None):
        t, p, s = self._code(in_type)
        out = []
        for what in ("state", "evaluate"):
            a = self.dfgpu.GroupsAccumulator(self.ctx, AGG[func], t, p, s)
            g = self.ctx.from_arrow(pa.array(np.asarray(gids, dtype=np.uint32)))
            f = self.ctx.from_arrow(filt) if filt is not None else None
            if merge_states is not None:
                a.merge_batch([self.ctx.from_arrow(x) for x in merge_states], g, f, total)
            else:
                a.update_batch(self.ctx.from_arrow(values) if values is not None else None, g, f, total)
            out.append([x.to_arrow() for x in a.state()] if what == "state" else a.evaluate().to_arrow())
        return out[1], out[0]


# ------------------------------------------------------------------ decimal.slt interpreter
def slt_table(fix):
    cols = {}
    for name, c in fix["table"]["columns"].items():
        if c["type"] == "decimal128":
            cols[name] = make_array({"decimal128": [c["precision"], c["scale"]]}, c["unscaled"])
        else:
            cols[name] = pa.array(c["values"], type=pa_type(c["type"]))
    return cols


def slt_operand(cols, spec):
    if "column" in spec:
        return cols[spec["column"]], False
    lit = spec["literal"]
    return make_array({"decimal128": [lit["precision"], lit["scale"]]}, [lit["unscaled"]]), True


def parse_token(tok, typ):
    if pa.types.is_decimal(typ):
        return decimal.Decimal(tok)
    if pa.types.is_floating(typ):
        return float(tok)
    if pa.types.is_boolean(typ):
        return tok == "true"
    return int(tok)


def rows_as_values(arrays):
    cols = [a.to_pylist() for a in arrays]
    return [tuple(c[i] for c in cols) for i in range(len(cols[0]))] if cols else []


def slt_predicate(eng, cols, case):
    l, ls = slt_operand(cols, case["left"])
    r, rs = slt_operand(cols, case["right"])
    if case.get("cast_left"):
        t = pa.decimal128(case["cast_left"]["precision"], case["cast_left"]["scale"])
        if not ls:
            l = eng.cast(l, t)
        if not rs and r.type != t:
            r = eng.cast(r, t)
    return eng.binary(case["op"], l, r, ls, rs)


def run_slt_case(eng, fix, case):
    """returns nothing; asserts the case's known answer"""
    cols = slt_table(fix)
    kind = case["kind"]
    if kind in ("binary", "binary_values"):
        if kind == "binary":
            (l, ls), (r, rs) = slt_operand(cols, case["left"]), slt_operand(cols, case["right"])
        else:
            mk = lambda d: make_array({"decimal128": [d["precision"], d["scale"]]}, d["unscaled"])
            l, ls, r, rs = mk(case["left"]), False, mk(case["right"]), False
        got = eng.binary(case["op"], l, r, ls, rs)
        if "expected_type" in case:
            assert type_string(got.type) == case["expected_type"], f"{case['name']}: {got.type}"
        want = sorted(decimal.Decimal(t) for t in case["expected_rowsort"])
        assert sorted(got.to_pylist()) == want, case["name"]
    elif kind == "filter":
        mask = slt_predicate(eng, cols, case)
        sel = [eng.filter(cols[c], mask) for c in case["select"]]
        want = sorted(tuple(parse_token(t, cols[c].type) for t, c in zip(row, case["select"])) for row in case["expected_rowsort"])
        assert sorted(rows_as_values(sel)) == want, case["name"]
    elif kind == "aggregate":
        v = cols[case["column"]]
        filt = None
        if case["where_c4_equals"] is not None:
            filt = cols["c4"] if case["where_c4_equals"] else eng.binary("=", cols["c4"], pa.array([False]), False, True)
        got, _ = eng.accumulate(case["func"], v, v.type, np.zeros(len(v), dtype=np.int64), 1, filt)
        assert type_string(got.type) == case["expected_type"], f"{case['name']}: {got.type}"
        assert got.to_pylist() == [decimal.Decimal(case["expected"])], f"{case['name']}: {got.to_pylist()}"
    elif kind == "sort":
        mask = slt_predicate(eng, cols, case["where"])
        kept = {c: eng.filter(a, mask) for c, a in cols.items()}
        idx = eng.sort_indices([kept[k] for k, _ in case["keys"]], [d for _, d in case["keys"]], [d for _, d in case["keys"]], case["fetch"])
        # SQL default null placement (NULLS LAST for ASC, NULLS FIRST for DESC) == nulls_first = descending; the table has no NULLs anyway
        out = [eng.take(kept[c], idx) for c in case["select"]]
        got = rows_as_values(out)
        want = [tuple(parse_token(t, cols[c].type) for t, c in zip(row, case["select"])) for row in case["expected_ordered"]]
        assert len(got) == len(want), case["name"]
        key_pos = [case["select"].index(k) for k, _ in case["keys"]]
        assert [tuple(r[p] for p in key_pos) for r in got] == [tuple(r[p] for p in key_pos) for r in want], f"{case['name']}: key order"
        if case["fetch"] is None or case["fetch"] >= len(kept["c1"]):
            assert sorted(got, key=repr) == sorted(want, key=repr), f"{case['name']}: rows"          # ties: any order, same multiset
    elif kind == "groupby":
        keys = [cols[k] for k in case["keys"]]
        ids, n, emitted = eng.group_ids(keys)
        got_counts, _ = eng.accumulate("COUNT", None, pa.int64(), ids, n)
        got = sorted((c,) + k for c, k in zip(got_counts.to_pylist(), rows_as_values(emitted)))
        want = sorted((row[0],) + tuple(parse_token(str(t).lower() if isinstance(t, bool) else t, cols[k].type) for t, k in zip(row[1:], case["keys"])) for row in case["expected"])
        assert got == want, case["name"]
    else:
        raise AssertionError(kind)


# ------------------------------------------------------------------ aggregates.json interpreter
def expected_scalar(case):
    typ = pa_type(case["expected_type"])
    return make_array(case["expected_type"], [case["expected"]]) if pa.types.is_decimal(typ) or pa.types.is_floating(typ) else pa.array([case["expected"]], type=typ)


def run_scalar_case(eng, case):
    arr = make_array(case["type"], case["values"])
    coerced = pa_type(case["coerced"])
    if arr.type != coerced:
        arr = eng.cast(arr, coerced)                  # assert_aggregate: try_cast(col, coerced type) (expressions/mod.rs:188-193)
    values = None if case["func"] == "COUNT" and len(arr) == 0 else arr
    got, _ = eng.accumulate(case["func"], arr, coerced, np.zeros(len(arr), dtype=np.int64), 1)
    want = expected_scalar(case)
    assert got.type == want.type, f"{case['name']}: {got.type} vs {want.type}"
    assert got.to_pylist() == want.to_pylist(), f"{case['name']}: {got.to_pylist()} vs {want.to_pylist()}"


def run_grouped_case(eng, fix, case):
    data = fix["some_data"]
    batches = [{c: pa.array(b[c], type=pa_type(data[c]["type"])) for c in ("a", "b")} for b in data["batches"]]
    agg = case["agg"]
    # Partial (one partition over both batches): intern keys batch by batch, first-seen ids; accumulate
    all_keys = [pa.concat_arrays([b[k] for b in batches]) for k in case["group_by"]]
    ids, n, emitted = eng.group_ids(all_keys)
    values = pa.concat_arrays([b[agg["column"]] for b in batches]) if agg["column"] else None
    in_type = pa_type(agg["type"])
    _, state = eng.accumulate(agg["func"], values, in_type, ids, n)
    partial = sorted(rows_as_values(list(emitted) + list(state)))
    want_partial = sorted(tuple(r) for r in case["partial_sorted"])
    # the reference lists AVG state as (count, sum); COUNT state as (count)
    assert partial == want_partial, f"{case['name']}: partial {partial}"
    # Final: merge the partial state rows
    ids2, n2, emitted2 = eng.group_ids(list(emitted))
    final, _ = eng.accumulate(agg["func"], None, in_type, ids2, n2, merge_states=list(state))
    got = sorted(rows_as_values(list(emitted2) + [final]))
    assert got == sorted(tuple(r) for r in case["final_sorted"]), f"{case['name']}: final {got}"


def run_sort_case(eng, case):
    cols = [make_array(c["type"], c["values"]) for c in case["columns"]]
    idx = eng.sort_indices(cols, [c["descending"] for c in case["columns"]], [c["nulls_first"] for c in case["columns"]])
    out = [eng.take(c, idx).to_pylist() for c in cols]
    norm = lambda v: None if v is None else ("NaN" if isinstance(v, float) and math.isnan(v) else v)
    got = [[norm(out[0][i]), norm(out[1][i])] for i in range(len(idx))]
    assert got == case["expected"], f"{case['name']}: {got}"


def run_clickbench_case(eng, fix, case):
    """GROUP BY (or one global group) with SUM / COUNT(*) / AVG; arguments cast to the coerced type first; rows compared as a
    multiset, Float64 results within 1e-9 relative (the parity contract for float SUM / AVG)."""
    cols = {n: pa.array(c["values"], type=pa_type(c["type"])) for n, c in fix["clickbench"]["columns"].items()}
    n = len(cols["UserID"])
    if case["group_by"]:
        ids, ng, emitted = eng.group_ids([cols[k] for k in case["group_by"]])
    else:
        ids, ng, emitted = np.zeros(n, dtype=np.int64), 1, []
    outs = list(emitted)
    for func, column, coerced in case["aggs"]:
        t = pa_type(coerced)
        v = None
        if column is not None:
            v = cols[column] if cols[column].type == t else eng.cast(cols[column], t)
        res, _ = eng.accumulate(func, v, t, ids, ng)
        outs.append(res)
    got = sorted(rows_as_values(outs), key=repr)
    want = sorted((tuple(r) for r in case["expected_rowsort"]), key=repr)
    assert len(got) == len(want), case["name"]
    for g, w in zip(got, want):
        for a, b in zip(g, w):
            if isinstance(b, float):
                assert a == b or abs(a - b) <= 1e-9 * abs(b), f"{case['name']}: {a} vs {b}"
            else:
                assert a == b, f"{case['name']}: {g} vs {w}"


# ------------------------------------------------------------------ unit_vectors.json (binary.rs / sort.rs / repartition)
def vector_array(spec):
    if "dict" in spec:
        d = spec["dict"]
        return pa.DictionaryArray.from_arrays(pa.array(d["keys"], type=pa_type(d["keys_type"])), vector_array(d["values"]))
    return make_array(spec["type"], spec["values"])


def run_binary_vector(eng, case):
    l, r, want = vector_array(case["left"]), vector_array(case["right"]), vector_array(case["expected"])
    got = eng.binary(case["op"], l, r, case["left_scalar"], case["right_scalar"])
    if pa.types.is_dictionary(got.type):
        got = got.dictionary_decode()
    assert got.type == want.type, (case["name"], got.type, want.type)
    assert got.equals(want), (case["name"], got.to_pylist(), want.to_pylist())


# ------------------------------------------------------------------ groupby_order_slt.json (group_by.slt / aggregate.slt / order.slt VALUES cases)
def table_column(spec):
    if "dict" in spec:
        d = spec["dict"]
        enc = pa.array(d["values"], type=pa.utf8()).dictionary_encode()
        return pa.DictionaryArray.from_arrays(enc.indices.cast(pa_type(d["keys_type"])), enc.dictionary)
    return make_array(spec["type"], spec["values"])


def _norm(v):
    return decimal.Decimal(v) if isinstance(v, str) and v[:1] in "-0123456789" and v.replace(".", "").replace("-", "").isdigit() else v


def run_table_case(eng, fix, case):
    """GROUP BY keys + aggregate calls over a VALUES table; arguments cast to the planner's coerced type; rows compared as a multiset (rowsort), Float64 within
    1e-9 relative, everything else exactly (Decimal128 numerically: 15.150000 == 15.15)."""
    cols = {n: table_column(c) for n, c in fix["tables"][case["table"]]["columns"].items()}
    plain = (lambda a: a.dictionary_decode() if pa.types.is_dictionary(a.type) else a) if eng.name == "oracle" else (lambda a: a)   # the restatement takes the values of a dictionary column
    ids, ng, emitted = eng.group_ids([plain(cols[k]) for k in case["group_by"]])
    outs = [e.dictionary_decode() if pa.types.is_dictionary(e.type) else e for e in emitted]
    for func, column, coerced in case["aggs"]:
        t = pa_type(coerced)
        v = plain(cols[column])
        if not pa.types.is_dictionary(v.type) and v.type != t:
            v = eng.cast(v, t)
        res, _ = eng.accumulate(func, v, t, ids, ng)
        outs.append(res)
    key = lambda row: repr([None if x is None else str(x) for x in row[:len(case["group_by"])]])
    got = sorted(rows_as_values(outs), key=key)
    want = sorted((tuple(_norm(x) for x in r) for r in case["expected_rowsort"]), key=key)
    assert len(got) == len(want), (case["name"], got)
    for g, w in zip(got, want):
        assert len(g) == len(w), (case["name"], g, w)
        for a, b in zip(g, w):
            if isinstance(b, float):
                assert a is not None and (a == b or abs(a - b) <= 1e-9 * abs(b)), f"{case['name']}: {g} vs {w}"
            else:
                assert a == b, f"{case['name']}: {g} vs {w}"


def run_order_case(eng, fix, case):
    cols = [table_column(c) for c in fix["order"]["table"]["columns"].values()]
    idx = eng.sort_indices([cols[0]], [case["descending"]], [case["nulls_first"]])
    got = [list(r) for r in rows_as_values([eng.take(c, idx) for c in cols])]
    assert got == case["expected"], f"{case['name']}: {got}"


def run_in_list_vector(eng, case):
    a, lst = make_array(case["type"], case["values"]), make_array(case["type"], case["list"])
    got = eng.in_list(a, lst, case["negated"])
    assert got.to_pylist() == case["expected"], (case["name"], got.to_pylist())
