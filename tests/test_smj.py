"""SortMergeJoinExec: the oracle restatement (two-cursor merge, oracle/pyoracle.py sort_merge_join) pinned on the reference's own tests
(physical-plan/src/joins/sort_merge_join.rs:1787-2448, rows and row ORDER), and -- with -m gpu -- the device operator against the same vectors and against
the oracle on random sorted inputs with duplicate and NULL keys."""
import numpy as np
import pyarrow as pa
import pytest

from helpers import load_golden

SMJ = load_golden("unit_vectors.json")["sort_merge_join"]
SMJ_FULL = load_golden("unit_vectors.json")["sort_merge_join_full"]          # the reference compares sorted rows for JoinType::Full
skey = lambda r: [(-1, 0) if v is None else (0, v) for v in r]


def cols(vals):
    return [pa.array(v, type=pa.int32()) for v in vals]


def rows(columns):
    c = [x.to_pylist() for x in columns]
    return [list(r) for r in zip(*c)] if c else []


@pytest.mark.parametrize("case", SMJ, ids=[c["name"] for c in SMJ])
def test_oracle_sort_merge_join_reference_cases(case):
    from oracle import pyoracle as po
    got = po.sort_merge_join(cols(case["left"]), cols(case["right"]), [tuple(x) for x in case["on"]], case["join_type"], case.get("descending", False), case.get("nulls_first", True),
                             case.get("null_equals_null", False))
    assert rows(got) == case["expected"]


@pytest.mark.parametrize("case", SMJ_FULL, ids=[c["name"] for c in SMJ_FULL])
def test_oracle_sort_merge_join_full_reference_cases(case):
    from oracle import pyoracle as po
    got = po.sort_merge_join(cols(case["left"]), cols(case["right"]), [tuple(x) for x in case["on"]], "Full")
    assert sorted(rows(got), key=skey) == sorted(case["expected"], key=skey)


def device_join(ctx, left_batches, right_batches, on, jt, nen=False, lnames=("a1", "b1", "c1"), rnames=("a2", "b2", "c2")):
    from dfgpu import physical_plan as ops
    tab = lambda c, names: ops.batch_from_arrow(ctx, pa.table(dict(zip(names, c))))
    mem = lambda batches, names: ops.MemoryExec([[tab(b, names) for b in batches]], tab(batches[0], names).schema)
    plan = ops.SortMergeJoinExec(mem(left_batches, lnames), mem(right_batches, rnames), [(ops.Column(lnames[l], l), ops.Column(rnames[r], r)) for l, r in on], jt, nen)
    out = list(plan.execute(0, ops.TaskContext(ctx, 8192)))
    if not out:
        return []
    t = [pa.concat_arrays([b.columns[i].to_arrow() for b in out]) for i in range(out[0].num_columns)]
    return rows(t)


def split(columns, sizes):
    out, o = [], 0
    for n in sizes:
        out.append([c.slice(o, n) for c in columns]); o += n
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("case", SMJ, ids=[c["name"] for c in SMJ])
def test_device_sort_merge_join_reference_cases(ctx, case):
    l, r = cols(case["left"]), cols(case["right"])
    lb = split(l, case["left_batches"]) if "left_batches" in case else [l]
    rb = split(r, case["right_batches"]) if "right_batches" in case else [r]
    assert device_join(ctx, lb, rb, case["on"], case["join_type"], case.get("null_equals_null", False)) == case["expected"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", SMJ_FULL, ids=[c["name"] for c in SMJ_FULL])
def test_device_sort_merge_join_full_reference_cases(ctx, case):
    l, r = cols(case["left"]), cols(case["right"])
    lb = split(l, case["left_batches"]) if "left_batches" in case else [l]
    rb = split(r, case["right_batches"]) if "right_batches" in case else [r]
    assert sorted(device_join(ctx, lb, rb, case["on"], "Full"), key=skey) == sorted(case["expected"], key=skey)


@pytest.mark.gpu
@pytest.mark.parametrize("jt", ["Inner", "Left", "Right", "LeftSemi", "LeftAnti", "RightAnti", "Full"])
@pytest.mark.parametrize("seed,nl,nr,nen", [(1, 300, 200, False), (2, 50, 400, True), (3, 1, 1, False), (4, 257, 0, False)])
def test_device_sort_merge_join_equals_oracle_on_sorted_inputs(ctx, jt, seed, nl, nr, nen):
    from oracle import pyoracle as po
    rng = np.random.default_rng(seed)

    def side(n):
        k = np.sort(rng.integers(0, 40, n)).astype(np.int32)
        mask = np.zeros(n, dtype=bool); mask[n - n // 10:] = n > 9            # NULL keys sort last (ascending, nulls last)
        return [pa.array(np.arange(n, dtype=np.int32)), pa.array(k, mask=mask), pa.array(rng.integers(0, 1000, n).astype(np.int32))]
    l, r = side(nl), side(nr)
    want = rows(po.sort_merge_join(l, r, [(1, 1)], jt, False, False, nen))
    got = device_join(ctx, [l], [r], [(1, 1)], jt, nen)
    if jt == "Full":                     # multiset contract (see the oracle's note): sorted rows
        got, want = sorted(got, key=skey), sorted(want, key=skey)
    assert got == want


@pytest.mark.gpu
def test_device_sort_merge_join_refuses_what_it_does_not_cover(ctx):
    import dfgpu
    l = cols([[1], [1], [1]])
    for jt in ("RightSemi",):
        with pytest.raises(dfgpu.DfgpuError) as e:
            device_join(ctx, [l], [l], [(1, 1)], jt)
        assert e.value.kind == "NotImplemented"


# ------------------------------------------------------------------ JoinFilter: the reference's sort_merge_join.slt vectors (rows compared sorted: the file says rowsort)
SMJ_FILTER = load_golden("unit_vectors.json")["sort_merge_join_filter"]
PA_TYPES = {"utf8": pa.string(), "int64": pa.int64()}
fkey = lambda r: [(0, "") if v is None else (1, str(v)) for v in r]


def table_cols(t):
    return [pa.array(v, type=PA_TYPES[ty]) for v, ty in zip(t["columns"], t["types"])]


def filter_fn(tree):
    """the vector's filter tree as a Python function over (left_row, right_row); NULL in -> NULL out"""
    import operator
    ops_ = {"<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge, "!=": operator.ne, "=": operator.eq, "*": operator.mul, "+": operator.add}

    def ev(t, l, r):
        if isinstance(t, list) and t and t[0] in ("l", "r"):
            return (l if t[0] == "l" else r)[t[1]]
        if isinstance(t, list):
            a, b = ev(t[1], l, r), ev(t[2], l, r)
            return None if a is None or b is None else ops_[t[0]](a, b)
        return t
    return lambda l, r: ev(tree, l, r)


def project(rws, case):
    return [[r[i] for i in case["project"]] for r in rws] if "project" in case else rws


@pytest.mark.parametrize("case", SMJ_FILTER, ids=[c["name"] for c in SMJ_FILTER])
def test_oracle_sort_merge_join_filter_reference_cases(case):
    from oracle import pyoracle as po
    got = po.sort_merge_join(table_cols(case["left"]), table_cols(case["right"]), [tuple(x) for x in case["on"]], case["join_type"], filter=filter_fn(case["filter"]))
    assert sorted(project(rows(got), case), key=fkey) == sorted(case["expected"], key=fkey)


def device_filter(ops, tree, left_types, right_types):
    """the filter tree as a JoinFilter: intermediate column i = the i-th distinct (side, index) the tree mentions"""
    from dfgpu import capi
    cols_, fields = [], []
    CT = {"utf8": capi.UTF8, "int64": capi.INT64}

    def ex(t):
        if isinstance(t, list) and t and t[0] in ("l", "r"):
            key = ("left" if t[0] == "l" else "right", t[1])
            if key not in cols_:
                cols_.append(key); fields.append(ops.Field(f"c{len(cols_)}", CT[(left_types if t[0] == "l" else right_types)[t[1]]]))
            return ops.Column(f"c{cols_.index(key) + 1}", cols_.index(key))
        if isinstance(t, list):
            return ops.BinaryExpr(ex(t[1]), t[0], ex(t[2]))
        return ops.Literal(t, pa.int64())
    e = ex(tree)
    return ops.JoinFilter(e, cols_, ops.Schema(fields))


@pytest.mark.gpu
@pytest.mark.parametrize("case", SMJ_FILTER, ids=[c["name"] for c in SMJ_FILTER])
def test_device_sort_merge_join_filter_reference_cases(ctx, case):
    from dfgpu import physical_plan as ops
    lc, rc = table_cols(case["left"]), table_cols(case["right"])
    ln, rn = [f"l{i}" for i in range(len(lc))], [f"r{i}" for i in range(len(rc))]
    mem = lambda c, names: (lambda b: ops.MemoryExec([[b]], b.schema))(ops.batch_from_arrow(ctx, pa.table(dict(zip(names, c)))))
    filt = device_filter(ops, case["filter"], case["left"]["types"], case["right"]["types"])
    plan = ops.SortMergeJoinExec(mem(lc, ln), mem(rc, rn), [(ops.Column(ln[l], l), ops.Column(rn[r], r)) for l, r in case["on"]], case["join_type"], False, filt)
    out = list(plan.execute(0, ops.TaskContext(ctx, 8192)))
    got = []
    for b in out:
        got += rows([c.to_arrow() for c in b.materialize().columns])
    assert sorted(project(got, case), key=fkey) == sorted(case["expected"], key=fkey)


@pytest.mark.gpu
@pytest.mark.parametrize("jt", ["Inner", "Left", "Right", "Full"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_device_sort_merge_join_filter_equals_oracle(ctx, jt, seed):
    """random sorted inputs with duplicate and NULL keys, a filter over both sides whose result is NULL for some pairs (a nullable payload column)"""
    from oracle import pyoracle as po
    from dfgpu import capi, physical_plan as ops
    rng = np.random.default_rng(seed)
    nl, nr = 400, 300

    def side(n):
        k = np.sort(rng.integers(0, 40, n)); v = rng.integers(0, 100, n)
        return [pa.array(k, type=pa.int64(), mask=rng.random(n) < 0.05), pa.array(v, type=pa.int64(), mask=rng.random(n) < 0.1)]
    lc, rc = side(nl), side(nr)
    # NULL keys sort first (SortOptions::default): move them to the front, keeping the rest in order
    def nulls_first(c):
        idx = np.argsort(~np.asarray(c[0].is_null()), kind="stable")
        return [x.take(pa.array(idx)) for x in c]
    lc, rc = nulls_first(lc), nulls_first(rc)
    want = po.sort_merge_join(lc, rc, [(0, 0)], jt, filter=lambda l, r: None if l[1] is None or r[1] is None else l[1] > r[1])
    mem = lambda c, names: (lambda b: ops.MemoryExec([[b]], b.schema))(ops.batch_from_arrow(ctx, pa.table(dict(zip(names, c)))))
    filt = ops.JoinFilter(ops.BinaryExpr(ops.Column("x", 0), ">", ops.Column("y", 1)), [("left", 1), ("right", 1)], ops.Schema([ops.Field("x", capi.INT64), ops.Field("y", capi.INT64)]))
    plan = ops.SortMergeJoinExec(mem(lc, ["k", "v"]), mem(rc, ["k2", "v2"]), [(ops.Column("k", 0), ops.Column("k2", 0))], jt, False, filt)
    got = []
    for b in plan.execute(0, ops.TaskContext(ctx, 8192)):
        got += rows([c.to_arrow() for c in b.materialize().columns])
    assert sorted(got, key=fkey) == sorted(rows(want), key=fkey)


@pytest.mark.gpu
def test_device_sort_merge_join_semi_anti_with_a_filter_say_not_implemented(ctx):
    import dfgpu
    from dfgpu import capi, physical_plan as ops
    l = [pa.array([1, 2], type=pa.int64()), pa.array([1, 2], type=pa.int64())]
    mem = lambda c, names: (lambda b: ops.MemoryExec([[b]], b.schema))(ops.batch_from_arrow(ctx, pa.table(dict(zip(names, c)))))
    filt = ops.JoinFilter(ops.BinaryExpr(ops.Column("x", 0), ">", ops.Column("y", 1)), [("left", 1), ("right", 1)], ops.Schema([ops.Field("x", capi.INT64), ops.Field("y", capi.INT64)]))
    for jt in ("LeftSemi", "LeftAnti", "RightAnti"):
        with pytest.raises(dfgpu.DfgpuError) as e:
            list(ops.SortMergeJoinExec(mem(l, ["a", "b"]), mem(l, ["c", "d"]), [(ops.Column("a", 0), ops.Column("c", 0))], jt, False, filt).execute(0, ops.TaskContext(ctx, 8192)))
        assert e.value.kind == "NotImplemented"
