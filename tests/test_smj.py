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
