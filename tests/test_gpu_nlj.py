"""-m gpu: NestedLoopJoinExec (csrc/exec/exec.cpp over dfgpu_cross_join_indices) against the reference's eight known answers and against the oracle
restatement on random inputs -- batch by batch, rows in the reference's ORDER (left-major pairs, then the batch's unmatched rows), NULL filter results
dropping the pair, empty sides, a cross join without filter, several streamed batches / partitions."""
import numpy as np
import pyarrow as pa
import pytest

from nlj_common import JOIN_TYPES, NLJ, golden_tables, random_tables, rows, sort_key, split

pytestmark = pytest.mark.gpu


def plan_for(ctx, lparts, rparts, lnames, rnames, jt, filt):
    from dfgpu import physical_plan as ops
    tab = lambda cols, names: ops.batch_from_arrow(ctx, pa.table(dict(zip(names, cols))))
    mem = lambda parts, names: ops.MemoryExec([[tab(b, names) for b in p] for p in parts], tab(parts[0][0], names).schema)
    return ops.NestedLoopJoinExec(mem(lparts, lnames), mem(rparts, rnames), filt, jt)


def run(ctx, plan):
    from dfgpu import physical_plan as ops
    tc = ops.TaskContext(ctx, 8192)
    out = []
    for p in range(plan.output_partitioning().partition_count()):
        out.append([[c.to_arrow() for c in b.columns] for b in plan.execute(p, tc)])
    return out


@pytest.mark.parametrize("case", NLJ["cases"], ids=[c["name"] for c in NLJ["cases"]])
@pytest.mark.parametrize("parts", [1, 3])
def test_device_nested_loop_join_reference_cases(ctx, case, parts):
    from dfgpu import physical_plan as ops
    C, B, L = ops.Column, ops.BinaryExpr, ops.Literal
    l, r = golden_tables()
    jt = case["join_type"]
    build_left = jt in ("Right", "RightSemi", "RightAnti", "Full")
    filt = ops.JoinFilter(B(B(C("x", 0), "!=", L(8, pa.int32())), "AND", B(C("x", 1), "!=", L(10, pa.int32()))), [("left", 1), ("right", 1)], None)
    lparts, rparts = ([[l]], [[b] for b in split(r, parts)]) if build_left else ([[b] for b in split(l, parts)], [[r]])
    if jt == "Full":
        rparts = [[b for p in rparts for b in p]]             # Full: single partition on both sides, the batches stream one after the other
    plan = plan_for(ctx, lparts, rparts, list(NLJ["left"]["columns"]), list(NLJ["right"]["columns"]), jt, filt)
    got = sorted((x for part in run(ctx, plan) for x in rows(part)), key=sort_key)
    assert got == sorted((tuple(x) for x in case["expected_sorted"]), key=sort_key)


@pytest.mark.parametrize("jt", JOIN_TYPES)
@pytest.mark.parametrize("shape", [(37, 23, 3), (1, 50, 1), (64, 1, 2), (0, 9, 1), (9, 0, 1), (300, 200, 4)], ids=lambda s: f"{s[0]}x{s[1]}-{s[2]}batches")
def test_device_nested_loop_join_equals_oracle_batch_by_batch(ctx, jt, shape):
    """filter: left.k < right.k (NULL keys drop the pair).  Every output batch equals the oracle's, row for row."""
    from dfgpu import physical_plan as ops
    from oracle import pyoracle as po
    nl, nr, parts = shape
    l, r = random_tables(nl * 1000 + nr, max(nl, 0), max(nr, 0))
    build_left = jt in ("Right", "RightSemi", "RightAnti", "Full")
    lb, rb = ([l], split(r, parts)) if build_left else (split(l, parts), [r])
    want = po.nested_loop_join(lb, rb, [("left", 0), ("right", 0)], lambda inter: po.binary("<", inter[0], inter[1]), jt)
    C, B = ops.Column, ops.BinaryExpr
    filt = ops.JoinFilter(B(C("x", 0), "<", C("x", 1)), [("left", 0), ("right", 0)], None)
    plan = plan_for(ctx, [lb], [rb], ["lk", "lv"], ["rk", "rv"], jt, filt)
    got = run(ctx, plan)[0]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert rows([g]) == rows([w])


def test_cross_join_without_filter_and_large_pair_count(ctx):
    """no JoinFilter: every pair, left-major; 3000 x 30000 = 90 M candidate pairs are generated in runs of left rows and filtered down by left.v = right.v"""
    from dfgpu import physical_plan as ops
    l = [pa.array(np.arange(5, dtype=np.int32)), pa.array(np.arange(5, dtype=np.int64) * 10)]
    r = [pa.array(np.arange(3, dtype=np.int32)), pa.array(np.arange(3, dtype=np.int64) * 7)]
    got = rows(run(ctx, plan_for(ctx, [[l]], [[r]], ["lk", "lv"], ["rk", "rv"], "Inner", None))[0])
    assert got == [(i, i * 10, j, j * 7) for i in range(5) for j in range(3)]
    rng = np.random.default_rng(3)
    big_l = [pa.array(np.arange(3000, dtype=np.int32)), pa.array(rng.integers(0, 10**6, 3000).astype(np.int64))]
    big_r = [pa.array(np.arange(30000, dtype=np.int32)), pa.array(rng.integers(0, 10**6, 30000).astype(np.int64))]
    C, B = ops.Column, ops.BinaryExpr
    filt = ops.JoinFilter(B(C("x", 0), "=", C("x", 1)), [("left", 1), ("right", 1)], None)
    got = rows(run(ctx, plan_for(ctx, [[big_l]], [[big_r]], ["lk", "lv"], ["rk", "rv"], "Inner", filt))[0])
    lv, rv = big_l[1].to_numpy(), big_r[1].to_numpy()
    pos = {}
    for j, v in enumerate(rv.tolist()):
        pos.setdefault(v, []).append(j)
    want = [(i, int(lv[i]), j, int(rv[j])) for i in range(3000) for j in pos.get(int(lv[i]), [])]
    assert got == want
