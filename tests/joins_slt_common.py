"""Shared by the CPU (oracle) and GPU (device) runs of tests/golden/joins_slt.json: table construction and the evaluation of one case."""
import json
import os

import numpy as np
import pyarrow as pa

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "joins_slt.json")))
PA = {"int32": pa.int32(), "uint32": pa.uint32(), "utf8": pa.utf8()}


def table(name):
    t = GOLDEN["tables"][name]
    cols = list(zip(*t["rows"]))
    return pa.table({n: pa.array(list(c), type=PA[ty]) for (n, ty), c in zip(t["columns"], cols)})


def null_last_key(v):
    return (1, 0) if v is None else (0, v)


def finish(rows, case):
    """WHERE on a join output column, projection, ORDER BY .. ASC NULLS LAST (stable) -- the trivial tail of every case, in Python."""
    if "where" in case:
        ci, pred = case["where"]
        rows = [r for r in rows if (r[ci] is None) == (pred == "is_null")]
    rows = [[r[i] for i in case["project"]] for r in rows]
    pos = case["project"].index(case["order_by"]) if case["order_by"] in case["project"] else None
    assert pos is not None
    return sorted(rows, key=lambda r: null_last_key(r[pos]))


def oracle_rows(case):
    """The join through the C oracle (oracle/dfo_join.c), rows assembled from its (build, probe) index pairs."""
    from oracle import pyoracle as po
    lt, rt = table(case["left"]), table(case["right"])
    lk = [lt.column(l) for l, _ in case["on"]]
    rk = [rt.column(r) for _, r in case["on"]]
    fn = None
    if "filter" in case:
        f = case["filter"]
        def fn(pb, bi, pi, f=f):
            vals = []
            for side, ci in f["columns"]:
                col, idx = (lt.column(ci), bi) if side == "left" else (rt.column(ci), pi)
                vals.append([col[int(i)].as_py() for i in idx])
            rhs = vals[1] if len(vals) > 1 else [f["literal"]] * len(vals[0])
            op = {">": lambda a, b: a > b, "!=": lambda a, b: a != b}[f["op"]]
            return np.array([0 if (a is None or b is None) else int(op(a, b)) for a, b in zip(vals[0], rhs)], dtype=np.uint8)      # NULL comparison = not kept
    res = po.hash_join([lk], [rk], case["join_type"], batch_size=8192, filter_fn=fn)
    jt = case["join_type"]
    rows = []
    for bi, pi in zip(res.build_idx, res.probe_idx):
        l = [None] * lt.num_columns if bi < 0 else [lt.column(c)[int(bi)].as_py() for c in range(lt.num_columns)]
        r = [None] * rt.num_columns if pi < 0 else [rt.column(c)[int(pi)].as_py() for c in range(rt.num_columns)]
        rows.append(l if jt in ("LeftSemi", "LeftAnti") else r if jt in ("RightSemi", "RightAnti") else l + r)
    return finish(rows, case)
