"""Shared by the CPU and -m gpu NestedLoopJoinExec tests: the transcribed reference cases (unit_vectors.json) and random inputs."""
import numpy as np
import pyarrow as pa

from helpers import load_golden

NLJ = load_golden("unit_vectors.json")["nested_loop_join"]
JOIN_TYPES = ["Inner", "Left", "Right", "Full", "LeftSemi", "LeftAnti", "RightSemi", "RightAnti"]


def golden_tables():
    l = [pa.array(v, type=pa.int32()) for v in NLJ["left"]["columns"].values()]
    r = [pa.array(v, type=pa.int32()) for v in NLJ["right"]["columns"].values()]
    return l, r


def golden_filter_oracle(inter):
    from oracle import pyoracle as po
    a = po.binary("!=", inter[0], pa.array([8], type=pa.int32()), r_scalar=True)
    b = po.binary("!=", inter[1], pa.array([10], type=pa.int32()), r_scalar=True)
    return po.binary("AND", a, b)


def rows(batches):
    out = []
    for b in batches:
        cols = [c.to_pylist() for c in b]
        out += [tuple(c[i] for c in cols) for i in range(len(cols[0]))] if cols else []
    return out


def sort_key(r):
    return tuple((x is None, x if x is not None else 0) for x in r)


def random_tables(seed, nl, nr, null_frac=0.15):
    rng = np.random.default_rng(seed)
    mk = lambda n: [pa.array(rng.integers(0, 12, n).astype(np.int32), mask=rng.random(n) < null_frac), pa.array(rng.integers(0, 1000, n).astype(np.int64))]
    return mk(nl), mk(nr)


def split(cols, parts):
    n = len(cols[0]); cuts = [n * i // parts for i in range(parts + 1)]
    return [[c.slice(a, b - a) for c in cols] for a, b in zip(cuts, cuts[1:])]
