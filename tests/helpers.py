"""Shared helpers for the parity tests: golden-case loading, row materialisation, canonical ordering."""
import json
import os

import numpy as np
import pyarrow as pa

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def pa_types_for(case, ncols):
    ts = case.get("dtype", "int32").split(",")
    m = {"int32": pa.int32(), "int64": pa.int64(), "date32": pa.date32()}
    ts = [m[t] for t in ts]
    return ts + [ts[-1]] * (ncols - len(ts)) if len(ts) < ncols else ts


def side_batches(case, side):
    """-> list of batches, each a list of pyarrow arrays (one per column)"""
    s = case[side]
    ts = pa_types_for(case, len(s["names"]))
    return [[pa.array(col, type=t) for col, t in zip(b, ts)] for b in s["batches"]]


def rows_of(columns):
    """list of pyarrow/numpy columns -> list of python row lists (None for NULL, dates as ints)"""
    cols = []
    for c in columns:
        if isinstance(c, pa.ChunkedArray):
            c = c.combine_chunks()
        if isinstance(c, pa.Array):
            if pa.types.is_date32(c.type):
                c = c.cast(pa.int32())
            cols.append(c.to_pylist())
        else:
            cols.append(list(c))
    return [list(r) for r in zip(*cols)] if cols else []


def sort_rows(rows):
    key = lambda r: tuple((0, 0) if v is None else (1, v) for v in r)
    return sorted(rows, key=key)


def eval_filter_spec(spec, build_cols, probe_cols, build_idx, probe_idx):
    """JoinFilter of the golden cases: one comparison over the intermediate batch. Returns keep mask (uint8)."""
    inter = []
    for side, index in spec["column_indices"]:
        src, idx = (build_cols[index], build_idx) if side == "left" else (probe_cols[index], probe_idx)
        inter.append(np.asarray(src)[idx])
    lhs = inter[0]
    rhs = inter[spec["rhs_column"]] if "rhs_column" in spec else spec["rhs_literal"]
    op = spec["op"]
    res = {"!=": lhs != rhs, ">": lhs > rhs, "<": lhs < rhs, "=": lhs == rhs}[op]
    return res.astype(np.uint8)
