"""Pins the CPU oracle (oracle/dfo_join.c) against the reference's own HashJoinExec known-answer tests
(tests/golden/hash_join.json, transcribed from joins/hash_join.rs mod tests): rows, row ORDER where the
reference asserts it (assert_batches_eq) and the number of output batches for every batch_size."""
import numpy as np
import pyarrow as pa
import pytest

from helpers import eval_filter_spec, load_golden, rows_of, side_batches, sort_rows
from oracle import pyoracle as po

CASES = load_golden("hash_join.json")["cases"]


def oracle_join_rows(case, batch_size, force_collisions=False):
    lb, rb = side_batches(case, "left"), side_batches(case, "right")
    ln, rn = case["left"]["names"], case["right"]["names"]
    lk = [[b[ln.index(l)] for l, _ in case["on"]] for b in lb]
    rk = [[b[rn.index(r)] for _, r in case["on"]] for b in rb]
    # reference-order concatenation of the build side: reversed input batches (hash_join.rs:746,764)
    bcat = [pa.concat_arrays([b[i] for b in lb[::-1]]) for i in range(len(ln))]
    bnp = [np.asarray(c.cast(pa.int32()) if pa.types.is_date32(c.type) else c) for c in bcat]
    pnp = [[np.asarray(c.cast(pa.int32()) if pa.types.is_date32(c.type) else c) for c in b] for b in rb]
    filt = None
    if case["filter"]:
        filt = lambda pb, bi, pi: eval_filter_spec(case["filter"], bnp, pnp[pb], bi, pi)
    res = po.hash_join(lk, rk, case["join_type"], case["null_equals_null"], batch_size, force_collisions, filt)
    jt = case["join_type"]
    rows = []
    for b, p, pb in zip(res.build_idx, res.probe_idx, res.probe_batch):
        left = [None] * len(ln) if b < 0 else [int(c[b]) for c in bnp]
        right = [None] * len(rn) if p < 0 else [int(c[p]) for c in pnp[pb]]
        rows.append(left if jt in ("LeftSemi", "LeftAnti") else right if jt in ("RightSemi", "RightAnti") else left + right)
    return rows, res


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_golden(case):
    for bs in case["batch_sizes"]:
        rows, res = oracle_join_rows(case, bs)
        if case["ordered"]:
            assert rows == case["expected"], f"batch_size={bs}"
        else:
            assert sort_rows(rows) == sort_rows(case["expected"]), f"batch_size={bs}"
        if case["batch_count"]:
            # the reference counts emitted RecordBatches; an empty right side emits none for the probe phase
            assert len(res.batch_offsets) - 1 == case["batch_count"][str(bs)], f"batch_size={bs}"


@pytest.mark.parametrize("case", [c for c in CASES if not c["name"].startswith("join_splitted")], ids=lambda c: c["name"])
def test_oracle_independent_of_hash_values(case):
    """cargo feature force_hash_collisions (common/src/hash_utils.rs:306-318, CI rust.yml:454-470): all hashes 0."""
    rows, _ = oracle_join_rows(case, 8192, force_collisions=True)
    assert sort_rows(rows) == sort_rows(case["expected"])
