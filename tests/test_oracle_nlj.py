"""CPU: pins the oracle's NestedLoopJoinExec restatement (oracle/pyoracle.py nested_loop_join) on the eight reference tests of
physical-plan/src/joins/nested_loop_join.rs:892-1130 (tests/golden/unit_vectors.json, rows compared sorted as assert_batches_sorted_eq does)."""
import pytest

from nlj_common import NLJ, golden_filter_oracle, golden_tables, rows, sort_key, split


@pytest.mark.parametrize("case", NLJ["cases"], ids=[c["name"] for c in NLJ["cases"]])
@pytest.mark.parametrize("parts", [1, 3])
def test_oracle_nested_loop_join_reference_cases(case, parts):
    from oracle import pyoracle as po
    l, r = golden_tables()
    jt = case["join_type"]
    build_left = jt in ("Right", "RightSemi", "RightAnti", "Full")
    lb, rb = ([l], split(r, parts)) if build_left else (split(l, parts), [r])          # the streamed side arrives in `parts` batches
    fc = [tuple(x) for x in NLJ["filter"]["column_indices"]]
    got = sorted(rows(po.nested_loop_join(lb, rb, fc, golden_filter_oracle, jt)), key=sort_key)
    want = sorted((tuple(x) for x in case["expected_sorted"]), key=sort_key)
    assert got == want
