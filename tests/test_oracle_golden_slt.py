"""Pins the CPU oracle against known answers the reference's own tests hold beyond HashJoinExec (test_oracle_golden.py):
decimal.slt over decimal_data.csv (arithmetic result types and values, filters, MIN/MAX/SUM/AVG types and values, ORDER BY,
GROUP BY COUNT), the Sum / Avg / Count / Min / Max unit tests, AggregateExec Partial -> Final, SortExec float ordering.
Fixtures: tests/golden/decimal_slt.json, tests/golden/aggregates.json (hand-transcribed; scripts beside them)."""
import pytest

from golden_engine import run_binary_vector, run_in_list_vector, run_order_case, run_table_case, run_clickbench_case, OracleEngine, run_grouped_case, run_scalar_case, run_slt_case, run_sort_case
from helpers import load_golden

SLT = load_golden("decimal_slt.json")
AGG = load_golden("aggregates.json")
UNIT = load_golden("unit_vectors.json")
GBO = load_golden("groupby_order_slt.json")


@pytest.fixture(scope="module")
def eng():
    return OracleEngine()


@pytest.mark.parametrize("case", SLT["cases"], ids=[c["name"] for c in SLT["cases"]])
def test_oracle_decimal_slt(eng, case):
    run_slt_case(eng, SLT, case)


@pytest.mark.parametrize("case", AGG["scalar"], ids=[c["name"] for c in AGG["scalar"]])
def test_oracle_aggregate_unit_tests(eng, case):
    run_scalar_case(eng, case)


@pytest.mark.parametrize("case", AGG["grouped"], ids=[c["name"] for c in AGG["grouped"]])
def test_oracle_aggregate_exec_partial_final(eng, case):
    run_grouped_case(eng, AGG, case)


@pytest.mark.parametrize("case", AGG["sort"], ids=[c["name"] for c in AGG["sort"]])
def test_oracle_sort_exec_known_answers(eng, case):
    run_sort_case(eng, case)


@pytest.mark.parametrize("case", AGG["clickbench"]["cases"], ids=[c["name"] for c in AGG["clickbench"]["cases"]])
def test_oracle_clickbench_sample(eng, case):
    run_clickbench_case(eng, AGG, case)


def test_oracle_grouping_sets_reference_vector():
    """aggregates/mod.rs:1353-1507 (`check_grouping_sets`): evaluate_group_by (:1161-1200) restated over the oracle's GroupValues /
    GroupsAccumulator -- every batch is interned once per grouping set (masked keys replaced by typed NULLs) into ONE group table and
    COUNT(1) updated with those ids -- gives the reference's Partial table (no-spill variant)."""
    import numpy as np
    import pyarrow as pa
    from oracle import pyoracle as po
    batches = [(pa.array([2, 3, 4, 4], type=pa.uint32()), pa.array([1.0, 2.0, 3.0, 4.0])), (pa.array([2, 3, 3, 4], type=pa.uint32()), pa.array([1.0, 2.0, 3.0, 4.0]))]
    sets = [[False, True], [True, False], [False, False]]
    groups, acc = po.Groups([pa.uint32(), pa.float64()]), po.Acc("COUNT", pa.int8())
    for a, b in batches:
        nulls = [pa.nulls(len(a), pa.uint32()), pa.nulls(len(b), pa.float64())]
        ones = pa.array(np.ones(len(a), dtype=np.int8))
        for mask in sets:
            gids = groups.intern([nulls[i] if m else c for i, (m, c) in enumerate(zip(mask, (a, b)))])
            acc.update_batch(ones, gids, None, len(groups))
    ka, kb = groups.emit()
    got = sorted(zip(ka.to_pylist(), kb.to_pylist(), acc.evaluate().to_pylist()), key=lambda r: tuple((0, 0) if v is None else (1, v) for v in r))
    want = [(None, 1.0, 2), (None, 2.0, 2), (None, 3.0, 2), (None, 4.0, 2), (2, None, 2), (2, 1.0, 2), (3, None, 3), (3, 2.0, 2), (3, 3.0, 1), (4, None, 3), (4, 3.0, 1), (4, 4.0, 2)]
    assert got == want


def _joins_slt_cases():
    from joins_slt_common import GOLDEN
    return GOLDEN["cases"]


@pytest.mark.parametrize("case", _joins_slt_cases(), ids=lambda c: c["name"])
def test_oracle_joins_slt(case):
    """sqllogictest/test_files/joins.slt equi-join cases (outer joins under IS [NOT] NULL filters, semi / anti joins with duplicate and
    NULL keys and join filters): the oracle's hash join reproduces the reference's expected rows."""
    from joins_slt_common import oracle_rows
    assert oracle_rows(case) == case["expected"]


@pytest.mark.parametrize("case", UNIT["binary"], ids=[c["name"] for c in UNIT["binary"]])
def test_oracle_binary_rs_unit_vectors(eng, case):
    """expressions/binary.rs unit tests: arithmetic over Int32 / dictionary / Decimal128 operands, Kleene logic, Boolean and Decimal128 comparisons"""
    run_binary_vector(eng, case)


@pytest.mark.parametrize("case", UNIT["sort"], ids=[c["name"] for c in UNIT["sort"]])
def test_oracle_sort_rs_unit_vectors(eng, case):
    import numpy as np
    import pyarrow as pa
    from golden_engine import make_array
    col = make_array(case["type"], [v for p in case["partitions"] for v in p])
    idx = eng.sort_indices([col], [case["descending"]], [case["nulls_first"]])
    out = eng.take(col, idx)
    assert len(out) == case.get("expected_rows", len(out))
    vals = out.to_pylist()
    assert vals == sorted(vals)
    if "expected" in case:
        assert vals == case["expected"]


def test_oracle_repartition_hash_conserves_rows():
    """repartition/mod.rs many_to_many_hash_partition: 3 x 50 batches of [1..8] hashed on c0 into 8 partitions keep every row"""
    import pyarrow as pa
    from oracle import pyoracle as po
    fix = UNIT["repartition"]; case = next(c for c in fix["cases"] if c["scheme"] == "Hash")
    batch = pa.array(fix["batch"]["values"], type=pa.uint32())
    total = 0
    for n_batches in case["inputs"]:
        for _ in range(n_batches):
            idx, counts = po.hash_partition([batch], case["n"])
            assert len(counts) == case["n"] and sum(counts) == len(batch) and sorted(idx) == list(range(len(batch)))
            total += sum(counts)
    assert total == case["expected_total_rows"]


@pytest.mark.parametrize("case", GBO["cases"], ids=[c["name"] for c in GBO["cases"]])
def test_oracle_group_by_and_aggregate_slt_values_cases(eng, case):
    """group_by.slt GROUP BY ALL (NULL key group) and the dictionary-key tables, aggregate.slt test_decimal_table"""
    run_table_case(eng, GBO, case)


@pytest.mark.parametrize("case", GBO["order"]["cases"], ids=[c["name"] for c in GBO["order"]["cases"]])
def test_oracle_order_slt_null_placement(eng, case):
    run_order_case(eng, GBO, case)


@pytest.mark.parametrize("case", UNIT["in_list"], ids=[c["name"] for c in UNIT["in_list"]])
def test_oracle_in_list_rs_unit_vectors(eng, case):
    run_in_list_vector(eng, case)
