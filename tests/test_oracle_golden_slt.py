"""Pins the CPU oracle against known answers the reference's own tests hold beyond HashJoinExec (test_oracle_golden.py):
decimal.slt over decimal_data.csv (arithmetic result types and values, filters, MIN/MAX/SUM/AVG types and values, ORDER BY,
GROUP BY COUNT), the Sum / Avg / Count / Min / Max unit tests, AggregateExec Partial -> Final, SortExec float ordering.
Fixtures: tests/golden/decimal_slt.json, tests/golden/aggregates.json (hand-transcribed; scripts beside them)."""
import pytest

from golden_engine import run_clickbench_case, OracleEngine, run_grouped_case, run_scalar_case, run_slt_case, run_sort_case
from helpers import load_golden

SLT = load_golden("decimal_slt.json")
AGG = load_golden("aggregates.json")


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
