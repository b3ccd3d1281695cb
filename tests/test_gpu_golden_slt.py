"""-m gpu: the device path against the same reference known answers that pin the oracle in test_oracle_golden_slt.py
(decimal.slt, aggregate unit tests, AggregateExec Partial -> Final, SortExec floats), through the C ABI."""
import pytest

from golden_engine import run_clickbench_case, DeviceEngine, run_grouped_case, run_scalar_case, run_slt_case, run_sort_case
from helpers import load_golden

pytestmark = pytest.mark.gpu
SLT = load_golden("decimal_slt.json")
AGG = load_golden("aggregates.json")


@pytest.fixture()
def eng(ctx):
    return DeviceEngine(ctx)


@pytest.mark.parametrize("case", SLT["cases"], ids=[c["name"] for c in SLT["cases"]])
def test_device_decimal_slt(eng, case):
    run_slt_case(eng, SLT, case)


@pytest.mark.parametrize("case", AGG["scalar"], ids=[c["name"] for c in AGG["scalar"]])
def test_device_aggregate_unit_tests(eng, case):
    run_scalar_case(eng, case)


@pytest.mark.parametrize("case", AGG["grouped"], ids=[c["name"] for c in AGG["grouped"]])
def test_device_aggregate_exec_partial_final(eng, case):
    run_grouped_case(eng, AGG, case)


@pytest.mark.parametrize("case", AGG["sort"], ids=[c["name"] for c in AGG["sort"]])
def test_device_sort_exec_known_answers(eng, case):
    run_sort_case(eng, case)


@pytest.mark.parametrize("case", AGG["clickbench"]["cases"], ids=[c["name"] for c in AGG["clickbench"]["cases"]])
def test_device_clickbench_sample(eng, case):
    run_clickbench_case(eng, AGG, case)
