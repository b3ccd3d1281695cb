"""-m gpu: the device path against the same reference known answers that pin the oracle in test_oracle_golden_slt.py
(decimal.slt, aggregate unit tests, AggregateExec Partial -> Final, SortExec floats), through the C ABI."""
import pytest

from golden_engine import run_binary_vector, run_in_list_vector, run_order_case, run_table_case, run_clickbench_case, DeviceEngine, run_grouped_case, run_scalar_case, run_slt_case, run_sort_case
from helpers import load_golden

pytestmark = pytest.mark.gpu
SLT = load_golden("decimal_slt.json")
AGG = load_golden("aggregates.json")
UNIT = load_golden("unit_vectors.json")
GBO = load_golden("groupby_order_slt.json")


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


def _joins_slt_cases():
    from joins_slt_common import GOLDEN
    return GOLDEN["cases"]


@pytest.mark.parametrize("mode", ["CollectLeft", "Partitioned"])
@pytest.mark.parametrize("case", _joins_slt_cases(), ids=lambda c: c["name"])
def test_device_joins_slt(ctx, case, mode):
    """The same joins.slt cases through the plan layer: HashJoinExec (with the JoinFilter where the query has one) -> FilterExec
    IS [NOT] NULL -> ProjectionExec -> SortExec ASC NULLS LAST, rows equal to the reference's expected output."""
    import pyarrow as pa
    import dfgpu
    from dfgpu import physical_plan as ops
    from joins_slt_common import GOLDEN, table
    lt, rt = table(case["left"]), table(case["right"])
    mk = lambda t: ops.MemoryExec([[ops.batch_from_arrow(ctx, t)]], ops.batch_from_arrow(ctx, t).schema)
    C = ops.Column
    on = [(C(lt.column_names[l], l), C(rt.column_names[r], r)) for l, r in case["on"]]
    filt = None
    if "filter" in case:
        f = case["filter"]
        tcode = {pa.int32(): dfgpu.capi.INT32, pa.uint32(): dfgpu.capi.UINT32, pa.utf8(): dfgpu.capi.UTF8}
        types = [(lt if side == "left" else rt).schema.field(ci).type for side, ci in f["columns"]]
        rhs = C("y", 1) if len(f["columns"]) > 1 else ops.Literal(f["literal"], types[0])
        filt = ops.JoinFilter(ops.BinaryExpr(C("x", 0), f["op"], rhs), [tuple(c) for c in f["columns"]],
                              ops.Schema([ops.Field(n, tcode[t]) for n, t in zip("xy", types)]))
    plan = ops.HashJoinExec(mk(lt), mk(rt), on, filt, case["join_type"], mode)
    names = plan.schema().names()
    if "where" in case:
        ci, pred = case["where"]
        plan = ops.FilterExec(ops.IsNullExpr(C(names[ci], ci), negated=(pred == "is_not_null")), plan)
    plan = ops.ProjectionExec([(C(names[i], i), names[i]) for i in case["project"]], plan)
    pos = case["project"].index(case["order_by"])
    plan = ops.SortExec([ops.PhysicalSortExpr(C("k", pos), False, False)], plan)
    got = []
    for b in ops.collect(plan, ops.TaskContext(ctx, 2)):          # batch_size 2, as the .slt file sets it
        got += [list(r) for r in zip(*[c.to_arrow().to_pylist() for c in b.columns])]
    assert got == case["expected"]


@pytest.mark.parametrize("case", UNIT["binary"], ids=[c["name"] for c in UNIT["binary"]])
def test_device_binary_rs_unit_vectors(eng, case):
    """expressions/binary.rs unit tests on the device; dictionary operands go in as dictionary arrays"""
    run_binary_vector(eng, case)


@pytest.mark.parametrize("case", UNIT["sort"], ids=[c["name"] for c in UNIT["sort"]])
def test_device_sort_rs_unit_vectors(ctx, case):
    """sorts/sort.rs test_in_mem_sort / test_sort_metadata through SortExec over CoalescePartitionsExec: one batch, every row, ascending"""
    import pyarrow as pa
    from dfgpu import physical_plan as ops
    from golden_engine import pa_type
    parts = [[ops.batch_from_arrow(ctx, pa.table({"i": pa.array(p, type=pa_type(case["type"]))}))] for p in case["partitions"]]
    src = ops.MemoryExec(parts, parts[0][0].schema)
    plan = ops.SortExec([ops.PhysicalSortExpr(ops.Column("i", 0), case["descending"], case["nulls_first"])], ops.CoalescePartitionsExec(src))
    out = list(plan.execute(0, ops.TaskContext(ctx, 8192)))
    assert len(out) == case.get("expected_batches", 1)
    vals = [v for b in out for v in b.to_arrow()["i"].to_pylist()]
    assert len(vals) == case.get("expected_rows", len(vals)) and vals == sorted(v for p in case["partitions"] for v in p)
    if "expected" in case:
        assert vals == case["expected"]


@pytest.mark.parametrize("case", UNIT["repartition"]["cases"], ids=[c["name"] for c in UNIT["repartition"]["cases"]])
def test_device_repartition_rs_unit_vectors(ctx, case):
    """repartition/mod.rs: batches per output partition under RoundRobinBatch (every input partition's rotation starts at output 0), row conservation under Hash"""
    import pyarrow as pa
    from dfgpu import physical_plan as ops
    fix = UNIT["repartition"]["batch"]
    batch = ops.batch_from_arrow(ctx, pa.table({fix["column"]: pa.array(fix["values"], type=pa.uint32())}))
    src = ops.MemoryExec([[batch] * n for n in case["inputs"]], batch.schema)
    part = ops.Partitioning.RoundRobinBatch(case["n"]) if case["scheme"] == "RoundRobinBatch" else ops.Partitioning.Hash([ops.Column(fix["column"], 0)], case["n"])
    plan = ops.RepartitionExec(src, part)
    tc = ops.TaskContext(ctx, 8192)
    outs = [list(plan.execute(p, tc)) for p in range(case["n"])]
    if "expected_batches" in case:
        assert [len(o) for o in outs] == case["expected_batches"]
        assert all(b.num_rows == len(fix["values"]) for o in outs for b in o)
    else:
        rows = [v for o in outs for b in o for v in b.to_arrow()[fix["column"]].to_pylist()]
        assert len(rows) == case["expected_total_rows"] and sorted(set(rows)) == fix["values"]
        for o in outs:                      # one value lands in one partition only
            for other in outs:
                if o is not other:
                    assert not ({v for b in o for v in b.to_arrow()[fix["column"]].to_pylist()} & {v for b in other for v in b.to_arrow()[fix["column"]].to_pylist()})


@pytest.mark.parametrize("case", GBO["cases"], ids=[c["name"] for c in GBO["cases"]])
def test_device_group_by_and_aggregate_slt_values_cases(eng, case):
    """group_by.slt GROUP BY ALL (NULL key group) and the dictionary-key tables, aggregate.slt test_decimal_table"""
    run_table_case(eng, GBO, case)


@pytest.mark.parametrize("case", GBO["order"]["cases"], ids=[c["name"] for c in GBO["order"]["cases"]])
def test_device_order_slt_null_placement(eng, case):
    run_order_case(eng, GBO, case)


@pytest.mark.parametrize("case", UNIT["in_list"], ids=[c["name"] for c in UNIT["in_list"]])
def test_device_in_list_rs_unit_vectors(eng, case):
    run_in_list_vector(eng, case)
