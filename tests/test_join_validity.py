"""check_join_is_valid (joins/utils.rs:387-430): the reference's own unit cases (joins/utils.rs:1524-1618), on the host-side mirror of HashJoinExec::try_new."""
import pytest

import dfgpu
from dfgpu import physical_plan as pp
from dfgpu.operators import Field, Schema


def schema(cols):
    return Schema([Field(n, 5) for n, _ in sorted(cols, key=lambda c: c[1])])


def check(left, right, on):
    pp.check_join_is_valid(schema(left), schema(right), on)


C = pp.Column


def test_check_valid():                    # utils.rs:1525
    check([("a", 0), ("b1", 1)], [("a", 0), ("b2", 1)], [(C("a", 0), C("a", 0))])


def test_check_not_in_right():             # :1538
    with pytest.raises(dfgpu.DfgpuError):
        check([("a", 0), ("b", 1)], [("b", 0)], [(C("a", 0), C("a", 0))])


def test_check_not_in_left():              # :1583
    with pytest.raises(dfgpu.DfgpuError):
        check([("b", 0)], [("a", 0)], [(C("a", 0), C("a", 0))])


def test_check_collision():                # :1595: column checks are per side, a name on both sides is fine
    check([("a", 0), ("c", 1)], [("a", 0), ("b", 1)], [(C("a", 0), C("b", 1))])


def test_check_in_right():                 # :1608
    check([("a", 0), ("c", 1)], [("b", 0)], [(C("a", 0), C("b", 0))])


def test_columns_inside_expressions_and_wrong_index():
    on = [(pp.BinaryExpr(C("a", 0), "+", C("c", 1)), pp.CastExpr(C("b", 0), 5))]
    check([("a", 0), ("c", 1)], [("b", 0)], on)
    with pytest.raises(dfgpu.DfgpuError) as e:
        check([("a", 0), ("c", 1)], [("b", 0)], [(C("c", 0), C("b", 0))])          # right name, wrong index: Column equality is (name, index)
    assert "Missing on the left" in str(e.value) and 'name: "c", index: 0' in str(e.value)


def test_hash_join_exec_refuses_an_invalid_on_at_construction():
    sch = Schema([Field("a", 5), Field("b", 5)])
    left, right = pp.MemoryExec([[]], sch), pp.MemoryExec([[]], Schema([Field("x", 5)]))
    pp.HashJoinExec(left, right, [(C("a", 0), C("x", 0))], None, "Inner")
    with pytest.raises(dfgpu.DfgpuError):
        pp.HashJoinExec(left, right, [(C("a", 0), C("a", 0))], None, "Inner")
