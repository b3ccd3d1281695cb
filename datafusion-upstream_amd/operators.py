"""Plain data shared by the host layers: schema fields, TaskContext, operator / join-type code tables.

The operators themselves (FilterExec, ProjectionExec, HashJoinExec, AggregateExec, SortExec, RepartitionExec,
CoalesceBatchesExec, ...) live in C++ (csrc/exec/exec.cpp, C ABI include/dfgpu_exec.h); `physical_plan.py` holds the typed
Python builders over them.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

from . import capi
from .capi import DfgpuError
from .device import Array, Context


@dataclass(frozen=True)
class Field:
    name: str
    dtype: int
    precision: int = 0
    scale: int = 0
    nullable: bool = True


@dataclass
class Schema:
    fields: List[Field]

    def index_of(self, name: str) -> int:
        for i, f in enumerate(self.fields):
            if f.name == name:
                return i
        raise DfgpuError(1, f"Schema error: No field named {name}")

    def names(self) -> List[str]:
        return [f.name for f in self.fields]


@dataclass
class TaskContext:
    """≙ datafusion_execution::TaskContext (execution/src/task.rs:44-59)."""
    ctx: Context
    batch_size: int = 8192              # datafusion.execution.batch_size (common/src/config.rs:215)


def field_of_array(name: str, a: Array) -> Field:
    d = a.describe()
    if d.type == capi.DICTIONARY:
        dd = d.dictionary.contents
        return Field(name, dd.type, dd.precision, dd.scale)
    return Field(name, d.type, d.precision, d.scale)


# BinaryExpr operators (datafusion_expr::Operator) -> DFGPU_OP_*
_OPS = {"+": capi.OP_ADD, "-": capi.OP_SUB, "*": capi.OP_MUL, "/": capi.OP_DIV, "%": capi.OP_REM,
        "=": capi.OP_EQ, "!=": capi.OP_NEQ, "<": capi.OP_LT, "<=": capi.OP_LTEQ, ">": capi.OP_GT, ">=": capi.OP_GTEQ,
        "IS DISTINCT FROM": capi.OP_DISTINCT, "IS NOT DISTINCT FROM": capi.OP_NOT_DISTINCT, "AND": capi.OP_AND, "OR": capi.OP_OR}

# JoinType (datafusion/common/src/join_type.rs:30-47) -> DFGPU_JOIN_*
JOIN_TYPES = {"Inner": capi.JOIN_INNER, "Left": capi.JOIN_LEFT, "Right": capi.JOIN_RIGHT, "Full": capi.JOIN_FULL,
              "LeftSemi": capi.JOIN_LEFT_SEMI, "RightSemi": capi.JOIN_RIGHT_SEMI, "LeftAnti": capi.JOIN_LEFT_ANTI, "RightAnti": capi.JOIN_RIGHT_ANTI}
