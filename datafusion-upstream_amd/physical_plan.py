"""Typed Python builders over the C++ host layer (csrc/exec/exec.cpp, include/dfgpu_exec.h).

Same class names and constructor arguments as the reference's operators (and as `operators.py`): a Python object only
describes a plan node; the node itself -- its `execute(partition, TaskContext)` stream, the per-batch call sequence into
the kernel ABI, late materialisation, selection-mask fusion -- is C++.  The C++ handles are built on first `execute`
(literals need the device context).  Nodes that exist only in Python (`exchange.ShuffleExec`, the multi-GPU shuffle over
torch.distributed) are run when they are reached and enter the C++ plan as a MemoryExec of the rows received.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Iterator, List, Optional, Sequence, Tuple

from . import capi
from .capi import DfgpuError
from .device import Array, Context
from .operators import Field, Schema, TaskContext, JOIN_TYPES, _OPS, field_of_array     # shared plain dataclasses / tables


def _lib():
    return capi.load_library()


def _check(st: int):
    if st != 0:
        msg = _lib().dfgpu_exec_last_error()
        raise DfgpuError(st, msg.decode() if msg else "")


def _strs(names: Sequence[str]):
    return (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])


def _ptrs(handles: Sequence):
    return (C.c_void_p * max(1, len(handles)))(*[h.value if isinstance(h, C.c_void_p) else h for h in handles])


class _Handle:
    _free = None

    def __init__(self, h: C.c_void_p):
        self.h = h

    def __del__(self):
        try:
            if self.h is not None and self.h.value:
                getattr(_lib(), self._free)(self.h)
                self.h = None
        except Exception:
            pass


class _BatchH(_Handle):
    _free = "dfgpu_batch_free"


class _ExprH(_Handle):
    _free = "dfgpu_expr_free"


class _PlanH(_Handle):
    _free = "dfgpu_plan_free"


class _StreamH(_Handle):
    _free = "dfgpu_stream_free"


# ----------------------------------------------------------------------------- RecordBatch
class RecordBatch:
    """RecordBatch resident in HBM, owned by the C++ layer (may carry a selection mask and lazy join-output columns)."""

    def __init__(self, ctx: Context, handle: _BatchH):
        self._ctx, self._h = ctx, handle
        self._cols = None

    @staticmethod
    def from_arrays(ctx: Context, names: Sequence[str], arrays: Sequence[Array]) -> "RecordBatch":
        out = C.c_void_p()
        _check(_lib().dfgpu_batch_new(_strs(names), _ptrs([a.h for a in arrays]), len(arrays), C.byref(out)))
        b = RecordBatch(ctx, _BatchH(out))
        b._keep = list(arrays)
        return b

    @property
    def ctx(self) -> Context:
        return self._ctx

    @property
    def num_columns(self) -> int:
        return _lib().dfgpu_batch_num_columns(self._h.h)

    @property
    def num_rows(self) -> int:
        n = C.c_int64()
        _check(_lib().dfgpu_batch_num_rows(self._ctx.h, self._h.h, C.byref(n)))
        return n.value

    def column(self, i: int) -> Array:
        out = C.c_void_p()
        _check(_lib().dfgpu_batch_column(self._ctx.h, self._h.h, i, C.byref(out)))
        return Array(self._ctx, out)

    @property
    def columns(self) -> List[Array]:
        if self._cols is None:
            _check(_lib().dfgpu_batch_materialize(self._ctx.h, self._h.h))      # all pending gathers at once (shared index arrays: one pass)
            self._cols = [self.column(i) for i in range(self.num_columns)]
        return self._cols

    @property
    def schema(self) -> Schema:
        names = [_lib().dfgpu_batch_column_name(self._h.h, i).decode() for i in range(self.num_columns)]
        return Schema([field_of_array(n, c) for n, c in zip(names, self.columns)])

    def materialize(self) -> "RecordBatch":
        return self

    def to_arrow(self):
        import pyarrow as pa
        names = self.schema.names()
        return pa.table({n + (f"#{i}" if names.count(n) > 1 else ""): c.to_arrow() for i, (n, c) in enumerate(zip(names, self.columns))})


def batch_from_arrow(ctx: Context, table) -> RecordBatch:
    cols = [ctx.from_arrow(table.column(i)) for i in range(table.num_columns)]
    return RecordBatch.from_arrays(ctx, table.schema.names, cols)


# ----------------------------------------------------------------------------- PhysicalExpr builders
class PhysicalExpr:
    def _build(self, ctx: Context) -> _ExprH:
        raise NotImplementedError

    def handle(self, ctx: Context) -> _ExprH:
        key = id(ctx)
        cache = self.__dict__.setdefault("_cache", {})
        if key not in cache:
            cache[key] = self._build(ctx)
        return cache[key]


class Column(PhysicalExpr):
    def __init__(self, name: str, index: int):
        self.name, self.index = name, index

    @staticmethod
    def new_with_schema(name: str, schema: Schema) -> "Column":
        return Column(name, schema.index_of(name))

    def _build(self, ctx):
        out = C.c_void_p()
        _check(_lib().dfgpu_expr_column(self.name.encode(), self.index, C.byref(out)))
        return _ExprH(out)


class Literal(PhysicalExpr):
    def __init__(self, value, pa_type):
        self.value, self.pa_type = value, pa_type

    def _build(self, ctx):
        import pyarrow as pa
        self._arr = ctx.from_arrow(pa.array([self.value], type=self.pa_type))
        out = C.c_void_p()
        _check(_lib().dfgpu_expr_literal(self._arr.h, C.byref(out)))
        return _ExprH(out)


class BinaryExpr(PhysicalExpr):
    def __init__(self, left: PhysicalExpr, op: str, right: PhysicalExpr):
        if op not in _OPS:
            raise DfgpuError(4, f"operator {op} is not supported on device")
        self.left, self.op, self.right = left, op, right

    def _build(self, ctx):
        out = C.c_void_p()
        _check(_lib().dfgpu_expr_binary(self.left.handle(ctx).h, _OPS[self.op], self.right.handle(ctx).h, C.byref(out)))
        return _ExprH(out)


class NotExpr(PhysicalExpr):
    def __init__(self, arg):
        self.arg = arg

    def _build(self, ctx):
        out = C.c_void_p()
        _check(_lib().dfgpu_expr_not(self.arg.handle(ctx).h, C.byref(out)))
        return _ExprH(out)


class IsNullExpr(PhysicalExpr):
    def __init__(self, arg, negated=False):
        self.arg, self.negated = arg, negated

    def _build(self, ctx):
        out = C.c_void_p()
        _check(_lib().dfgpu_expr_is_null(self.arg.handle(ctx).h, int(self.negated), C.byref(out)))
        return _ExprH(out)


class NegativeExpr(PhysicalExpr):
    def __init__(self, arg):
        self.arg = arg

    def _build(self, ctx):
        out = C.c_void_p()
        _check(_lib().dfgpu_expr_negative(self.arg.handle(ctx).h, C.byref(out)))
        return _ExprH(out)


class CastExpr(PhysicalExpr):
    def __init__(self, arg, to_type: int, precision: int = 0, scale: int = 0):
        self.arg, self.to_type, self.precision, self.scale = arg, to_type, precision, scale

    def _build(self, ctx):
        out = C.c_void_p()
        _check(_lib().dfgpu_expr_cast(self.arg.handle(ctx).h, self.to_type, self.precision, self.scale, C.byref(out)))
        return _ExprH(out)


class InListExpr(PhysicalExpr):
    def __init__(self, arg, values, pa_type, negated=False):
        self.arg, self.values, self.pa_type, self.negated = arg, values, pa_type, negated

    def _build(self, ctx):
        import pyarrow as pa
        self._lst = ctx.from_arrow(pa.array(self.values, type=self.pa_type))
        out = C.c_void_p()
        _check(_lib().dfgpu_expr_in_list(self.arg.handle(ctx).h, self._lst.h, int(self.negated), C.byref(out)))
        return _ExprH(out)


# ----------------------------------------------------------------------------- ExecutionPlan builders
class Partitioning:
    def __init__(self, kind: str, n: int, exprs: Optional[List[PhysicalExpr]] = None):
        self.kind, self.n, self.exprs = kind, n, exprs or []

    @staticmethod
    def Hash(exprs, n):
        return Partitioning("Hash", n, list(exprs))

    @staticmethod
    def RoundRobinBatch(n):
        return Partitioning("RoundRobinBatch", n)

    @staticmethod
    def UnknownPartitioning(n):
        return Partitioning("UnknownPartitioning", n)

    def partition_count(self):
        return self.n


class ExecutionPlan:
    """Python description of a plan node; `handle(context)` builds (once per context) the C++ node."""

    def children(self) -> List["ExecutionPlan"]:
        return []

    def _build(self, context: TaskContext) -> _PlanH:
        raise NotImplementedError

    def handle(self, context: TaskContext) -> _PlanH:
        key = id(context.ctx)
        cache = self.__dict__.setdefault("_pcache", {})
        if key not in cache:
            cache[key] = self._build(context)
        return cache[key]

    def _new(self, out: C.c_void_p) -> _PlanH:
        return _PlanH(out)

    def output_partitioning(self) -> Partitioning:
        ch = self.children()
        return ch[0].output_partitioning() if ch else Partitioning.UnknownPartitioning(1)

    def schema(self) -> Schema:
        cache = self.__dict__.get("_pcache", {})
        for h in cache.values():
            n = _lib().dfgpu_plan_schema_len(h.h)
            return Schema([Field(_lib().dfgpu_plan_schema_name(h.h, i).decode(), 0) for i in range(n)])
        ch = self.children()
        return ch[0].schema() if ch else Schema([])

    def metrics(self, context: TaskContext) -> List[dict]:
        """≙ ExecutionPlan::metrics() over the tree (dfgpu_plan_metrics): one dict per node, pre-order, with the reference's metric names
        (output_rows, elapsed_compute, build_time, join_time, repartition_time; times in ns of device time).  Collected while the ctx option
        "collect_metrics" is 1."""
        buf = C.create_string_buffer(1 << 16)
        _check(_lib().dfgpu_plan_metrics(self.handle(context).h, buf, len(buf)))
        out = []
        for line in buf.value.decode().splitlines():
            depth, name, *kv = line.split(" ")
            out.append({"depth": int(depth), "name": name, **{k: int(v) for k, v in (x.split("=") for x in kv)}})
        return out

    def execute(self, partition: int, context: TaskContext) -> Iterator[RecordBatch]:
        """≙ ExecutionPlan::execute(partition, ctx) -> SendableRecordBatchStream; iterating = poll_next."""
        h = self.handle(context)
        s = C.c_void_p()
        _check(_lib().dfgpu_plan_execute(h.h, partition, context.ctx.h, context.batch_size, C.byref(s)))
        stream = _StreamH(s)
        while True:
            b = C.c_void_p()
            _check(_lib().dfgpu_stream_next(stream.h, C.byref(b)))
            if not b.value:
                return
            yield RecordBatch(context.ctx, _BatchH(b))


class FreshPlan(ExecutionPlan):
    """A node whose C++ tree is `template` with fresh run-once state (≙ ExecutionPlan::with_new_children over the same
    children, physical-plan/src/lib.rs:198-201): build the description once, take `with_fresh_state()` per execution."""

    def __init__(self, template: ExecutionPlan):
        self.template = template

    def children(self):
        return self.template.children()

    def output_partitioning(self):
        return self.template.output_partitioning()

    def _build(self, context: TaskContext) -> _PlanH:
        out = C.c_void_p()
        _check(_lib().dfgpu_plan_with_fresh_state(self.template.handle(context).h, C.byref(out)))
        return _PlanH(out)


def with_fresh_state(plan: ExecutionPlan) -> ExecutionPlan:
    return FreshPlan(plan.template if isinstance(plan, FreshPlan) else plan)


def _child_handle(child, context: TaskContext) -> _PlanH:
    """C++ handle of a child node; Python-only nodes (ShuffleExec) are run here and enter as a MemoryExec."""
    if isinstance(child, ExecutionPlan):
        return child.handle(context)
    cache = child.__dict__.setdefault("_pcache", {})
    key = id(context.ctx)
    if key not in cache:
        nparts = child.output_partitioning().partition_count()
        parts = [[b for b in child.execute(p, context)] for p in range(nparts)]
        cache[key] = MemoryExec(parts, None).handle(context)
    return cache[key]


def collect(plan: ExecutionPlan, context: TaskContext) -> List[RecordBatch]:
    """≙ physical_plan::collect (lib.rs:678-709): every partition; results re-sliced to batch_size."""
    out: List[RecordBatch] = []
    for p in range(plan.output_partitioning().partition_count()):
        for b in plan.execute(p, context):
            n = b.num_rows
            if n == 0:
                continue
            bs = context.batch_size
            if n <= bs:
                out.append(b)
            else:
                names = b.schema.names()
                cols = b.columns
                for off in range(0, n, bs):
                    ln = min(bs, n - off)
                    out.append(RecordBatch.from_arrays(context.ctx, names, [c.slice(off, ln) for c in cols]))
    return out


class MemoryExec(ExecutionPlan):
    def __init__(self, partitions: List[List[RecordBatch]], schema: Optional[Schema]):
        self.partitions, self._schema = partitions, schema

    def output_partitioning(self):
        return Partitioning.UnknownPartitioning(len(self.partitions))

    def schema(self):
        if self._schema is not None:
            return self._schema
        for p in self.partitions:
            for b in p:
                return b.schema
        return Schema([])

    def _build(self, context):
        flat = [b for p in self.partitions for b in p]
        sizes = (C.c_int32 * max(1, len(self.partitions)))(*[len(p) for p in self.partitions])
        out = C.c_void_p()
        _check(_lib().dfgpu_plan_memory(_ptrs([b._h.h for b in flat]), sizes, len(self.partitions), C.byref(out)))
        return self._new(out)

    def replace(self, partitions: List[List[RecordBatch]]):
        """New batches (same schema) for every C++ node already built from this description: the input slot of a plan segment that is
        built once and executed per step with `with_fresh_state`."""
        self.partitions = partitions
        flat = [b for p in partitions for b in p]
        sizes = (C.c_int32 * max(1, len(partitions)))(*[len(p) for p in partitions])
        for h in self.__dict__.get("_pcache", {}).values():
            _check(_lib().dfgpu_plan_memory_replace(h.h, _ptrs([b._h.h for b in flat]), sizes, len(partitions)))


class ParquetExec(ExecutionPlan):
    """≙ ParquetExec (core/src/datasource/physical_plan/parquet/mod.rs:78): scan of one parquet.ParquetFile.  projection = leaf indices or
    names; the file's row groups are dealt to `partitions` output partitions in contiguous runs; `prune` = [(column, min, max)] bounds checked
    against the row-group statistics (≙ parquet/row_groups.rs)."""

    def __init__(self, file, projection=None, partitions: int = 1, row_groups_per_batch: int = 1, prune=()):
        names = file.column_names()
        self.file = file
        self.projection = list(range(file.num_columns)) if projection is None else [names.index(c) if isinstance(c, str) else int(c) for c in projection]
        self.partitions, self.row_groups_per_batch = partitions, row_groups_per_batch
        self.prune = [((names.index(c) if isinstance(c, str) else int(c)), lo, hi) for c, lo, hi in prune]

    def output_partitioning(self):
        return Partitioning.UnknownPartitioning(self.partitions)

    def schema(self):
        return self.file.schema(self.projection)

    def _build(self, context):
        out = C.c_void_p()
        idx = (C.c_int32 * max(1, len(self.projection)))(*self.projection)
        _check(_lib().dfgpu_plan_parquet(self.file.h, idx, len(self.projection), self.partitions, self.row_groups_per_batch, C.byref(out)))
        h = self._new(out)
        for c, lo, hi in self.prune:
            _check(_lib().dfgpu_plan_parquet_prune(h.h, c, lo, hi))
        return h

    def row_groups_pruned(self, context) -> int:
        return _lib().dfgpu_plan_parquet_pruned(self.handle(context).h)


class CsvExec(ExecutionPlan):
    """≙ CsvExec (core/src/datasource/physical_plan/csv.rs:53): scan of one CSV file image (bytes, kept alive by this node) under the table's schema
    [(name, DFGPU type, precision, scale)] of every file column.  The image is cut at record boundaries into pieces of ~batch_bytes, one batch each, dealt to
    `partitions` in contiguous runs (the reference's byte-range file groups, csv.rs:362-420)."""

    def __init__(self, data: bytes, file_schema, projection=None, partitions: int = 1, has_header: bool = True, delimiter: str = ",", quote: str = '"', batch_bytes: int = 0, escape: Optional[str] = None):
        names = [f[0] for f in file_schema]
        self.data, self.file_schema = bytes(data) if not isinstance(data, bytes) else data, [tuple(f) for f in file_schema]
        self.projection = sorted(range(len(names)) if projection is None else [names.index(c) if isinstance(c, str) else int(c) for c in projection])
        self.partitions, self.has_header, self.delimiter, self.quote, self.batch_bytes, self.escape = partitions, has_header, delimiter, quote, batch_bytes, escape

    def output_partitioning(self):
        return Partitioning.UnknownPartitioning(self.partitions)

    def schema(self):
        return Schema([Field(*self.file_schema[c][:4]) for c in self.projection])

    def _build(self, context):
        out = C.c_void_p(); n = len(self.file_schema)
        names = (C.c_char_p * n)(*[f[0].encode() for f in self.file_schema])
        types = (C.c_int32 * (3 * n))(*[v for f in self.file_schema for v in f[1:4]])
        idx = (C.c_int32 * len(self.projection))(*self.projection)
        _check(_lib().dfgpu_plan_csv(C.cast(C.c_char_p(self.data), C.c_void_p), len(self.data), ord(self.delimiter), ord(self.quote), ord(self.escape) if self.escape else 0, 1 if self.has_header else 0, names, types, n,
                                     idx, len(self.projection), self.partitions, self.batch_bytes, C.byref(out)))
        return self._new(out)


class FilterExec(ExecutionPlan):
    def __init__(self, predicate: PhysicalExpr, input):
        self.predicate, self.input = predicate, input

    def children(self):
        return [self.input]

    def _build(self, context):
        out = C.c_void_p()
        _check(_lib().dfgpu_plan_filter(self.predicate.handle(context.ctx).h, _child_handle(self.input, context).h, C.byref(out)))
        return self._new(out)


class ProjectionExec(ExecutionPlan):
    def __init__(self, exprs: List[Tuple[PhysicalExpr, str]], input):
        self.exprs, self.input = exprs, input

    def children(self):
        return [self.input]

    def _build(self, context):
        out = C.c_void_p()
        _check(_lib().dfgpu_plan_projection(_ptrs([e.handle(context.ctx).h for e, _ in self.exprs]), _strs([n for _, n in self.exprs]), len(self.exprs),
                                            _child_handle(self.input, context).h, C.byref(out)))
        return self._new(out)


class CoalesceBatchesExec(ExecutionPlan):
    def __init__(self, input, target_batch_size: int):
        self.input, self.target_batch_size = input, target_batch_size

    def children(self):
        return [self.input]

    def _build(self, context):
        out = C.c_void_p()
        _check(_lib().dfgpu_plan_coalesce_batches(_child_handle(self.input, context).h, self.target_batch_size, C.byref(out)))
        return self._new(out)


class CoalescePartitionsExec(ExecutionPlan):
    def __init__(self, input):
        self.input = input

    def children(self):
        return [self.input]

    def output_partitioning(self):
        return Partitioning.UnknownPartitioning(1)

    def _build(self, context):
        out = C.c_void_p()
        _check(_lib().dfgpu_plan_coalesce_partitions(_child_handle(self.input, context).h, C.byref(out)))
        return self._new(out)


class RepartitionExec(ExecutionPlan):
    def __init__(self, input, partitioning: Partitioning):
        if partitioning.kind not in ("Hash", "RoundRobinBatch"):
            raise DfgpuError(4, f"Unsupported repartitioning scheme {partitioning.kind}")
        self.input, self.partitioning = input, partitioning

    def children(self):
        return [self.input]

    def output_partitioning(self):
        return self.partitioning

    def _build(self, context):
        exprs = self.partitioning.exprs if self.partitioning.kind == "Hash" else []
        out = C.c_void_p()
        _check(_lib().dfgpu_plan_repartition(_child_handle(self.input, context).h, _ptrs([e.handle(context.ctx).h for e in exprs]), len(exprs), self.partitioning.n, C.byref(out)))
        return self._new(out)


@dataclass
class JoinFilter:
    expression: PhysicalExpr
    column_indices: List[Tuple[str, int]]
    schema: Schema


def collect_columns(expr) -> set:
    """≙ physical_expr::utils::collect_columns: every Column (name, index) an expression refers to."""
    out, stack = set(), [expr]
    while stack:
        e = stack.pop()
        if isinstance(e, Column):
            out.add((e.name, e.index))
        elif isinstance(e, PhysicalExpr):
            stack.extend(v for v in e.__dict__.values() if isinstance(v, PhysicalExpr))
            stack.extend(x for v in e.__dict__.values() if isinstance(v, (list, tuple)) for x in v if isinstance(x, PhysicalExpr))
    return out


def check_join_is_valid(left: Schema, right: Schema, on) -> None:
    """≙ joins/utils.rs:387-430 (check_join_is_valid / check_join_set_is_valid): every column of the `on` pairs must be a (name, index) of its side's schema."""
    lcols = {(f.name, i) for i, f in enumerate(left.fields)}
    rcols = {(f.name, i) for i, f in enumerate(right.fields)}
    on_left = set().union(*[collect_columns(l) for l, _ in on]) if on else set()
    on_right = set().union(*[collect_columns(r) for _, r in on]) if on else set()
    lm, rm = on_left - lcols, on_right - rcols
    if lm or rm:
        fmt = lambda m: "{" + ", ".join(f'Column {{ name: "{n}", index: {i} }}' for n, i in sorted(m)) + "}"
        raise DfgpuError(1, f'Error during planning: The left or right side of the join does not have all columns on "on": \nMissing on the left: {fmt(lm)}\nMissing on the right: {fmt(rm)}')


def _known_schema(plan):
    """The schema of a child when it is known without building the plan (leaf scans); None otherwise."""
    return plan.schema() if isinstance(plan, (MemoryExec, ParquetExec, CsvExec)) and (not isinstance(plan, MemoryExec) or plan._schema is not None) else None


class HashJoinExec(ExecutionPlan):
    def __init__(self, left, right, on: List[Tuple[PhysicalExpr, PhysicalExpr]], filter: Optional[JoinFilter], join_type: str,
                 partition_mode: str = "CollectLeft", null_equals_null: bool = False):
        if not on:
            raise DfgpuError(1, "Plan error: On constraints in HashJoinExec should be non-empty")     # hash_join.rs:303-305
        if join_type not in JOIN_TYPES:
            raise DfgpuError(5, f"unknown join type {join_type}")
        ls, rs = _known_schema(left), _known_schema(right)
        if ls is not None and rs is not None:
            check_join_is_valid(ls, rs, on)                       # hash_join.rs:312 (try_new)
        self.left, self.right, self.on, self.filter = left, right, on, filter
        self.join_type, self.mode, self.null_equals_null = join_type, partition_mode, null_equals_null

    def children(self):
        return [self.left, self.right]

    def output_partitioning(self):
        return self.right.output_partitioning()

    def schema(self):
        ls, rs = self.left.schema().fields, self.right.schema().fields
        if self.join_type in ("LeftSemi", "LeftAnti"):
            return Schema(list(ls))
        if self.join_type in ("RightSemi", "RightAnti"):
            return Schema(list(rs))
        return Schema(list(ls) + list(rs))

    def _build(self, context):
        ctx = context.ctx
        f = self.filter
        sides = (C.c_int32 * max(1, len(f.column_indices) if f else 1))(*([0 if s == "left" else 1 for s, _ in f.column_indices] if f else [0]))
        idxs = (C.c_int32 * max(1, len(f.column_indices) if f else 1))(*([i for _, i in f.column_indices] if f else [0]))
        out = C.c_void_p()
        _check(_lib().dfgpu_plan_hash_join(_child_handle(self.left, context).h, _child_handle(self.right, context).h,
                                           _ptrs([l.handle(ctx).h for l, _ in self.on]), _ptrs([r.handle(ctx).h for _, r in self.on]), len(self.on),
                                           f.expression.handle(ctx).h if f else None, sides, idxs, len(f.column_indices) if f else 0,
                                           JOIN_TYPES[self.join_type], 0 if self.mode == "CollectLeft" else 1, int(self.null_equals_null), C.byref(out)))
        return self._new(out)


class SortMergeJoinExec(ExecutionPlan):
    """≙ SortMergeJoinExec (joins/sort_merge_join.rs:62): equi-join of inputs sorted on the keys; rows in streamed-side order."""

    def __init__(self, left, right, on: List[Tuple[PhysicalExpr, PhysicalExpr]], join_type: str, null_equals_null: bool = False, filter: Optional[JoinFilter] = None):
        if join_type not in JOIN_TYPES:
            raise DfgpuError(5, f"unknown join type {join_type}")
        self.left, self.right, self.on, self.join_type, self.null_equals_null, self.filter = left, right, on, join_type, null_equals_null, filter

    def children(self):
        return [self.left, self.right]

    def output_partitioning(self):
        return self.left.output_partitioning()

    def schema(self):
        ls, rs = self.left.schema().fields, self.right.schema().fields
        return Schema(list(ls)) if self.join_type in ("LeftSemi", "LeftAnti") else Schema(list(ls) + list(rs))

    def _build(self, context):
        ctx = context.ctx
        f = self.filter
        nf = len(f.column_indices) if f else 0
        sides = (C.c_int32 * max(1, nf))(*([0 if s == "left" else 1 for s, _ in f.column_indices] if f else [0]))
        idx = (C.c_int32 * max(1, nf))(*([i for _, i in f.column_indices] if f else [0]))
        out = C.c_void_p()
        _check(_lib().dfgpu_plan_sort_merge_join(_child_handle(self.left, context).h, _child_handle(self.right, context).h,
                                                 _ptrs([l.handle(ctx).h for l, _ in self.on]), _ptrs([r.handle(ctx).h for _, r in self.on]), len(self.on),
                                                 f.expression.handle(ctx).h if f else None, sides, idx, nf, JOIN_TYPES[self.join_type], int(self.null_equals_null), C.byref(out)))
        return self._new(out)


class NestedLoopJoinExec(ExecutionPlan):
    """≙ NestedLoopJoinExec (joins/nested_loop_join.rs:84): join without equi-keys; `filter` = JoinFilter or None (cross join)."""

    def __init__(self, left, right, filter: Optional[JoinFilter], join_type: str):
        if join_type not in JOIN_TYPES:
            raise DfgpuError(5, f"unknown join type {join_type}")
        self.left, self.right, self.filter, self.join_type = left, right, filter, join_type

    def children(self):
        return [self.left, self.right]

    def output_partitioning(self):
        build_left = self.join_type in ("Right", "RightSemi", "RightAnti", "Full")
        return (self.right if build_left else self.left).output_partitioning()

    def schema(self):
        ls, rs = self.left.schema().fields, self.right.schema().fields
        if self.join_type in ("LeftSemi", "LeftAnti"):
            return Schema(list(ls))
        if self.join_type in ("RightSemi", "RightAnti"):
            return Schema(list(rs))
        return Schema(list(ls) + list(rs))

    def _build(self, context):
        ctx = context.ctx
        f = self.filter
        sides = (C.c_int32 * max(1, len(f.column_indices) if f else 1))(*([0 if s == "left" else 1 for s, _ in f.column_indices] if f else [0]))
        idxs = (C.c_int32 * max(1, len(f.column_indices) if f else 1))(*([i for _, i in f.column_indices] if f else [0]))
        out = C.c_void_p()
        _check(_lib().dfgpu_plan_nested_loop_join(_child_handle(self.left, context).h, _child_handle(self.right, context).h, f.expression.handle(ctx).h if f else None,
                                                  sides, idxs, len(f.column_indices) if f else 0, JOIN_TYPES[self.join_type], C.byref(out)))
        return self._new(out)


@dataclass
class AggregateFunctionExpr:
    fun: str
    expr: Optional[PhysicalExpr]
    name: str
    filter: Optional[PhysicalExpr] = None
    input_field: Optional[Field] = None      # argument data type (the AggregateExpr knows it in every mode)

    @property
    def kind(self) -> int:
        return {"SUM": capi.AGG_SUM, "AVG": capi.AGG_AVG, "COUNT": capi.AGG_COUNT, "MIN": capi.AGG_MIN, "MAX": capi.AGG_MAX, "COUNT DISTINCT": 5}[self.fun.upper()]


_AGG_MODES = {"Partial": 0, "Final": 1, "FinalPartitioned": 2, "Single": 3, "SinglePartitioned": 4}


@dataclass
class PhysicalGroupBy:
    """aggregates/mod.rs:103-160: `expr` the group expressions, `null_expr` the typed NULL literal standing in for each of them, `groups[s][i]`
    True when grouping set s replaces expression i by its NULL (GROUPING SETS / CUBE / ROLLUP).  A plain list of (expr, name) pairs is
    PhysicalGroupBy::new_single."""
    expr: List[Tuple[PhysicalExpr, str]]
    null_expr: List[Tuple[PhysicalExpr, str]]
    groups: List[List[bool]]


class AggregateExec(ExecutionPlan):
    def __init__(self, mode: str, group_by, aggr_expr: List[AggregateFunctionExpr], input, input_order_mode="Linear"):
        """input_order_mode ≙ InputOrderMode (physical-plan/src/ordering.rs:33): "Linear", "Sorted" (input sorted on every group key) or ("PartiallySorted", [indices of
        the group keys the input is sorted on]) -- the reference derives it from the input's output ordering (aggregates/mod.rs:get_aggregate_exprs_requirement /
        get_working_mode); here the caller states it.  Ordered modes emit finished groups after every input batch (GroupOrdering, aggregates/order/)."""
        self.input_order_mode = input_order_mode
        if mode not in _AGG_MODES:
            raise DfgpuError(5, f"unknown AggregateMode {mode}")
        self.grouping = group_by if isinstance(group_by, PhysicalGroupBy) else None
        if self.grouping is not None:
            group_by = self.grouping.expr
        for a in aggr_expr:
            if a.kind != capi.AGG_COUNT and a.input_field is None:
                raise DfgpuError(5, f"aggregate {a.name}: input_field (argument data type) is required")
        self.mode, self.group_by, self.aggr_expr, self.input = mode, group_by, aggr_expr, input

    def children(self):
        return [self.input]

    def output_partitioning(self):
        if self.mode in ("Final", "Single"):
            return Partitioning.UnknownPartitioning(1)
        return self.input.output_partitioning()

    def _build(self, context):
        ctx = context.ctx
        na = len(self.aggr_expr)
        kinds = (C.c_int32 * max(1, na))(*[a.kind for a in self.aggr_expr])
        args = (C.c_void_p * max(1, na))(*[a.expr.handle(ctx).h.value if a.expr is not None else None for a in self.aggr_expr])
        filts = (C.c_void_p * max(1, na))(*[a.filter.handle(ctx).h.value if a.filter is not None else None for a in self.aggr_expr])
        types = []
        for a in self.aggr_expr:
            f = a.input_field or Field("", capi.INT64)
            types += [f.dtype, f.precision, f.scale]
        tarr = (C.c_int32 * max(1, len(types)))(*types)
        out = C.c_void_p()
        _check(_lib().dfgpu_plan_aggregate(_AGG_MODES[self.mode], _ptrs([e.handle(ctx).h for e, _ in self.group_by]), _strs([n for _, n in self.group_by]), len(self.group_by),
                                           kinds, args, filts, _strs([a.name for a in self.aggr_expr]), tarr, na, _child_handle(self.input, context).h, C.byref(out)))
        om = self.input_order_mode
        if om != "Linear":
            idx = list(om[1]) if isinstance(om, (tuple, list)) else []
            code = 2 if om == "Sorted" else 1 if isinstance(om, (tuple, list)) and om[0] == "PartiallySorted" else -1
            _check(_lib().dfgpu_plan_aggregate_input_order(out, code, (C.c_int32 * max(1, len(idx)))(*idx), len(idx)))
        if self.grouping is not None:
            g = self.grouping
            flat = [1 if m else 0 for s in g.groups for m in s]
            _check(_lib().dfgpu_plan_aggregate_grouping_sets(out, _ptrs([e.handle(ctx).h for e, _ in g.null_expr]), len(g.null_expr), (C.c_uint8 * max(1, len(flat)))(*flat), len(g.groups)))
        return self._new(out)


@dataclass
class PhysicalSortExpr:
    expr: PhysicalExpr
    descending: bool = False
    nulls_first: bool = True


class SortExec(ExecutionPlan):
    def __init__(self, expr: List[PhysicalSortExpr], input, fetch: Optional[int] = None, preserve_partitioning: bool = False):
        self.expr, self.input, self.fetch, self.preserve_partitioning = expr, input, fetch, preserve_partitioning

    def children(self):
        return [self.input]

    def output_partitioning(self):
        return self.input.output_partitioning() if self.preserve_partitioning else Partitioning.UnknownPartitioning(1)

    def _build(self, context):
        ctx = context.ctx
        out = C.c_void_p()
        _check(_lib().dfgpu_plan_sort(_ptrs([s.expr.handle(ctx).h for s in self.expr]), bytes(int(s.descending) for s in self.expr), bytes(int(s.nulls_first) for s in self.expr),
                                      len(self.expr), -1 if self.fetch is None else int(self.fetch), int(self.preserve_partitioning), _child_handle(self.input, context).h, C.byref(out)))
        return self._new(out)


class SortPreservingMergeExec(ExecutionPlan):
    """≙ SortPreservingMergeExec::new(expr, input).with_fetch(fetch) (sorts/sort_preserving_merge.rs:67-120)."""

    def __init__(self, expr: List[PhysicalSortExpr], input, fetch: Optional[int] = None):
        self.expr, self.input, self.fetch = expr, input, fetch

    def children(self):
        return [self.input]

    def output_partitioning(self):
        return Partitioning.UnknownPartitioning(1)

    def _build(self, context):
        ctx = context.ctx
        out = C.c_void_p()
        _check(_lib().dfgpu_plan_sort_preserving_merge(_ptrs([s.expr.handle(ctx).h for s in self.expr]), bytes(int(s.descending) for s in self.expr),
                                                       bytes(int(s.nulls_first) for s in self.expr), len(self.expr), -1 if self.fetch is None else int(self.fetch),
                                                       _child_handle(self.input, context).h, C.byref(out)))
        return self._new(out)


def concat_batches(schema: Schema, batches: Sequence[RecordBatch]) -> Optional[RecordBatch]:
    batches = [b for b in batches if b.num_rows > 0] or list(batches[:1])
    if not batches:
        return None
    if len(batches) == 1:
        return batches[0]
    ctx = batches[0].ctx
    names = batches[0].schema.names()
    cols = [ctx.concat([b.columns[i] for b in batches]) for i in range(len(names))]
    return RecordBatch.from_arrays(ctx, names, cols)
