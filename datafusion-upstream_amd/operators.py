"""Host-side mirror of the reference's operator interface for the hot path, driving the HIP kernels
through the C ABI.  Same names, argument meaning and error behaviour as

  trait ExecutionPlan          datafusion/physical-plan/src/lib.rs:115-405   (execute(partition, ctx) -> stream)
  trait PhysicalExpr           datafusion/physical-expr/src/physical_expr.rs:96-123
  FilterExec / ProjectionExec  physical-plan/src/filter.rs:56-66, projection.rs:52-62
  HashJoinExec                 physical-plan/src/joins/hash_join.rs:283-330, 574-656
  AggregateExec                physical-plan/src/aggregates/mod.rs:242-269 (+ row_hash.rs stream)
  SortExec                     physical-plan/src/sorts/sort.rs:719-733
  RepartitionExec              physical-plan/src/repartition/mod.rs:232-294
  CoalesceBatchesExec          physical-plan/src/coalesce_batches.rs

so the parity tests read like the reference's own (`MemoryExec` + `execute` + `collect`).

MI355X-first differences (results identical, DESIGN.md "late materialisation"):
  * a RecordBatch may carry a `selection` mask (bit-packed Boolean in HBM) instead of compacted columns:
    FilterExec only computes the mask; join build/probe and group interning consume it fused, so filtered
    columns that are never needed are never gathered (the reference compacts every column in
    filter_record_batch, filter.rs:315-327).
  * batches are whole partitions (millions of rows) rather than 8192 rows, so every launch fills the
    256 CUs; `collect(..., batch_size)` re-slices the result to the session batch_size at the boundary.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Iterator, List, Optional, Sequence, Tuple

from . import capi
from .capi import DfgpuError
from .device import Array, Context, GroupValues, GroupsAccumulator, JoinTable, join_adjust_indices


# ----------------------------------------------------------------------------- schema / batches
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


class LazyColumn:
    """A join / gather output column that is not materialised yet: `take(source, indices)` runs on first use.
    Columns that a downstream ProjectionExec drops are never gathered, and a gather of a gather is composed on
    the (narrow) index arrays instead of moving the (wide) values twice (late materialisation)."""

    def __init__(self, source: Array, indices: Array):
        self.source, self.indices, self._arr = source, indices, None
        self.ctx = source.ctx

    def __len__(self) -> int:
        return len(self.indices)

    def get(self) -> Array:
        if self._arr is None:
            self._arr = self.ctx.take(self.source, self.indices)
        return self._arr

    def take(self, idx: Array):
        if self._arr is not None:
            return LazyColumn(self._arr, idx)
        return LazyColumn(self.source, self.ctx.take(self.indices, idx))


def lazy_take(col, idx: Array):
    """take(col, idx) without touching the values yet."""
    return col.take(idx) if isinstance(col, LazyColumn) else LazyColumn(col, idx)


class RecordBatch:
    """Columns in HBM (materialised Arrays or LazyColumns) + optional selection mask over their rows."""

    def __init__(self, schema: Schema, columns: Sequence, num_rows: Optional[int] = None, selection: Optional[Array] = None):
        self.schema = schema
        self.raw_columns = list(columns)
        self.base_rows = len(columns[0]) if columns else (num_rows or 0)
        self.selection = selection
        self._num_rows = num_rows

    @property
    def ctx(self) -> Context:
        return self.raw_columns[0].ctx

    def column(self, i: int) -> Array:
        c = self.raw_columns[i]
        if isinstance(c, LazyColumn):
            c = self.raw_columns[i] = c.get()
        return c

    @property
    def columns(self) -> List[Array]:
        return [self.column(i) for i in range(len(self.raw_columns))]

    def materialize(self) -> "RecordBatch":
        """Apply the selection: ≙ filter_record_batch (filter.rs:325)."""
        if self.selection is None:
            return self
        ctx = self.ctx
        sel = ctx.mask_to_indices(self.selection)
        cols = [lazy_take(c, sel) for c in self.raw_columns]
        return RecordBatch(self.schema, cols, num_rows=len(sel))

    @property
    def num_rows(self) -> int:
        if self.selection is None:
            return self.base_rows if self._num_rows is None else self._num_rows
        return len(self.ctx.mask_to_indices(self.selection))

    def to_arrow(self):
        import pyarrow as pa
        b = self.materialize()
        return pa.table({f.name + (f"#{i}" if b.schema.names().count(f.name) > 1 else ""): c.to_arrow() for i, (f, c) in enumerate(zip(b.schema.fields, b.columns))})


def field_of_array(name: str, a: Array) -> Field:
    d = a.describe()
    if d.type == capi.DICTIONARY:
        dd = d.dictionary.contents
        return Field(name, dd.type, dd.precision, dd.scale)
    return Field(name, d.type, d.precision, d.scale)


def batch_from_arrow(ctx: Context, table) -> RecordBatch:
    """pyarrow Table/RecordBatch -> device RecordBatch."""
    cols = [ctx.from_arrow(table.column(i)) for i in range(table.num_columns)]
    names = table.schema.names
    return RecordBatch(Schema([field_of_array(n, c) for n, c in zip(names, cols)]), cols, num_rows=table.num_rows)


@dataclass
class TaskContext:
    """≙ datafusion_execution::TaskContext (execution/src/task.rs:44-59)."""
    ctx: Context
    batch_size: int = 8192              # datafusion.execution.batch_size (common/src/config.rs:215)


# ----------------------------------------------------------------------------- PhysicalExpr
class ColumnarValue:
    """≙ ColumnarValue::{Array, Scalar} (expr/src/columnar_value.rs:35-40)."""

    def __init__(self, array: Array, is_scalar: bool = False):
        self.array, self.is_scalar = array, is_scalar

    def into_array(self, ctx: Context, num_rows: int) -> Array:
        if not self.is_scalar:
            return self.array
        import pyarrow as pa
        idx = ctx.from_arrow(pa.array([0] * num_rows, type=pa.uint32())) if num_rows else ctx.from_arrow(pa.array([], type=pa.uint32()))
        return ctx.take(self.array, idx)


class PhysicalExpr:
    def evaluate(self, batch: RecordBatch) -> ColumnarValue:
        raise NotImplementedError

    def data_field(self, schema: Schema, name: str) -> Field:
        raise NotImplementedError


class Column(PhysicalExpr):
    """expressions/column.rs:91"""

    def __init__(self, name: str, index: int):
        self.name, self.index = name, index

    @staticmethod
    def new_with_schema(name: str, schema: Schema) -> "Column":
        return Column(name, schema.index_of(name))

    def evaluate(self, batch):
        if self.index >= len(batch.raw_columns):
            raise DfgpuError(2, f"PhysicalExpr Column references column '{self.name}' at index {self.index} (zero-based) but input schema only has {len(batch.raw_columns)} columns")
        return ColumnarValue(batch.column(self.index))

    def __repr__(self):
        return f"{self.name}@{self.index}"


class Literal(PhysicalExpr):
    """expressions/literal.rs:73; value is a pyarrow scalar-compatible Python value + pyarrow type."""

    def __init__(self, value, pa_type):
        self.value, self.pa_type = value, pa_type
        self._cache = {}

    def evaluate(self, batch):
        import pyarrow as pa
        ctx = batch.ctx
        key = id(ctx)
        if key not in self._cache:
            self._cache[key] = ctx.from_arrow(pa.array([self.value], type=self.pa_type))
        return ColumnarValue(self._cache[key], is_scalar=True)

    def __repr__(self):
        return repr(self.value)


_OPS = {"+": capi.OP_ADD, "-": capi.OP_SUB, "*": capi.OP_MUL, "/": capi.OP_DIV, "%": capi.OP_REM,
        "=": capi.OP_EQ, "!=": capi.OP_NEQ, "<": capi.OP_LT, "<=": capi.OP_LTEQ, ">": capi.OP_GT, ">=": capi.OP_GTEQ,
        "IS DISTINCT FROM": capi.OP_DISTINCT, "IS NOT DISTINCT FROM": capi.OP_NOT_DISTINCT, "AND": capi.OP_AND, "OR": capi.OP_OR}


class BinaryExpr(PhysicalExpr):
    """expressions/binary.rs:259-315"""

    def __init__(self, left: PhysicalExpr, op: str, right: PhysicalExpr):
        if op not in _OPS:
            raise DfgpuError(4, f"operator {op} is not supported on device")
        self.left, self.op, self.right = left, op, right

    def evaluate(self, batch):
        l, r = self.left.evaluate(batch), self.right.evaluate(batch)
        ctx = batch.ctx
        out = ctx.binary(_OPS[self.op], l.array, r.array, l.is_scalar, r.is_scalar)
        return ColumnarValue(out, is_scalar=l.is_scalar and r.is_scalar)

    def __repr__(self):
        return f"{self.left!r} {self.op} {self.right!r}"


class NotExpr(PhysicalExpr):
    def __init__(self, arg):
        self.arg = arg

    def evaluate(self, batch):
        v = self.arg.evaluate(batch)
        return ColumnarValue(batch.ctx.not_(v.array), v.is_scalar)


class IsNullExpr(PhysicalExpr):
    def __init__(self, arg, negated=False):
        self.arg, self.negated = arg, negated

    def evaluate(self, batch):
        v = self.arg.evaluate(batch)
        return ColumnarValue(batch.ctx.is_null(v.array, self.negated), v.is_scalar)


class NegativeExpr(PhysicalExpr):
    def __init__(self, arg):
        self.arg = arg

    def evaluate(self, batch):
        v = self.arg.evaluate(batch)
        return ColumnarValue(batch.ctx.negative(v.array), v.is_scalar)


class CastExpr(PhysicalExpr):
    """expressions/cast.rs:121 (DEFAULT_DATAFUSION_CAST_OPTIONS: safe = false)"""

    def __init__(self, arg, to_type: int, precision: int = 0, scale: int = 0):
        self.arg, self.to_type, self.precision, self.scale = arg, to_type, precision, scale

    def evaluate(self, batch):
        v = self.arg.evaluate(batch)
        return ColumnarValue(batch.ctx.cast(v.array, self.to_type, self.precision, self.scale), v.is_scalar)


class InListExpr(PhysicalExpr):
    """expressions/in_list.rs:349; list is a pyarrow array of literals."""

    def __init__(self, arg, values, pa_type, negated=False):
        self.arg, self.values, self.pa_type, self.negated = arg, values, pa_type, negated

    def evaluate(self, batch):
        import pyarrow as pa
        v = self.arg.evaluate(batch)
        lst = batch.ctx.from_arrow(pa.array(self.values, type=self.pa_type))
        return ColumnarValue(batch.ctx.in_list(v.array, lst, self.negated), v.is_scalar)


# ----------------------------------------------------------------------------- ExecutionPlan
class Partitioning:
    """≙ physical-expr/src/partitioning.rs:108-116"""

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
    def schema(self) -> Schema:
        raise NotImplementedError

    def children(self) -> List["ExecutionPlan"]:
        return []

    def output_partitioning(self) -> Partitioning:
        ch = self.children()
        return ch[0].output_partitioning() if ch else Partitioning.UnknownPartitioning(1)

    def execute(self, partition: int, context: TaskContext) -> Iterator[RecordBatch]:
        raise NotImplementedError


def collect(plan: ExecutionPlan, context: TaskContext) -> List[RecordBatch]:
    """≙ physical_plan::collect (lib.rs:678-709): all partitions, results re-sliced to batch_size."""
    out: List[RecordBatch] = []
    for p in range(plan.output_partitioning().partition_count()):
        for b in plan.execute(p, context):
            b = b.materialize()
            n = b.num_rows
            if n == 0:
                continue
            bs = context.batch_size
            if n <= bs:
                out.append(b)
            else:
                for off in range(0, n, bs):
                    ln = min(bs, n - off)
                    out.append(RecordBatch(b.schema, [c.slice(off, ln) for c in b.columns], num_rows=ln))
    return out


def concat_batches(schema: Schema, batches: Sequence[RecordBatch]) -> Optional[RecordBatch]:
    batches = [b.materialize() for b in batches]
    batches = [b for b in batches if b.num_rows > 0] or batches[:1]
    if not batches:
        return None
    if len(batches) == 1:
        return batches[0]
    ctx = batches[0].ctx
    cols = [ctx.concat([b.columns[i] for b in batches]) for i in range(len(schema.fields))]
    return RecordBatch(schema, cols, num_rows=sum(b.num_rows for b in batches))


class MemoryExec(ExecutionPlan):
    """≙ physical-plan/src/memory.rs:40: partitions of in-memory (HBM) batches."""

    def __init__(self, partitions: List[List[RecordBatch]], schema: Schema):
        self.partitions, self._schema = partitions, schema

    def schema(self):
        return self._schema

    def output_partitioning(self):
        return Partitioning.UnknownPartitioning(len(self.partitions))

    def execute(self, partition, context):
        yield from self.partitions[partition]


class FilterExec(ExecutionPlan):
    """filter.rs:56-66.  Emits the input columns with a selection mask (predicate AND incoming mask);
    NULL predicate rows are dropped (arrow-select filter semantics)."""

    def __init__(self, predicate: PhysicalExpr, input: ExecutionPlan):
        self.predicate, self.input = predicate, input

    def schema(self):
        return self.input.schema()

    def children(self):
        return [self.input]

    def execute(self, partition, context):
        for batch in self.input.execute(partition, context):
            if batch.selection is not None and not _is_safe(self.predicate):
                batch = batch.materialize()
            v = self.predicate.evaluate(batch)
            mask = v.into_array(batch.ctx, batch.base_rows)
            if mask.type != capi.BOOL:
                raise DfgpuError(2, "Cannot create filter_array from non-boolean predicates")
            if batch.selection is not None:
                mask = batch.ctx.binary(capi.OP_AND, _known_mask(batch.ctx, mask), batch.selection)
            yield RecordBatch(batch.schema, batch.raw_columns, selection=mask)


def _known_mask(ctx: Context, mask: Array) -> Array:
    """NULL -> false so that Kleene AND with an upstream selection cannot resurrect rows."""
    if mask.describe().validity:
        return ctx.binary(capi.OP_AND, mask, ctx.is_null(mask, negate=True))
    return mask


def _is_safe(e: PhysicalExpr) -> bool:
    """Expressions that cannot raise on rows a selection mask has dropped."""
    if isinstance(e, (Column, Literal)):
        return True
    if isinstance(e, BinaryExpr):
        return e.op in ("=", "!=", "<", "<=", ">", ">=", "AND", "OR", "IS DISTINCT FROM", "IS NOT DISTINCT FROM") and _is_safe(e.left) and _is_safe(e.right)
    if isinstance(e, (NotExpr, IsNullExpr)):
        return _is_safe(e.arg)
    return False


class ProjectionExec(ExecutionPlan):
    """projection.rs:52-62; exprs = [(expr, output name)]."""

    def __init__(self, exprs: List[Tuple[PhysicalExpr, str]], input: ExecutionPlan):
        self.exprs, self.input = exprs, input
        self._schema = None

    def schema(self):
        if self._schema is None:
            ins = self.input.schema()
            fs = []
            for e, name in self.exprs:
                if isinstance(e, Column):
                    f = ins.fields[e.index]
                    fs.append(Field(name, f.dtype, f.precision, f.scale, f.nullable))
                else:
                    fs.append(Field(name, 0))          # resolved from the first batch
            self._schema = Schema(fs)
        return self._schema

    def children(self):
        return [self.input]

    def execute(self, partition, context):
        only_columns = all(isinstance(e, Column) for e, _ in self.exprs)
        for batch in self.input.execute(partition, context):
            if batch.selection is not None and not only_columns:
                needed = sorted(_columns_of([e for e, _ in self.exprs]))
                batch = _materialize_subset(batch, needed)
            if only_columns:            # Column = Arc clone in the reference (expressions/column.rs:91): pass (lazy) columns through untouched
                cols = [batch.raw_columns[e.index] for e, _ in self.exprs]
                schema = Schema([Field(n, f.dtype, f.precision, f.scale, f.nullable) for (e, n) in self.exprs for f in [batch.schema.fields[e.index]]])
            else:
                cols = [batch.raw_columns[e.index] if isinstance(e, Column) else e.evaluate(batch).into_array(batch.ctx, batch.base_rows) for e, _ in self.exprs]
                schema = Schema([batch.schema.fields[e.index] if isinstance(e, Column) else field_of_array(n, c) for (e, n), c in zip(self.exprs, cols)])
                schema = Schema([Field(n, f.dtype, f.precision, f.scale, f.nullable) for (_, n), f in zip(self.exprs, schema.fields)])
            self._schema = schema
            yield RecordBatch(schema, cols, num_rows=batch.base_rows, selection=batch.selection)


def _columns_of(exprs) -> set:
    out = set()
    for e in exprs:
        if isinstance(e, Column):
            out.add(e.index)
        for attr in ("left", "right", "arg"):
            if hasattr(e, attr):
                out |= _columns_of([getattr(e, attr)])
    return out


def _materialize_subset(batch: RecordBatch, needed: Sequence[int]) -> RecordBatch:
    """Compact only the referenced columns; unreferenced ones become all-NULL placeholders."""
    ctx = batch.ctx
    sel = ctx.mask_to_indices(batch.selection)
    cols = []
    for i in range(len(batch.raw_columns)):
        cols.append(ctx.take(batch.column(i), sel) if i in needed else None)
    n = len(sel)
    filler = None
    for i, c in enumerate(cols):
        if c is None:
            if filler is None:
                filler = ctx.new_null(capi.INT8, n)
            cols[i] = filler
    return RecordBatch(batch.schema, cols, num_rows=n)


class CoalesceBatchesExec(ExecutionPlan):
    """coalesce_batches.rs:198-260: buffer until >= target_batch_size rows, then concat_batches."""

    def __init__(self, input: ExecutionPlan, target_batch_size: int):
        self.input, self.target_batch_size = input, target_batch_size

    def schema(self):
        return self.input.schema()

    def children(self):
        return [self.input]

    def execute(self, partition, context):
        buf, rows = [], 0
        for batch in self.input.execute(partition, context):
            if batch.selection is not None:
                if batch.base_rows >= self.target_batch_size:      # device mega-batch: keep the fused mask
                    yield batch
                    continue
                batch = batch.materialize()
            n = batch.num_rows
            if n == 0:
                continue
            if n >= self.target_batch_size and not buf:
                yield batch
                continue
            buf.append(batch)
            rows += n
            if rows >= self.target_batch_size:
                yield concat_batches(self.schema(), buf)
                buf, rows = [], 0
        if buf:
            yield concat_batches(self.schema(), buf)


class CoalescePartitionsExec(ExecutionPlan):
    """coalesce_partitions.rs: N partitions -> 1 (unordered)."""

    def __init__(self, input: ExecutionPlan):
        self.input = input

    def schema(self):
        return self.input.schema()

    def children(self):
        return [self.input]

    def output_partitioning(self):
        return Partitioning.UnknownPartitioning(1)

    def execute(self, partition, context):
        for p in range(self.input.output_partitioning().partition_count()):
            yield from self.input.execute(p, context)


class RepartitionExec(ExecutionPlan):
    """repartition/mod.rs:232-294.  Hash: destination = create_hashes(exprs) % n (BatchPartitioner,
    :148-221).  In one process every output partition sees the slices of every input partition; on N GPUs
    the same per-destination slices are the all-to-all send buffers (exchange.py)."""

    def __init__(self, input: ExecutionPlan, partitioning: Partitioning):
        self.input, self.partitioning = input, partitioning
        self._cache = None

    def schema(self):
        return self.input.schema()

    def children(self):
        return [self.input]

    def output_partitioning(self):
        return self.partitioning

    def _run(self, context):
        n = self.partitioning.n
        outs: List[List[RecordBatch]] = [[] for _ in range(n)]
        rr = 0
        for p in range(self.input.output_partitioning().partition_count()):
            for batch in self.input.execute(p, context):
                if self.partitioning.kind == "RoundRobinBatch":
                    batch = batch.materialize()
                    if batch.num_rows == 0:
                        continue
                    outs[rr % n].append(batch)
                    rr += 1
                    continue
                if self.partitioning.kind != "Hash":
                    raise DfgpuError(4, f"Unsupported repartitioning scheme {self.partitioning.kind}")
                for dest, part in partition_batch(batch, self.partitioning.exprs, n):
                    outs[dest].append(part)
        return outs

    def execute(self, partition, context):
        if self._cache is None:
            self._cache = self._run(context)
        yield from self._cache[partition]


def partition_batch(batch: RecordBatch, exprs: Sequence[PhysicalExpr], n: int) -> List[Tuple[int, RecordBatch]]:
    """≙ BatchPartitioner::partition_iter (repartition/mod.rs:148-221) for Partitioning::Hash.  A fused selection
    mask is honoured without compacting the whole batch first: only the key columns are gathered for hashing and
    every output column is gathered once, straight into its destination slice."""
    ctx = batch.ctx
    sel = None
    kb = batch
    if batch.selection is not None:
        sel = ctx.mask_to_indices(batch.selection)
        needed = sorted(_columns_of(list(exprs)))
        cols = [ctx.take(batch.column(i), sel) if i in needed else None for i in range(len(batch.raw_columns))]
        filler = ctx.new_null(capi.INT8, len(sel))
        kb = RecordBatch(batch.schema, [c if c is not None else filler for c in cols], num_rows=len(sel))
    if kb.base_rows == 0:
        return []
    keys = [e.evaluate(kb).into_array(ctx, kb.base_rows) for e in exprs]
    indices, counts = ctx.hash_partition(keys, n)
    out, off = [], 0
    for dest, cnt in enumerate(counts):
        if cnt:
            idx = indices.slice(off, cnt)
            rows = ctx.take(sel, idx) if sel is not None else idx
            out.append((dest, RecordBatch(batch.schema, [lazy_take(c, rows) for c in batch.raw_columns], num_rows=cnt)))
        off += cnt
    return out


# ----------------------------------------------------------------------------- HashJoinExec
@dataclass
class JoinFilter:
    """≙ joins/utils.rs JoinFilter: expression over an intermediate batch built from
    column_indices = [(side, index)], side in {"left", "right"}."""
    expression: PhysicalExpr
    column_indices: List[Tuple[str, int]]
    schema: Schema


JOIN_TYPES = {"Inner": capi.JOIN_INNER, "Left": capi.JOIN_LEFT, "Right": capi.JOIN_RIGHT, "Full": capi.JOIN_FULL,
              "LeftSemi": capi.JOIN_LEFT_SEMI, "RightSemi": capi.JOIN_RIGHT_SEMI, "LeftAnti": capi.JOIN_LEFT_ANTI, "RightAnti": capi.JOIN_RIGHT_ANTI}


class HashJoinExec(ExecutionPlan):
    """joins/hash_join.rs: left = build side, right = probe side; mode "CollectLeft" | "Partitioned"."""

    def __init__(self, left: ExecutionPlan, right: ExecutionPlan, on: List[Tuple[PhysicalExpr, PhysicalExpr]], filter: Optional[JoinFilter],
                 join_type: str, partition_mode: str = "CollectLeft", null_equals_null: bool = False):
        if not on:
            raise DfgpuError(1, "Plan error: On constraints in HashJoinExec should be non-empty")     # hash_join.rs:303-305
        if join_type not in JOIN_TYPES:
            raise DfgpuError(5, f"unknown join type {join_type}")
        self.left, self.right, self.on, self.filter = left, right, on, filter
        self.join_type, self.mode, self.null_equals_null = join_type, partition_mode, null_equals_null
        self._shared_build = None

    # build_join_schema (joins/utils.rs:657-729)
    def schema(self):
        ls, rs = self.left.schema().fields, self.right.schema().fields
        jt = self.join_type
        if jt in ("LeftSemi", "LeftAnti"):
            return Schema(list(ls))
        if jt in ("RightSemi", "RightAnti"):
            return Schema(list(rs))
        return Schema(list(ls) + list(rs))

    def children(self):
        return [self.left, self.right]

    def output_partitioning(self):
        return self.right.output_partitioning()

    # collect_left_input (hash_join.rs:678-768)
    def _collect_build(self, partition: Optional[int], context: TaskContext):
        lp = self.left
        if partition is None:
            parts = range(lp.output_partitioning().partition_count())
        else:
            parts = [partition]
        batches = [b for p in parts for b in lp.execute(p, context)]
        ctx = context.ctx
        schema = lp.schema()
        fused_mask = None
        if len(batches) == 1 and batches[0].selection is not None:
            build = batches[0]                      # fused FilterExec: table built under the mask
            fused_mask = build.selection
            build = RecordBatch(build.schema, build.raw_columns, num_rows=build.base_rows)
        else:
            mats = [b.materialize() for b in batches]
            mats = [b for b in mats if b.num_rows > 0]
            seg = [b.num_rows for b in mats]
            build = concat_batches(schema, mats) if mats else None      # ORIGINAL input order (see include/dfgpu.h)
            self._segments = seg
        if build is None:
            return None
        keys = [l.evaluate(build).into_array(ctx, build.base_rows) for l, _ in self.on]
        table = JoinTable(ctx, keys, mask=fused_mask, null_equals_null=self.null_equals_null)
        return build, table, ([build.base_rows] if fused_mask is not None else getattr(self, "_segments", [build.base_rows]))

    def execute(self, partition, context):
        ctx = context.ctx
        jt = JOIN_TYPES[self.join_type]
        if self.mode == "CollectLeft":
            if self._shared_build is None:
                self._shared_build = (self._collect_build(None, context),)
            built = self._shared_build[0]
        else:
            built = self._collect_build(partition, context)
        out_schema = self.schema()
        lfields, rfields = self.left.schema().fields, self.right.schema().fields
        need_final = self.join_type in ("Left", "Full", "LeftSemi", "LeftAnti")       # need_produce_result_in_final
        any_probe = False
        for probe in self.right.execute(partition, context):
            if probe.base_rows == 0:
                continue
            any_probe = True
            mask = probe.selection
            pbase = RecordBatch(probe.schema, probe.raw_columns, num_rows=probe.base_rows)
            if built is None:
                bidx = ctx.from_arrow(_pa_empty("uint64"))
                pidx = ctx.from_arrow(_pa_empty("uint32"))
                build = None
                table = None
            else:
                build, table, _ = built
                pkeys = [r.evaluate(pbase).into_array(ctx, pbase.base_rows) for _, r in self.on]
                bidx, pidx = table.probe(pkeys, mask=mask)
                if self.filter is not None and len(bidx):
                    bidx, pidx = self._apply_filter(ctx, build, pbase, bidx, pidx)
                if need_final:
                    table.mark_visited(bidx)
            if self.join_type in ("Right", "Full", "RightSemi", "RightAnti"):
                if mask is not None:
                    # unmatched probe rows must only be produced for selected rows: compact first (rare path)
                    raise DfgpuError(4, "Right/Full/RightSemi/RightAnti join over a fused probe-side selection; materialise the probe input (CoalesceBatchesExec)")
                bidx, pidx = join_adjust_indices(ctx, bidx, pidx, 0, pbase.base_rows, jt)
            elif self.join_type in ("LeftSemi", "LeftAnti"):
                continue
            yield self._build_batch(ctx, out_schema, build, pbase, bidx, pidx, lfields, rfields)
        if need_final and built is not None:
            build, table, segments = built
            fidx = table.final_indices(jt)
            fidx = _reference_final_order(ctx, fidx, segments)
            n = len(fidx)
            lcols = [lazy_take(c, fidx) for c in build.raw_columns]
            if self.join_type in ("LeftSemi", "LeftAnti"):
                yield RecordBatch(out_schema, lcols, num_rows=n)
            else:
                rcols = [ctx.new_null(f.dtype, n, f.precision, f.scale) for f in rfields]
                yield RecordBatch(out_schema, lcols + rcols, num_rows=n)

    def _apply_filter(self, ctx, build, probe, bidx, pidx):
        """apply_join_filter_to_indices (joins/utils.rs:1143-1176)"""
        cols = []
        for side, index in self.filter.column_indices:
            cols.append(ctx.take(build.column(index), bidx) if side == "left" else ctx.take(probe.column(index), pidx))
        inter = RecordBatch(self.filter.schema, cols, num_rows=len(bidx))
        m = self.filter.expression.evaluate(inter).into_array(ctx, len(bidx))
        return ctx.filter(bidx, m), ctx.filter(pidx, m)

    def _build_batch(self, ctx, schema, build, probe, bidx, pidx, lfields, rfields):
        """build_batch_from_indices (joins/utils.rs:1180-1230)"""
        n = len(pidx)
        jt = self.join_type
        cols = []
        if jt not in ("RightSemi", "RightAnti"):
            for i, f in enumerate(lfields):
                cols.append(lazy_take(build.raw_columns[i], bidx) if build is not None else ctx.new_null(f.dtype, n, f.precision, f.scale))
        if jt not in ("LeftSemi", "LeftAnti"):
            for i in range(len(rfields)):
                cols.append(lazy_take(probe.raw_columns[i], pidx))
        return RecordBatch(schema, cols, num_rows=n)


def _pa_empty(kind: str):
    import pyarrow as pa
    return pa.array([], type=getattr(pa, kind)())


def _reference_final_order(ctx: Context, fidx: Array, segments: Sequence[int]) -> Array:
    """The device table indexes the build side in original input order; the reference concatenates the build
    batches in REVERSED order (hash_join.rs:746,764) and emits final unmatched/semi rows in ascending index of
    that batch (joins/utils.rs:1119-1141) -- i.e. last input batch first.  Re-order the ascending list by segment."""
    if len(segments) <= 1 or len(fidx) == 0:
        return fidx
    import numpy as np
    import pyarrow as pa
    host = fidx.to_numpy().astype(np.uint64)
    bounds = np.cumsum([0] + list(segments))
    pieces = [host[(host >= bounds[s]) & (host < bounds[s + 1])] for s in range(len(segments))]
    return ctx.from_arrow(pa.array(np.concatenate(pieces[::-1]) if pieces else host, type=pa.uint64()))


# ----------------------------------------------------------------------------- AggregateExec
@dataclass
class AggregateFunctionExpr:
    """≙ AggregateExpr (physical-expr/src/aggregate/mod.rs:74-123) for Sum / Avg / Count / Min / Max."""
    fun: str                       # "SUM" | "AVG" | "COUNT" | "MIN" | "MAX"
    expr: Optional[PhysicalExpr]   # None = COUNT(*)
    name: str
    filter: Optional[PhysicalExpr] = None
    input_field: Optional[Field] = None   # argument data type (needed by Final modes for AVG(Decimal128) result precision)

    @property
    def kind(self) -> int:
        return {"SUM": capi.AGG_SUM, "AVG": capi.AGG_AVG, "COUNT": capi.AGG_COUNT, "MIN": capi.AGG_MIN, "MAX": capi.AGG_MAX}[self.fun.upper()]

    def state_names(self) -> List[str]:
        if self.fun.upper() == "AVG":                       # average.rs state_fields: count, sum
            return [f"{self.name}[count]", f"{self.name}[sum]"]
        return [f"{self.name}[{self.fun.lower()}]" if self.fun.upper() != "COUNT" else f"{self.name}[count]"]


class AggregateExec(ExecutionPlan):
    """aggregates/mod.rs:242-269.  mode in Partial | Final | FinalPartitioned | Single | SinglePartitioned.
    group_by = [(expr, name)]."""

    def __init__(self, mode: str, group_by: List[Tuple[PhysicalExpr, str]], aggr_expr: List[AggregateFunctionExpr], input: ExecutionPlan):
        if mode not in ("Partial", "Final", "FinalPartitioned", "Single", "SinglePartitioned"):
            raise DfgpuError(5, f"unknown AggregateMode {mode}")
        self.mode, self.group_by, self.aggr_expr, self.input = mode, group_by, aggr_expr, input
        self._schema = None

    def children(self):
        return [self.input]

    def output_partitioning(self):
        if self.mode in ("Final", "Single"):
            return Partitioning.UnknownPartitioning(1)
        return self.input.output_partitioning()

    def schema(self):
        if self._schema is None:
            names = [n for _, n in self.group_by]
            for a in self.aggr_expr:
                names += a.state_names() if self.mode == "Partial" else [a.name]
            self._schema = Schema([Field(n, 0) for n in names])
        return self._schema

    def execute(self, partition, context):
        ctx = context.ctx
        merging = self.mode in ("Final", "FinalPartitioned")
        if self.mode in ("Final", "Single"):
            inputs = [b for p in range(self.input.output_partitioning().partition_count()) for b in self.input.execute(p, context)]
        else:
            inputs = self.input.execute(partition, context)
        if not self.group_by:
            raise DfgpuError(4, "AggregateExec without GROUP BY (AggregateStream, no_grouping.rs) is not on the device path yet")
        groups = GroupValues(ctx, len(self.group_by))
        accs: List[Optional[GroupsAccumulator]] = [None] * len(self.aggr_expr)
        for batch in inputs:
            if batch.base_rows == 0:
                continue
            mask = batch.selection
            base = RecordBatch(batch.schema, batch.raw_columns, num_rows=batch.base_rows)
            gcols = [e.evaluate(base).into_array(ctx, base.base_rows) for e, _ in self.group_by]
            gids = groups.intern(gcols, mask=mask)                         # group_aggregate_batch (row_hash.rs:524-613)
            total = len(groups)
            col = len(self.group_by)
            for i, a in enumerate(self.aggr_expr):
                if merging:
                    nst = 2 if a.fun.upper() == "AVG" else 1
                    states = [base.column(col + k) for k in range(nst)]
                    col += nst
                    if accs[i] is None:
                        accs[i] = self._make_acc_from_state(ctx, a, states)
                    accs[i].merge_batch(states, gids, None, total)
                else:
                    vals = a.expr.evaluate(base).into_array(ctx, base.base_rows) if a.expr is not None else None
                    filt = a.filter.evaluate(base).into_array(ctx, base.base_rows) if a.filter is not None else None
                    if accs[i] is None:
                        accs[i] = self._make_acc(ctx, a, vals)
                    accs[i].update_batch(vals, gids, filt, total)
        if len(groups) == 0:
            return
        out_cols = groups.emit()
        total = len(groups)
        empty_ids = ctx.from_arrow(_pa_empty("uint32"))
        for i, a in enumerate(self.aggr_expr):
            if accs[i] is None:
                raise DfgpuError(2, "aggregate saw no input batch")
            accs[i].update_batch(None, empty_ids, None, total)      # zero-row update: only grows the state to `total` groups
            out_cols += accs[i].state() if self.mode == "Partial" else [accs[i].evaluate()]
        names = [n for _, n in self.group_by]
        for a in self.aggr_expr:
            names += a.state_names() if self.mode == "Partial" else [a.name]
        schema = Schema([field_of_array(n, c) for n, c in zip(names, out_cols)])
        self._schema = schema
        yield RecordBatch(schema, out_cols, num_rows=total)

    @staticmethod
    def _make_acc(ctx, a: AggregateFunctionExpr, vals: Optional[Array]) -> GroupsAccumulator:
        if vals is None:
            return GroupsAccumulator(ctx, a.kind, capi.INT64)
        f = field_of_array("v", vals)
        return GroupsAccumulator(ctx, a.kind, f.dtype, f.precision, f.scale)

    @staticmethod
    def _make_acc_from_state(ctx, a: AggregateFunctionExpr, states: Sequence[Array]) -> GroupsAccumulator:
        """Final modes receive state columns; recover the input type the Partial accumulator was created for."""
        kind = a.kind
        if kind == capi.AGG_COUNT:
            return GroupsAccumulator(ctx, kind, capi.INT64)
        if a.input_field is not None:           # the AggregateExpr knows its argument type in every mode
            f = a.input_field
            return GroupsAccumulator(ctx, kind, f.dtype, f.precision, f.scale)
        f = field_of_array("s", states[-1])
        p = f.precision
        if f.dtype == capi.DECIMAL128 and kind in (capi.AGG_SUM, capi.AGG_AVG):
            p = max(1, f.precision - 10)        # inverse of sum_return_type (aggregates.rs:397-416); exact unless clamped at 38
        return GroupsAccumulator(ctx, kind, f.dtype, p, f.scale)


def _pa_type_of(f: Field):
    import pyarrow as pa
    m = {capi.BOOL: pa.bool_(), capi.INT8: pa.int8(), capi.INT16: pa.int16(), capi.INT32: pa.int32(), capi.INT64: pa.int64(),
         capi.UINT8: pa.uint8(), capi.UINT16: pa.uint16(), capi.UINT32: pa.uint32(), capi.UINT64: pa.uint64(),
         capi.FLOAT32: pa.float32(), capi.FLOAT64: pa.float64(), capi.DATE32: pa.date32(), capi.UTF8: pa.utf8()}
    if f.dtype == capi.DECIMAL128:
        return pa.decimal128(f.precision, f.scale)
    return m[f.dtype]


# ----------------------------------------------------------------------------- SortExec
@dataclass
class PhysicalSortExpr:
    """physical-expr/src/sort_expr.rs:34-74; SortOptions default = ASC NULLS LAST in SQL, arrow default nulls_first=True."""
    expr: PhysicalExpr
    descending: bool = False
    nulls_first: bool = True


class SortExec(ExecutionPlan):
    """sorts/sort.rs:719-733: buffer the partition, lexsort_to_indices + take (sort_batch :584-609);
    fetch = TopK row count."""

    def __init__(self, expr: List[PhysicalSortExpr], input: ExecutionPlan, fetch: Optional[int] = None, preserve_partitioning: bool = False):
        self.expr, self.input, self.fetch, self.preserve_partitioning = expr, input, fetch, preserve_partitioning

    def schema(self):
        return self.input.schema()

    def children(self):
        return [self.input]

    def output_partitioning(self):
        return self.input.output_partitioning() if self.preserve_partitioning else Partitioning.UnknownPartitioning(1)

    def execute(self, partition, context):
        ctx = context.ctx
        if self.preserve_partitioning:
            batches = list(self.input.execute(partition, context))
        else:
            batches = [b for p in range(self.input.output_partitioning().partition_count()) for b in self.input.execute(p, context)]
        schema = batches[0].schema if batches else self.input.schema()
        batch = concat_batches(schema, batches) if batches else None
        if batch is None or batch.num_rows == 0:
            return
        keys = [s.expr.evaluate(batch).into_array(ctx, batch.num_rows) for s in self.expr]
        idx = ctx.sort_to_indices(keys, [s.descending for s in self.expr], [s.nulls_first for s in self.expr], self.fetch)
        yield RecordBatch(batch.schema, [lazy_take(c, idx) for c in batch.raw_columns], num_rows=len(idx))
