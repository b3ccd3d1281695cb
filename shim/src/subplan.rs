//! `GpuSubplanExec`: a GPU-supported subtree of the physical plan executed by libdfgpu.so's C++ plan layer.
//!
//! The subtree is translated node by node into `dfgpu_plan_*` handles (children first).  A child the device cannot run stays a CPU
//! `ExecutionPlan`: it is executed here, its batches are imported (Arrow C Data Interface) and enter the device plan as a `MemoryExec`
//! (`dfgpu_plan_memory`), partition for partition.  `execute(partition)` = `dfgpu_plan_execute`; the stream's `poll_next` =
//! `dfgpu_stream_next` + one `dfgpu_array_export_arrow` per column.  All plan properties (schema, partitioning, ordering, distribution
//! requirements) are those of the CPU subtree it replaces, which is kept for exactly that purpose.
use std::any::Any;
use std::ffi::c_char;
use std::fmt;
use std::pin::Pin;
use std::ptr::{null, null_mut};
use std::sync::Arc;
use std::task::{Context, Poll};

use arrow::datatypes::{DataType, SchemaRef};
use arrow::record_batch::RecordBatch;
use datafusion_common::{internal_err, not_impl_err, DataFusionError, JoinType, Result, ScalarValue};
use datafusion_execution::{RecordBatchStream, SendableRecordBatchStream, TaskContext};
use datafusion_expr::Operator;
use datafusion_physical_expr::expressions::{BinaryExpr, CastExpr, Column, IsNotNullExpr, IsNullExpr, Literal, NegativeExpr, NotExpr};
use datafusion_physical_expr::{AggregateExpr, PhysicalExpr, PhysicalSortExpr};
use datafusion_physical_plan::aggregates::{AggregateExec, AggregateMode};
use datafusion_physical_plan::coalesce_batches::CoalesceBatchesExec;
use datafusion_physical_plan::coalesce_partitions::CoalescePartitionsExec;
use datafusion_physical_plan::filter::FilterExec;
use datafusion_physical_plan::joins::{HashJoinExec, PartitionMode};
use datafusion_physical_plan::projection::ProjectionExec;
use datafusion_physical_plan::repartition::RepartitionExec;
use datafusion_physical_plan::sorts::sort::SortExec;
use datafusion_physical_plan::sorts::sort_preserving_merge::SortPreservingMergeExec;
use datafusion_physical_plan::{DisplayAs, DisplayFormatType, ExecutionPlan, Partitioning, PlanProperties};
use futures::{Stream, StreamExt, TryStreamExt};

use crate::ffi::*;

/// Arrow DataType -> (DFGPU type, precision, scale); None = not supported on the device
pub fn gpu_type(t: &DataType) -> Option<(i32, i32, i32)> {
    Some(match t {
        DataType::Boolean => (1, 0, 0), DataType::Int8 => (2, 0, 0), DataType::Int16 => (3, 0, 0), DataType::Int32 => (4, 0, 0), DataType::Int64 => (5, 0, 0),
        DataType::UInt8 => (6, 0, 0), DataType::UInt16 => (7, 0, 0), DataType::UInt32 => (8, 0, 0), DataType::UInt64 => (9, 0, 0),
        DataType::Float32 => (10, 0, 0), DataType::Float64 => (11, 0, 0), DataType::Date32 => (12, 0, 0),
        DataType::Decimal128(p, s) => (13, *p as i32, *s as i32), DataType::Utf8 => (14, 0, 0),
        DataType::Dictionary(_, v) if matches!(**v, DataType::Utf8) => (14, 0, 0),
        _ => return None,
    })
}

/// A device plan under construction: the root handle plus everything that must outlive it.
struct Built { plan: GpuPlan, _exprs: Vec<GpuExpr>, _children: Vec<Built>, _batches: Vec<GpuBatch>, _arrays: Vec<GpuArray> }

/// PhysicalExpr -> dfgpu_expr (physical-expr/src/expressions/{column,literal,binary,not,is_null,negative,cast}.rs)
fn translate_expr(ctx: &GpuCtx, e: &Arc<dyn PhysicalExpr>, keep: &mut Vec<GpuExpr>, arrays: &mut Vec<GpuArray>) -> Result<*const dfgpu_expr> {
    let any = e.as_any();
    let mut out = null_mut();
    if let Some(c) = any.downcast_ref::<Column>() {
        check_exec(unsafe { dfgpu_expr_column(cstring(c.name()).as_ptr(), c.index() as i32, &mut out) })?;
    } else if let Some(l) = any.downcast_ref::<Literal>() {
        let scalar: &ScalarValue = l.value();
        let a = import_array(ctx, &scalar.to_array_of_size(1)?)?;        // ColumnarValue::Scalar = a length-1 array
        check_exec(unsafe { dfgpu_expr_literal(a.0, &mut out) })?;
        arrays.push(a);
    } else if let Some(b) = any.downcast_ref::<BinaryExpr>() {
        let op = match b.op() {
            Operator::Plus => OP_ADD, Operator::Minus => OP_SUB, Operator::Multiply => OP_MUL, Operator::Divide => OP_DIV, Operator::Modulo => OP_REM,
            Operator::Eq => OP_EQ, Operator::NotEq => OP_NEQ, Operator::Lt => OP_LT, Operator::LtEq => OP_LTEQ, Operator::Gt => OP_GT, Operator::GtEq => OP_GTEQ,
            Operator::IsDistinctFrom => OP_DISTINCT, Operator::IsNotDistinctFrom => OP_NOT_DISTINCT, Operator::And => OP_AND, Operator::Or => OP_OR,
            other => return not_impl_err!("binary operator {other} on the device"),
        };
        let l = translate_expr(ctx, b.left(), keep, arrays)?;
        let r = translate_expr(ctx, b.right(), keep, arrays)?;
        check_exec(unsafe { dfgpu_expr_binary(l, op, r, &mut out) })?;
    } else if let Some(n) = any.downcast_ref::<NotExpr>() {
        let a = translate_expr(ctx, n.arg(), keep, arrays)?; check_exec(unsafe { dfgpu_expr_not(a, &mut out) })?;
    } else if let Some(n) = any.downcast_ref::<IsNullExpr>() {
        let a = translate_expr(ctx, n.arg(), keep, arrays)?; check_exec(unsafe { dfgpu_expr_is_null(a, 0, &mut out) })?;
    } else if let Some(n) = any.downcast_ref::<IsNotNullExpr>() {
        let a = translate_expr(ctx, n.arg(), keep, arrays)?; check_exec(unsafe { dfgpu_expr_is_null(a, 1, &mut out) })?;
    } else if let Some(n) = any.downcast_ref::<NegativeExpr>() {
        let a = translate_expr(ctx, n.arg(), keep, arrays)?; check_exec(unsafe { dfgpu_expr_negative(a, &mut out) })?;
    } else if let Some(c) = any.downcast_ref::<CastExpr>() {
        let (t, p, s) = gpu_type(c.cast_type()).ok_or_else(|| DataFusionError::NotImplemented(format!("cast to {} on the device", c.cast_type())))?;
        let a = translate_expr(ctx, c.expr(), keep, arrays)?; check_exec(unsafe { dfgpu_expr_cast(a, t, p, s, &mut out) })?;
    } else {
        return not_impl_err!("PhysicalExpr {e:?} on the device");
    }
    keep.push(GpuExpr(out));
    Ok(out as *const dfgpu_expr)
}

fn sort_args(ctx: &GpuCtx, exprs: &[PhysicalSortExpr], keep: &mut Vec<GpuExpr>, arrays: &mut Vec<GpuArray>) -> Result<(Vec<*const dfgpu_expr>, Vec<u8>, Vec<u8>)> {
    let mut e = vec![]; let (mut d, mut nf) = (vec![], vec![]);
    for s in exprs { e.push(translate_expr(ctx, &s.expr, keep, arrays)?); d.push(s.options.descending as u8); nf.push(s.options.nulls_first as u8); }
    Ok((e, d, nf))
}

/// Is this node one the device plan layer runs (include/dfgpu_exec.h)?  Data types and expressions are checked when it is translated.
pub fn is_gpu_node(p: &dyn ExecutionPlan) -> bool {
    let a = p.as_any();
    a.is::<FilterExec>() || a.is::<ProjectionExec>() || a.is::<CoalesceBatchesExec>() || a.is::<CoalescePartitionsExec>() || a.is::<HashJoinExec>() || a.is::<AggregateExec>()
        || a.is::<SortExec>() || a.is::<SortPreservingMergeExec>() || a.downcast_ref::<RepartitionExec>().map_or(false, |r| !matches!(r.partitioning(), Partitioning::UnknownPartitioning(_)))
}

/// Runs a CPU child to completion (all of its partitions) and hands its batches to the device as a MemoryExec.
async fn cpu_leaf(ctx: &GpuCtx, child: &Arc<dyn ExecutionPlan>, task: &Arc<TaskContext>) -> Result<Built> {
    let nparts = child.output_partitioning().partition_count();
    let (mut batches, mut sizes, mut arrays) = (Vec::<GpuBatch>::new(), Vec::<i32>::new(), Vec::<GpuArray>::new());
    let names: Vec<_> = child.schema().fields().iter().map(|f| cstring(f.name())).collect();
    let name_ptrs: Vec<*const c_char> = names.iter().map(|n| n.as_ptr()).collect();
    for p in 0..nparts {
        let got: Vec<RecordBatch> = child.execute(p, task.clone())?.try_collect().await?;
        sizes.push(got.len() as i32);
        for b in got {
            let cols: Vec<GpuArray> = b.columns().iter().map(|c| import_array(ctx, c)).collect::<Result<_>>()?;
            let ptrs: Vec<*const dfgpu_array> = cols.iter().map(|c| c.0 as *const _).collect();
            let mut out = null_mut();
            check_exec(unsafe { dfgpu_batch_new(name_ptrs.as_ptr(), ptrs.as_ptr(), ptrs.len() as i32, &mut out) })?;
            batches.push(GpuBatch(out)); arrays.extend(cols);
        }
    }
    let bptrs: Vec<*const dfgpu_batch> = batches.iter().map(|b| b.0 as *const _).collect();
    let mut plan = null_mut();
    check_exec(unsafe { dfgpu_plan_memory(bptrs.as_ptr(), sizes.as_ptr(), nparts as i32, &mut plan) })?;
    Ok(Built { plan: GpuPlan(plan), _exprs: vec![], _children: vec![], _batches: batches, _arrays: arrays })
}

/// Translates the subtree rooted at `node`; children that are not device nodes become CPU leaves.
#[async_recursion::async_recursion]
async fn translate(ctx: &GpuCtx, node: &Arc<dyn ExecutionPlan>, task: &Arc<TaskContext>) -> Result<Built> {
    if !is_gpu_node(node.as_ref()) { return cpu_leaf(ctx, node, task).await; }
    let mut children = vec![];
    for c in node.children() { children.push(translate(ctx, &c, task).await?); }
    let (mut keep, mut arrays) = (vec![], vec![]);
    let mut out = null_mut();
    let any = node.as_any();
    let ch = |i: usize| children[i].plan.0 as *const dfgpu_plan;
    if let Some(f) = any.downcast_ref::<FilterExec>() {
        let p = translate_expr(ctx, f.predicate(), &mut keep, &mut arrays)?;
        check_exec(unsafe { dfgpu_plan_filter(p, ch(0), &mut out) })?;
    } else if let Some(p) = any.downcast_ref::<ProjectionExec>() {
        let names: Vec<_> = p.expr().iter().map(|(_, n)| cstring(n)).collect();
        let np: Vec<*const c_char> = names.iter().map(|n| n.as_ptr()).collect();
        let mut ex = vec![]; for (e, _) in p.expr() { ex.push(translate_expr(ctx, e, &mut keep, &mut arrays)?); }
        check_exec(unsafe { dfgpu_plan_projection(ex.as_ptr(), np.as_ptr(), ex.len() as i32, ch(0), &mut out) })?;
    } else if let Some(c) = any.downcast_ref::<CoalesceBatchesExec>() {
        check_exec(unsafe { dfgpu_plan_coalesce_batches(ch(0), c.target_batch_size() as i64, &mut out) })?;
    } else if any.is::<CoalescePartitionsExec>() {
        check_exec(unsafe { dfgpu_plan_coalesce_partitions(ch(0), &mut out) })?;
    } else if let Some(r) = any.downcast_ref::<RepartitionExec>() {
        match r.partitioning() {
            Partitioning::Hash(exprs, n) => {
                let mut ex = vec![]; for e in exprs { ex.push(translate_expr(ctx, e, &mut keep, &mut arrays)?); }
                check_exec(unsafe { dfgpu_plan_repartition(ch(0), ex.as_ptr(), ex.len() as i32, *n as i32, &mut out) })?;
            }
            Partitioning::RoundRobinBatch(n) => check_exec(unsafe { dfgpu_plan_repartition(ch(0), null(), 0, *n as i32, &mut out) })?,
            other => return not_impl_err!("Unsupported repartitioning scheme {other:?}"),
        }
    } else if let Some(j) = any.downcast_ref::<HashJoinExec>() {
        let (mut l, mut r) = (vec![], vec![]);
        for (a, b) in j.on() { l.push(translate_expr(ctx, a, &mut keep, &mut arrays)?); r.push(translate_expr(ctx, b, &mut keep, &mut arrays)?); }
        let (filter, sides, idx) = match j.filter() {
            None => (null(), vec![], vec![]),
            Some(f) => (translate_expr(ctx, f.expression(), &mut keep, &mut arrays)?,
                        f.column_indices().iter().map(|c| matches!(c.side, datafusion_common::JoinSide::Right) as i32).collect::<Vec<_>>(),
                        f.column_indices().iter().map(|c| c.index as i32).collect::<Vec<_>>()),
        };
        let jt = match j.join_type() { JoinType::Inner => 0, JoinType::Left => 1, JoinType::Right => 2, JoinType::Full => 3, JoinType::LeftSemi => 4, JoinType::RightSemi => 5, JoinType::LeftAnti => 6, JoinType::RightAnti => 7 };
        let mode = match j.partition_mode() { PartitionMode::CollectLeft => 0, PartitionMode::Partitioned => 1, PartitionMode::Auto => return internal_err!("PartitionMode::Auto reaches execution") };
        check_exec(unsafe { dfgpu_plan_hash_join(ch(0), ch(1), l.as_ptr(), r.as_ptr(), l.len() as i32, filter, sides.as_ptr(), idx.as_ptr(), idx.len() as i32, jt, mode, j.null_equals_null() as i32, &mut out) })?;
    } else if let Some(a) = any.downcast_ref::<AggregateExec>() {
        if !a.group_by().null_expr().is_empty() && a.group_by().groups().len() > 1 { return not_impl_err!("grouping sets through the shim (bind dfgpu_plan_aggregate_grouping_sets)"); }
        let mode = match a.mode() { AggregateMode::Partial => 0, AggregateMode::Final => 1, AggregateMode::FinalPartitioned => 2, AggregateMode::Single => 3, AggregateMode::SinglePartitioned => 4 };
        let gnames: Vec<_> = a.group_by().expr().iter().map(|(_, n)| cstring(n)).collect();
        let gn: Vec<*const c_char> = gnames.iter().map(|n| n.as_ptr()).collect();
        let mut ge = vec![]; for (e, _) in a.group_by().expr() { ge.push(translate_expr(ctx, e, &mut keep, &mut arrays)?); }
        let (mut kinds, mut args, mut filters, mut types, mut names) = (vec![], vec![], vec![], vec![], vec![]);
        for (agg, filt) in a.aggr_expr().iter().zip(a.filter_expr()) {
            let kind = match agg.name().split('(').next().unwrap_or("").to_ascii_uppercase().as_str() { "SUM" => AGG_SUM, "AVG" => AGG_AVG, "COUNT" => AGG_COUNT, "MIN" => AGG_MIN, "MAX" => AGG_MAX,
                                                                                                       other => return not_impl_err!("aggregate {other} on the device") };
            let ex = agg.expressions();
            let arg = match ex.first() { Some(e) if !(kind == AGG_COUNT && e.as_any().is::<Literal>()) => translate_expr(ctx, e, &mut keep, &mut arrays)?, _ => null() };   // COUNT(1) = COUNT(*)
            let it = match ex.first() { Some(e) => e.data_type(&a.input_schema())?, None => DataType::Int64 };
            let (t, p, s) = gpu_type(&it).ok_or_else(|| DataFusionError::NotImplemented(format!("aggregate over {it} on the device")))?;
            kinds.push(kind); args.push(arg); types.extend([t, p, s]); names.push(cstring(agg.name()));
            filters.push(match filt { Some(f) => translate_expr(ctx, f, &mut keep, &mut arrays)?, None => null() });
        }
        let an: Vec<*const c_char> = names.iter().map(|n| n.as_ptr()).collect();
        check_exec(unsafe { dfgpu_plan_aggregate(mode, ge.as_ptr(), gn.as_ptr(), ge.len() as i32, kinds.as_ptr(), args.as_ptr(), filters.as_ptr(), an.as_ptr(), types.as_ptr(), kinds.len() as i32, ch(0), &mut out) })?;
    } else if let Some(s) = any.downcast_ref::<SortExec>() {
        let (e, d, nf) = sort_args(ctx, s.expr(), &mut keep, &mut arrays)?;
        check_exec(unsafe { dfgpu_plan_sort(e.as_ptr(), d.as_ptr(), nf.as_ptr(), e.len() as i32, s.fetch().map_or(-1, |f| f as i64), s.preserve_partitioning() as i32, ch(0), &mut out) })?;
    } else if let Some(s) = any.downcast_ref::<SortPreservingMergeExec>() {
        let (e, d, nf) = sort_args(ctx, s.expr(), &mut keep, &mut arrays)?;
        check_exec(unsafe { dfgpu_plan_sort_preserving_merge(e.as_ptr(), d.as_ptr(), nf.as_ptr(), e.len() as i32, s.fetch().map_or(-1, |f| f as i64), ch(0), &mut out) })?;
    } else {
        return internal_err!("is_gpu_node and translate disagree on {}", node.name());
    }
    Ok(Built { plan: GpuPlan(out), _exprs: keep, _children: children, _batches: vec![], _arrays: arrays })
}

/// The `ExecutionPlan` the optimizer rule puts in place of a GPU-supported subtree.
#[derive(Debug)]
pub struct GpuSubplanExec { cpu: Arc<dyn ExecutionPlan>, device: i32 }
impl GpuSubplanExec {
    pub fn new(cpu: Arc<dyn ExecutionPlan>, device: i32) -> Self { Self { cpu, device } }
    /// the CPU subtree this node stands for (fallback, EXPLAIN)
    pub fn cpu_plan(&self) -> &Arc<dyn ExecutionPlan> { &self.cpu }
}
impl DisplayAs for GpuSubplanExec {
    fn fmt_as(&self, _t: DisplayFormatType, f: &mut fmt::Formatter) -> fmt::Result { write!(f, "GpuSubplanExec: device={}, root={}", self.device, self.cpu.name()) }
}
impl ExecutionPlan for GpuSubplanExec {
    fn as_any(&self) -> &dyn Any { self }
    fn properties(&self) -> &PlanProperties { self.cpu.properties() }
    fn children(&self) -> Vec<Arc<dyn ExecutionPlan>> { vec![self.cpu.clone()] }          // the optimizer keeps seeing (and may rewrite) the CPU subtree
    fn with_new_children(self: Arc<Self>, mut c: Vec<Arc<dyn ExecutionPlan>>) -> Result<Arc<dyn ExecutionPlan>> {
        Ok(Arc::new(GpuSubplanExec { cpu: c.pop().ok_or_else(|| DataFusionError::Internal("GpuSubplanExec wrong number of children".into()))?, device: self.device }))
    }
    fn required_input_distribution(&self) -> Vec<datafusion_physical_plan::Distribution> { vec![datafusion_physical_plan::Distribution::UnspecifiedDistribution] }
    fn maintains_input_order(&self) -> Vec<bool> { vec![true] }
    fn execute(&self, partition: usize, context: Arc<TaskContext>) -> Result<SendableRecordBatchStream> {
        // lazy and cheap (lib.rs:233-237): the device plan is built and started on the first poll
        let (cpu, device, schema, batch_size) = (self.cpu.clone(), self.device, self.cpu.schema(), context.session_config().batch_size() as i64);
        let fut = async move {
            let ctx = GpuCtx::new(device)?;
            let built = translate(&ctx, &cpu, &context).await?;
            let mut s = null_mut();
            check_exec(unsafe { dfgpu_plan_execute(built.plan.0, partition as i32, ctx.0, batch_size, &mut s) })?;
            Ok::<_, DataFusionError>(GpuSubplanStream { ctx, _built: built, stream: GpuStream(s), schema: schema.clone() })
        };
        let schema = self.cpu.schema();
        Ok(Box::pin(datafusion_physical_plan::stream::RecordBatchStreamAdapter::new(schema, futures::stream::once(fut).try_flatten())))
    }
}

/// `Stream<Item = Result<RecordBatch>>` over `dfgpu_stream_next`; dropping it frees the device stream, plan and context.
struct GpuSubplanStream { stream: GpuStream, _built: Built, ctx: Arc<GpuCtx>, schema: SchemaRef }
unsafe impl Send for GpuSubplanStream {}
impl Stream for GpuSubplanStream {
    type Item = Result<RecordBatch>;
    fn poll_next(self: Pin<&mut Self>, _cx: &mut Context<'_>) -> Poll<Option<Self::Item>> {
        let mut b = null_mut();
        if let Err(e) = check_exec(unsafe { dfgpu_stream_next(self.stream.0, &mut b) }) { return Poll::Ready(Some(Err(e))); }
        if b.is_null() { return Poll::Ready(None); }
        let batch = GpuBatch(b);
        let mut cols = vec![];
        for i in 0..unsafe { dfgpu_batch_num_columns(batch.0) } {
            let mut a = null_mut();
            if let Err(e) = check_exec(unsafe { dfgpu_batch_column(self.ctx.0, batch.0, i, &mut a) }) { return Poll::Ready(Some(Err(e))); }
            let a = GpuArray(a);
            match export_array(&self.ctx, a.0) { Ok(x) => cols.push(x), Err(e) => return Poll::Ready(Some(Err(e))) }
        }
        // dictionary-encoded string columns of the CPU schema come back as plain Utf8: cast to the declared field types
        let cols: Result<Vec<_>> = cols.into_iter().zip(self.schema.fields()).map(|(c, f)| if c.data_type() == f.data_type() { Ok(c) } else { Ok(arrow::compute::cast(&c, f.data_type())?) }).collect();
        Poll::Ready(Some(cols.and_then(|c| Ok(RecordBatch::try_new(self.schema.clone(), c)?))))
    }
}
impl RecordBatchStream for GpuSubplanStream { fn schema(&self) -> SchemaRef { self.schema.clone() } }
