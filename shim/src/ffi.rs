//! C ABI of libdfgpu.so (include/dfgpu.h, include/dfgpu_exec.h) as the shim uses it, plus RAII wrappers.
use std::ffi::{c_char, c_void, CStr, CString};
use std::ptr::null_mut;
use std::sync::Arc;

use arrow::array::{make_array, Array, ArrayRef};
use arrow::ffi::{from_ffi, to_ffi, FFI_ArrowArray, FFI_ArrowSchema};
use datafusion_common::{DataFusionError, Result};

macro_rules! opaque { ($($n:ident),*) => { $(#[repr(C)] pub struct $n { _p: [u8; 0] })* } }
opaque!(dfgpu_ctx, dfgpu_array, dfgpu_expr, dfgpu_plan, dfgpu_batch, dfgpu_stream, dfgpu_comm, dfgpu_parquet, dfgpu_join_table);

/// dfgpu_comm_vtable (include/dfgpu.h): a transport the host provides instead of RCCL
#[repr(C)]
pub struct dfgpu_comm_vtable {
    pub user: *mut c_void, pub rank: i32, pub world: i32,
    pub all_gather_host: Option<unsafe extern "C" fn(*mut c_void, *const c_void, i64, *mut c_void) -> i32>,
    pub all_to_all_v: Option<unsafe extern "C" fn(*mut c_void, *const c_void, *const i64, *const i64, *mut c_void, *const i64, *const i64) -> i32>,
}

// DFGPU_OP_* (include/dfgpu.h)
pub const OP_ADD: i32 = 0; pub const OP_SUB: i32 = 1; pub const OP_MUL: i32 = 2; pub const OP_DIV: i32 = 3; pub const OP_REM: i32 = 4;
pub const OP_EQ: i32 = 10; pub const OP_NEQ: i32 = 11; pub const OP_LT: i32 = 12; pub const OP_LTEQ: i32 = 13; pub const OP_GT: i32 = 14; pub const OP_GTEQ: i32 = 15;
pub const OP_DISTINCT: i32 = 16; pub const OP_NOT_DISTINCT: i32 = 17; pub const OP_AND: i32 = 20; pub const OP_OR: i32 = 21;
// DFGPU_AGG_*
pub const AGG_SUM: i32 = 0; pub const AGG_AVG: i32 = 1; pub const AGG_COUNT: i32 = 2; pub const AGG_MIN: i32 = 3; pub const AGG_MAX: i32 = 4;

extern "C" {
    // ---- include/dfgpu.h
    pub fn dfgpu_ctx_create(device_id: i32, stream: *mut c_void, out: *mut *mut dfgpu_ctx) -> i32;
    pub fn dfgpu_ctx_destroy(ctx: *mut dfgpu_ctx);
    pub fn dfgpu_last_error(ctx: *const dfgpu_ctx) -> *const c_char;
    pub fn dfgpu_ctx_set_option(ctx: *mut dfgpu_ctx, key: *const c_char, value: i64) -> i32;
    pub fn dfgpu_array_import_arrow(ctx: *mut dfgpu_ctx, a: *mut FFI_ArrowArray, s: *mut FFI_ArrowSchema, out: *mut *mut dfgpu_array) -> i32;
    pub fn dfgpu_array_export_arrow(ctx: *mut dfgpu_ctx, a: *const dfgpu_array, out_a: *mut FFI_ArrowArray, out_s: *mut FFI_ArrowSchema) -> i32;
    pub fn dfgpu_array_release(a: *mut dfgpu_array);
    // ---- round 2 (include/dfgpu.h): lent device memory, exchange between ranks, Parquet scan
    pub fn dfgpu_array_wrap_device_owned(ctx: *mut dfgpu_ctx, desc: *const c_void /* dfgpu_array_desc */, release: Option<unsafe extern "C" fn(*mut c_void)>, cookie: *mut c_void, out: *mut *mut dfgpu_array) -> i32;
    pub fn dfgpu_comm_unique_id(out_id128: *mut u8) -> i32;
    pub fn dfgpu_comm_create_rccl(ctx: *mut dfgpu_ctx, id128: *const u8, rank: i32, world: i32, out: *mut *mut dfgpu_comm) -> i32;
    pub fn dfgpu_comm_create_custom(vtable: *const dfgpu_comm_vtable, out: *mut *mut dfgpu_comm) -> i32;
    pub fn dfgpu_comm_free(comm: *mut dfgpu_comm);
    pub fn dfgpu_exchange(ctx: *mut dfgpu_ctx, comm: *mut dfgpu_comm, keys: *const *const dfgpu_array, nkeys: i32, cols: *const *const dfgpu_array, ncols: i32,
                          opt_mask: *const dfgpu_array, out_cols: *mut *mut dfgpu_array, out_counts: *mut i64) -> i32;
    pub fn dfgpu_parquet_open(ctx: *mut dfgpu_ctx, file_bytes: *const u8, len: i64, device_bytes: *const u8, out: *mut *mut dfgpu_parquet) -> i32;
    pub fn dfgpu_parquet_close(file: *mut dfgpu_parquet);
    pub fn dfgpu_parquet_num_row_groups(file: *const dfgpu_parquet) -> i32;
    pub fn dfgpu_parquet_column_stats(file: *const dfgpu_parquet, row_group: i32, column: i32, min_value: *mut i64, max_value: *mut i64, null_count: *mut i64, has_min_max: *mut i32) -> i32;
    pub fn dfgpu_parquet_read(ctx: *mut dfgpu_ctx, file: *mut dfgpu_parquet, first_row_group: i32, num_row_groups: i32, columns: *const i32, ncols: i32, out: *mut *mut dfgpu_array) -> i32;
    // ---- include/dfgpu_exec.h
    pub fn dfgpu_sort_to_indices_keys(ctx: *mut dfgpu_ctx, cols: *const *const dfgpu_array, descending: *const u8, nulls_first: *const u8, k: i32, fetch: i64, out: *mut *mut dfgpu_array, out_sorted: *mut *mut dfgpu_array) -> i32;
    /// sort_batch in one call: the batch's other columns go down as payload; entries of out_payload the sort's last pass could not gather stay null (dfgpu_take them).
    pub fn dfgpu_sort_take(ctx: *mut dfgpu_ctx, cols: *const *const dfgpu_array, descending: *const u8, nulls_first: *const u8, k: i32, fetch: i64, payload: *const *const dfgpu_array, n_payload: i32,
                           out: *mut *mut dfgpu_array, out_sorted: *mut *mut dfgpu_array, out_payload: *mut *mut dfgpu_array) -> i32;
    /// Partial stage inside the operator; flags: 1 = DFGPU_PREAGG_ANY_ORDER (set below a SortExec over all group columns); value_casts[i] = DFGPU_FLOAT64: argument i is CAST(values[i] AS DOUBLE).
    pub fn dfgpu_agg_preaggregate_flags(ctx: *mut dfgpu_ctx, keys: *const *const dfgpu_array, nkeys: i32, kinds: *const i32, values: *const *const dfgpu_array, value_casts: *const i32, n_aggs: i32,
                                        opt_mask: *const dfgpu_array, flags: i32, out_keys: *mut *mut dfgpu_array, out_states: *mut *mut dfgpu_array) -> i32;
    pub fn dfgpu_csv_read(ctx: *mut dfgpu_ctx, bytes: *const u8, len: i64, bytes_on_device: i32, delimiter: i32, quote: i32, escape: i32, has_header: i32, ncols_file: i32, columns: *const i32, types: *const i32, ncols: i32, out: *mut *mut dfgpu_array, out_rows: *mut i64) -> i32;
    pub fn dfgpu_plan_aggregate_input_order(aggregate: *mut dfgpu_plan, input_order_mode: i32, order_indices: *const i32, n: i32) -> i32;
    pub fn dfgpu_plan_parquet(file: *mut dfgpu_parquet, columns: *const i32, ncols: i32, npartitions: i32, row_groups_per_batch: i32, out: *mut *mut dfgpu_plan) -> i32;
    pub fn dfgpu_plan_csv(bytes: *const u8, len: i64, delimiter: i32, quote: i32, escape: i32, has_header: i32, names: *const *const c_char, types: *const i32, ncols_file: i32, columns: *const i32, ncols: i32, npartitions: i32, batch_bytes: i64, out: *mut *mut dfgpu_plan) -> i32;
    pub fn dfgpu_plan_parquet_prune(parquet_exec: *mut dfgpu_plan, column: i32, min_value: i64, max_value: i64) -> i32;
    // ---- round 3
    pub fn dfgpu_list_from_counts(ctx: *mut dfgpu_ctx, counts: *const dfgpu_array, values: *const dfgpu_array, out: *mut *mut dfgpu_array) -> i32;
    pub fn dfgpu_list_flatten(ctx: *mut dfgpu_ctx, list: *const dfgpu_array, value_type: i32, precision: i32, scale: i32, out_values: *mut *mut dfgpu_array, out_row_of: *mut *mut dfgpu_array) -> i32;
    pub fn dfgpu_join_probe_deferred(ctx: *mut dfgpu_ctx, table: *const dfgpu_join_table, probe_keys: *const *const dfgpu_array, nkeys: i32, opt_mask: *const dfgpu_array,
                                     out_build_idx: *mut *mut dfgpu_array, out_probe_idx: *mut *mut dfgpu_array) -> i32;
    pub fn dfgpu_join_probe_selection(ctx: *mut dfgpu_ctx, table: *const dfgpu_join_table, probe_keys: *const *const dfgpu_array, nkeys: i32, opt_mask: *const dfgpu_array, out_selection: *mut *mut dfgpu_array) -> i32;
    pub fn dfgpu_join_lookup(ctx: *mut dfgpu_ctx, table: *const dfgpu_join_table, probe_keys: *const *const dfgpu_array, nkeys: i32, rows: *const dfgpu_array, out_build_idx: *mut *mut dfgpu_array) -> i32;
    pub fn dfgpu_plan_sort_merge_join(left: *const dfgpu_plan, right: *const dfgpu_plan, on_l: *const *const dfgpu_expr, on_r: *const *const dfgpu_expr, non: i32, filter: *const dfgpu_expr,
                                      filter_sides: *const i32, filter_indices: *const i32, nfilter_cols: i32, join_type: i32, null_equals_null: i32, out: *mut *mut dfgpu_plan) -> i32;
    pub fn dfgpu_plan_nested_loop_join(left: *const dfgpu_plan, right: *const dfgpu_plan, filter: *const dfgpu_expr, filter_sides: *const i32, filter_indices: *const i32,
                                       nfilter_cols: i32, join_type: i32, out: *mut *mut dfgpu_plan) -> i32;
    pub fn dfgpu_plan_metrics(p: *const dfgpu_plan, buf: *mut c_char, capacity: i64) -> i32;
    pub fn dfgpu_exec_last_error() -> *const c_char;
    pub fn dfgpu_batch_new(names: *const *const c_char, columns: *const *const dfgpu_array, ncols: i32, out: *mut *mut dfgpu_batch) -> i32;
    pub fn dfgpu_batch_free(b: *mut dfgpu_batch);
    pub fn dfgpu_batch_num_columns(b: *const dfgpu_batch) -> i32;
    pub fn dfgpu_batch_column(ctx: *mut dfgpu_ctx, b: *mut dfgpu_batch, i: i32, out: *mut *mut dfgpu_array) -> i32;
    pub fn dfgpu_expr_column(name: *const c_char, index: i32, out: *mut *mut dfgpu_expr) -> i32;
    pub fn dfgpu_expr_literal(scalar_len1: *const dfgpu_array, out: *mut *mut dfgpu_expr) -> i32;
    pub fn dfgpu_expr_binary(l: *const dfgpu_expr, op: i32, r: *const dfgpu_expr, out: *mut *mut dfgpu_expr) -> i32;
    pub fn dfgpu_expr_not(e: *const dfgpu_expr, out: *mut *mut dfgpu_expr) -> i32;
    pub fn dfgpu_expr_is_null(e: *const dfgpu_expr, negated: i32, out: *mut *mut dfgpu_expr) -> i32;
    pub fn dfgpu_expr_negative(e: *const dfgpu_expr, out: *mut *mut dfgpu_expr) -> i32;
    pub fn dfgpu_expr_cast(e: *const dfgpu_expr, to_type: i32, precision: i32, scale: i32, out: *mut *mut dfgpu_expr) -> i32;
    pub fn dfgpu_expr_free(e: *mut dfgpu_expr);
    pub fn dfgpu_plan_memory(batches: *const *const dfgpu_batch, partition_sizes: *const i32, npartitions: i32, out: *mut *mut dfgpu_plan) -> i32;
    pub fn dfgpu_plan_filter(predicate: *const dfgpu_expr, input: *const dfgpu_plan, out: *mut *mut dfgpu_plan) -> i32;
    pub fn dfgpu_plan_projection(exprs: *const *const dfgpu_expr, names: *const *const c_char, n: i32, input: *const dfgpu_plan, out: *mut *mut dfgpu_plan) -> i32;
    pub fn dfgpu_plan_coalesce_batches(input: *const dfgpu_plan, target: i64, out: *mut *mut dfgpu_plan) -> i32;
    pub fn dfgpu_plan_coalesce_partitions(input: *const dfgpu_plan, out: *mut *mut dfgpu_plan) -> i32;
    pub fn dfgpu_plan_repartition(input: *const dfgpu_plan, exprs: *const *const dfgpu_expr, nexprs: i32, num_partitions: i32, out: *mut *mut dfgpu_plan) -> i32;
    pub fn dfgpu_plan_hash_join(left: *const dfgpu_plan, right: *const dfgpu_plan, on_l: *const *const dfgpu_expr, on_r: *const *const dfgpu_expr, non: i32,
                                filter: *const dfgpu_expr, filter_sides: *const i32, filter_indices: *const i32, nfilter_cols: i32,
                                join_type: i32, mode: i32, null_equals_null: i32, out: *mut *mut dfgpu_plan) -> i32;
    pub fn dfgpu_plan_aggregate(mode: i32, group_exprs: *const *const dfgpu_expr, group_names: *const *const c_char, ngroups: i32,
                                agg_kinds: *const i32, agg_args: *const *const dfgpu_expr, agg_filters: *const *const dfgpu_expr, agg_names: *const *const c_char,
                                agg_arg_types: *const i32, naggs: i32, input: *const dfgpu_plan, out: *mut *mut dfgpu_plan) -> i32;
    pub fn dfgpu_plan_sort(exprs: *const *const dfgpu_expr, descending: *const u8, nulls_first: *const u8, n: i32, fetch: i64, preserve_partitioning: i32,
                           input: *const dfgpu_plan, out: *mut *mut dfgpu_plan) -> i32;
    pub fn dfgpu_plan_sort_preserving_merge(exprs: *const *const dfgpu_expr, descending: *const u8, nulls_first: *const u8, n: i32, fetch: i64,
                                            input: *const dfgpu_plan, out: *mut *mut dfgpu_plan) -> i32;
    pub fn dfgpu_plan_free(p: *mut dfgpu_plan);
    pub fn dfgpu_plan_metrics(p: *const dfgpu_plan, buf: *mut c_char, capacity: i64) -> i32;
    pub fn dfgpu_plan_execute(p: *const dfgpu_plan, partition: i32, ctx: *mut dfgpu_ctx, batch_size: i64, out: *mut *mut dfgpu_stream) -> i32;
    pub fn dfgpu_stream_next(s: *mut dfgpu_stream, out: *mut *mut dfgpu_batch) -> i32;
    pub fn dfgpu_stream_free(s: *mut dfgpu_stream);
}

/// dfgpu_status -> DataFusionError (common/src/error.rs:52-122)
pub fn status_to_error(st: i32, msg: String) -> DataFusionError {
    match st {
        1 => DataFusionError::Execution(msg),
        3 => DataFusionError::ResourcesExhausted(msg),
        4 => DataFusionError::NotImplemented(msg),
        _ => DataFusionError::Internal(msg),
    }
}
pub fn check_ctx(ctx: *mut dfgpu_ctx, st: i32) -> Result<()> {
    if st == 0 { return Ok(()); }
    Err(status_to_error(st, unsafe { CStr::from_ptr(dfgpu_last_error(ctx)) }.to_string_lossy().into_owned()))
}
pub fn check_exec(st: i32) -> Result<()> {
    if st == 0 { return Ok(()); }
    Err(status_to_error(st, unsafe { CStr::from_ptr(dfgpu_exec_last_error()) }.to_string_lossy().into_owned()))
}

/// One device context per `execute(partition)` call: a device + a private HIP stream (include/dfgpu.h "context").
pub struct GpuCtx(pub *mut dfgpu_ctx);
unsafe impl Send for GpuCtx {}
unsafe impl Sync for GpuCtx {}
impl GpuCtx {
    pub fn new(device: i32) -> Result<Arc<Self>> {
        let mut c = null_mut();
        let st = unsafe { dfgpu_ctx_create(device, null_mut(), &mut c) };
        if st != 0 { return Err(status_to_error(st, format!("dfgpu_ctx_create(device {device}) failed: no usable gfx950 device"))); }
        Ok(Arc::new(GpuCtx(c)))
    }
}
impl Drop for GpuCtx { fn drop(&mut self) { unsafe { dfgpu_ctx_destroy(self.0) } } }

macro_rules! handle {
    ($name:ident, $raw:ident, $free:ident) => {
        /// Dropping the handle releases the device object: dropping a stream cancels and frees everything it holds
        /// (the cancellation contract of ExecutionPlan::execute, physical-plan/src/lib.rs:251-267).
        pub struct $name(pub *mut $raw);
        unsafe impl Send for $name {}
        unsafe impl Sync for $name {}
        impl Drop for $name { fn drop(&mut self) { if !self.0.is_null() { unsafe { $free(self.0) } } } }
    };
}
handle!(GpuArray, dfgpu_array, dfgpu_array_release);
handle!(GpuExpr, dfgpu_expr, dfgpu_expr_free);
handle!(GpuPlan, dfgpu_plan, dfgpu_plan_free);
handle!(GpuBatch, dfgpu_batch, dfgpu_batch_free);
handle!(GpuStream, dfgpu_stream, dfgpu_stream_free);

/// RecordBatch column -> HBM (Arrow C Data Interface; the library copies, the Arrow buffers stay the caller's)
pub fn import_array(ctx: &GpuCtx, a: &ArrayRef) -> Result<GpuArray> {
    let (mut fa, mut fs) = to_ffi(&a.to_data())?;
    let mut out = null_mut();
    check_ctx(ctx.0, unsafe { dfgpu_array_import_arrow(ctx.0, &mut fa, &mut fs, &mut out) })?;
    Ok(GpuArray(out))
}
/// HBM column -> Arrow array (host buffers owned through the Arrow release callback)
pub fn export_array(ctx: &GpuCtx, a: *const dfgpu_array) -> Result<ArrayRef> {
    let mut fa = FFI_ArrowArray::empty();
    let mut fs = FFI_ArrowSchema::empty();
    check_ctx(ctx.0, unsafe { dfgpu_array_export_arrow(ctx.0, a, &mut fa, &mut fs) })?;
    Ok(make_array(unsafe { from_ffi(fa, &fs) }?))
}
pub fn cstring(s: &str) -> CString { CString::new(s.replace('\0', "")).expect("no interior NUL") }
