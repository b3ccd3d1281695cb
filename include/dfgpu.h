/*
 * dfgpu.h -- C ABI of the MI355X (gfx950) execution path for DataFusion 36's data-parallel
 * physical operators.  This is the drop-in boundary: plain pointers and sizes, opaque handles,
 * int32 status codes.  A Rust `impl ExecutionPlan` shim binds exactly these
 * symbols; each one names the reference function it replaces (paths relative to
 * /root/reference/datafusion/).
 *
 * Conventions
 *  - Every call returns dfgpu_status; on failure dfgpu_last_error(ctx) holds the message.
 *    The codes mirror DataFusionError variants (common/src/error.rs:52-122):
 *    EXECUTION -> DataFusionError::Execution / ArrowError (e.g. "Divide by zero", decimal overflow),
 *    INTERNAL -> Internal, RESOURCES_EXHAUSTED -> ResourcesExhausted, NOT_IMPLEMENTED ->
 *    NotImplemented (the shim then keeps the CPU operator for that plan node).
 *  - dfgpu_array is an immutable, reference counted Arrow-layout column resident in HBM
 *    (Arrow columnar format: values buffer, LSB validity bitmap, int32 offsets for Utf8).
 *    Callee never frees caller arrays; every `out` array is owned by the caller and released with
 *    dfgpu_array_release (called from Rust Drop).
 *  - A dfgpu_ctx is bound to one device + one HIP stream; calls on one ctx are stream ordered
 *    and a ctx must be used by one thread at a time (ExecutionPlan::execute gives one ctx per
 *    output partition, physical-plan/src/lib.rs:378-382).
 *  - There is NO CPU fallback behind this ABI: if no HIP device is present ctx creation fails.
 */
#ifndef DFGPU_H
#define DFGPU_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFGPU_API __attribute__((visibility("default")))

typedef int32_t dfgpu_status;
enum { DFGPU_OK = 0, DFGPU_EXECUTION = 1, DFGPU_INTERNAL = 2, DFGPU_RESOURCES_EXHAUSTED = 3,
       DFGPU_NOT_IMPLEMENTED = 4, DFGPU_INVALID_ARGUMENT = 5 };

/* Arrow data types supported on device (arrow-schema DataType) */
enum { DFGPU_BOOL = 1, DFGPU_INT8 = 2, DFGPU_INT16 = 3, DFGPU_INT32 = 4, DFGPU_INT64 = 5,
       DFGPU_UINT8 = 6, DFGPU_UINT16 = 7, DFGPU_UINT32 = 8, DFGPU_UINT64 = 9,
       DFGPU_FLOAT32 = 10, DFGPU_FLOAT64 = 11, DFGPU_DATE32 = 12, DFGPU_DECIMAL128 = 13,
       DFGPU_UTF8 = 14, DFGPU_DICTIONARY = 15 };

/* Plain description of an Arrow column (host or device pointers, bit offset 0). */
typedef struct dfgpu_array_desc {
  int32_t type;
  int32_t precision;             /* DECIMAL128 */
  int32_t scale;                 /* DECIMAL128 */
  int32_t key_type;              /* DICTIONARY: integer type of the keys in `values` */
  int64_t length;
  int64_t null_count;            /* -1 = unknown */
  const void *values;            /* fixed width values | utf8 bytes | dictionary keys | bool bits */
  const uint8_t *validity;       /* LSB-first bitmap or NULL */
  const int32_t *offsets;        /* UTF8: length+1 */
  int64_t values_bytes;          /* UTF8: bytes in `values` */
  const struct dfgpu_array_desc *dictionary;  /* DICTIONARY: value column */
} dfgpu_array_desc;

typedef struct dfgpu_ctx dfgpu_ctx;
typedef struct dfgpu_array dfgpu_array;
typedef struct dfgpu_join_table dfgpu_join_table;
typedef struct dfgpu_groups dfgpu_groups;
typedef struct dfgpu_acc dfgpu_acc;

/* Arrow C Data Interface (https://arrow.apache.org/docs/format/CDataInterface.html); arrow-rs 50
 * `arrow::ffi::{FFI_ArrowArray, FFI_ArrowSchema}` and pyarrow `_export_to_c` are layout compatible. */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
  const char *format; const char *name; const char *metadata; int64_t flags; int64_t n_children;
  struct ArrowSchema **children; struct ArrowSchema *dictionary;
  void (*release)(struct ArrowSchema *); void *private_data;
};
struct ArrowArray {
  int64_t length; int64_t null_count; int64_t offset; int64_t n_buffers; int64_t n_children;
  const void **buffers; struct ArrowArray **children; struct ArrowArray *dictionary;
  void (*release)(struct ArrowArray *); void *private_data;
};
#endif

/* ------------------------------------------------------------------ context */
/* stream: a hipStream_t to enqueue on (e.g. the caller's / torch's current stream) or NULL to
 * create a private non-blocking stream.  Fails (DFGPU_EXECUTION) when no gfx950 device is usable. */
DFGPU_API dfgpu_status dfgpu_ctx_create(int32_t device_id, void *stream, dfgpu_ctx **out);
DFGPU_API void dfgpu_ctx_destroy(dfgpu_ctx *ctx);
DFGPU_API const char *dfgpu_last_error(const dfgpu_ctx *ctx);
DFGPU_API dfgpu_status dfgpu_ctx_synchronize(dfgpu_ctx *ctx);
/* options: "force_hash_collisions" (0/1) == cargo feature of common/src/hash_utils.rs:306-318;
 * "first_seen_group_order" (1/0) == group ids in first-seen order (group_values/primitive.rs:137-141);
 * "join_rank_index" (1/0) == let join_build replace the hash table by a bitmap rank index when the single integer key
 * column is strictly increasing (results identical either way; the switch exists for A/B tests);
 * "join_rank_index_unsorted" (1/0) == the same index for UNIQUE integer keys over a dense domain (<= 256 slots per key) in any order -- a primary-key column after a hash
 * repartition: one more array maps rank -> build row (results identical);
 * "join_key_packing" (1/0) == let join_build pack 2..4 integer key columns with small value ranges into one Int64 key (tuple
 * equality == packed equality), so that the single-key paths apply; results identical;
 * "group_run_detection" (1/0) == let groups_intern number groups by runs when a batch arrives clustered on its keys
 * (≙ GroupOrdering::Full, aggregates/order/full.rs; ids identical to the hash path: first-seen order);
 * "group_dictionary_canon" (1/0) == let groups_intern map dictionary key columns through a de-duplicated dictionary (u32 id of
 * the distinct VALUE per code) and intern those ids; groups, ids and emitted keys are identical to interning the values;
 * "join_swap_small_semi" (1/0) == let the plan layer's HashJoinExec index the RIGHT input of a LeftSemi / LeftAnti join when it is
 * at least 8x smaller than the collected left input (rows and row order identical: both are "left rows by ascending index");
 * "fused_aggregate_min_rows" (rows, default 2^20; negative = never) == smallest batch for which the plan layer's AggregateExec hands
 * accumulator argument expressions to dfgpu_acc_update_batch_fused instead of evaluating them node by node (results identical);
 * "join_partitioned" (1/0), "join_partitioned_min_build" / "join_partitioned_min_probe" (rows), "join_partition_rows" (build rows per partition, <= 14000) ==
 * radix-partitioned hash join (csrc/pjoin.hip): large builds on unsorted integer keys with a sparse domain are split by key hash so that
 * every partition's table sits in LDS; probe batches of >= min_probe rows are partitioned the same way.  Pairs and their order are identical;
 * "join_partitioned_big" (1/0) == builds of more than 2048 x join_partition_rows rows take up to 4096 partitions instead of declining to the global table (pairs identical);
 * "join_partitioned_hashed" (1/0) == large builds on key columns that mode does not take (several columns that do not pack, Utf8 / dictionary keys, null_equals_null) go through
 * the same path on 64-bit key hashes, every emitted pair verified in the columns; "join_partitioned_hash_mask" (0 = all bits; tests) == bits of the hash that are kept;
 * "agg_partitioned" (1/0), "agg_partitioned_min_rows", "agg_partitioned_force" (1 = skip the sample's verdict; tests) == let the plan layer's
 * AggregateExec pre-aggregate large batches of high-cardinality unclustered keys partition by partition out of LDS (dfgpu_agg_preaggregate);
 * "agg_preaggregate_distinct" (read only) == 1 when the last dfgpu_agg_preaggregate call on this ctx emitted every key in exactly one partial row;
 * "mailbox_readback" (1/0) == the host reads device words (counts, ranges, error flags) through a pinned mailbox -- a one-workgroup kernel posts them with a sequence number, the host
 * polls -- instead of a device-to-host copy followed by a stream synchronisation (same values; the switch exists for A/B runs);
 * "join_selection_output" (1/0) == let the plan layer's HashJoinExec answer an Inner join whose build side contributes key columns only with the probe batch under a selection
 * (dfgpu_join_probe_selection) when the operator above fuses selections (rows and row order identical); "join_lazy_build_rows" (1/0) == build rows of a deferred probe looked up on demand;
 * "join_bitmap_partitioned" (0/1, default 0), "join_bitmap_partitioned_min_rows" == probe a membership bitmap larger than an L2 by key range when a sample finds the probe keys
 * unclustered (pairs identical; measured slower than the plain probe on MI355X, kept for A/B);
 * "group_lazy_keys" (1/0) == a run-numbered first batch keeps its group keys as (key columns, first rows) until somebody needs them stored (dfgpu_groups_emit_deferred);
 * "agg_order_inverse_map" (1/0) == first-seen order of millions of pre-aggregated partial rows through an inverse map over the input rows instead of sort passes (same order);
 * "sort_fused_small_passes" (1/0) == sorts below 2^20 rows fold every pass's offset scan into its scatter (identical indices);
 * "sort_packed_keys" (1/0) == let sort_to_indices sort large inputs over fixed-width keys through range-packed 64-bit keys (identical indices);
 * "sort_estimate_ranges" (1/0) == from 2^22 rows on, take those ranges from a sample and check them while packing (a miss repeats the step with exact ranges; identical indices);
 * "sort_onesweep_rows" (16 / 8 / 0, default 16), "sort_onesweep_min_rows" (default 2^20), "sort_onesweep_fused_finish" (1/0) == packed-key sorts whose key and row number share
 * one 64-bit word run every LSD pass as ONE launch (tile offsets by look-back over the digit counts the tiles in front have published; histograms counted while the words are
 * encoded; tiles of 16 x 512 rows from 2 M rows on, else 8 x 512; 0 = the three-launch passes), the last pass writing row numbers and rebuilt key columns itself (identical indices);
 * "sort_topk_words_min_rows" (default 2^23) == a sort with fetch <= n / 16 over at least this many rows whose keys pack selects on the packed keys (radix select, then the
 * few candidates sorted) instead of on byte planes (identical indices); "sort_one_block_max_rows" (default 8192, 0 = off) == byte-plane sorts of at most this many rows run
 * every pass inside one launch of one workgroup (identical indices); "sort_payload_in_last_pass" (0/1, default 0) == dfgpu_sort_take gathers payload columns in the sort's last
 * pass (measured slower than the gather afterwards; kept for A/B);
 * "partition_two_round_staging" (0/1, default 0) == the radix partition into 513 .. 2048 partitions stages every column in two rounds of half a tile, two workgroups per CU
 * (same rows in the same partitions; measured slower, kept for A/B);
 * "memory_limit" (bytes, 0 = none) == live device memory this ctx may hold; an allocation beyond it fails with DFGPU_RESOURCES_EXHAUSTED and the
 * message of MemoryPool::try_grow (≙ RuntimeConfig::with_memory_limit, execution/src/runtime_env.rs); "live_bytes" / "cached_bytes" (read only);
 * "agg_spill_state_bytes" (0 = never) == the state size (group table + accumulators) above which AggregateExec spills to host memory (non-Partial modes,
 * row_hash.rs:667-705) or emits early (Partial, :720-733) -- the reservation a MemoryPool would grant the operator; "agg_spill_ranges" (16) == key ranges a spill is cut into;
 * "sort_spill_bytes" (0 = never) == bytes of input SortExec keeps on the device before it sorts them and spills the sorted run to host memory (ExternalSorter,
 * sorts/sort.rs:283-313; not with a fetch); "sort_spill_ranges" (16) == key ranges a run is cut into -- the merge brings one range of every run back at a time;
 * "spm_merge_rows" (2^25) == rows SortPreservingMergeExec loads over all its inputs per merge step (an input that ends inside its share needs no further step);
 * "collect_metrics" (1/0) == the plan layer records per-operator metrics (dfgpu_plan_metrics);
 * "defer_flag_checks" (1 = enter / 0 = leave a deferred region, nests) == kernel error flags (overflow, divide by zero,
 * cast range, index bounds -- the ArrowError cases of arrow-arith / arrow-cast / arrow-select) are normally checked by the
 * call that ran the kernel; inside a region they are checked once, by the call that leaves it (which returns the error),
 * and always before array_export_host / ctx_synchronize return.  dfgpu_stream_next polls inside one region. */
DFGPU_API dfgpu_status dfgpu_ctx_set_option(dfgpu_ctx *ctx, const char *key, int64_t value);
DFGPU_API dfgpu_status dfgpu_ctx_get_option(dfgpu_ctx *ctx, const char *key, int64_t *out_value);
DFGPU_API void *dfgpu_ctx_stream(dfgpu_ctx *ctx);
/* Selection-vector evaluation (the reference evaluates expressions on compacted batches, filter.rs:315-327 then
 * projection.rs:295-317; here a dense selection is carried instead of compacting): while a row selection (Boolean array; NULL
 * counts as false) is set, dfgpu_binary / dfgpu_cast over full-length columns of the same length still compute every row but
 * raise ArrowError conditions (overflow, divide by zero, cast range) only for selected rows; values of unselected rows are
 * unspecified and must stay behind the selection.  NULL clears it.  dfgpu_mask_count = number of true bits. */
DFGPU_API dfgpu_status dfgpu_ctx_set_row_selection(dfgpu_ctx *ctx, const dfgpu_array *mask);
DFGPU_API dfgpu_status dfgpu_mask_count(dfgpu_ctx *ctx, const dfgpu_array *mask, int64_t *out);
DFGPU_API const char *dfgpu_version(void);
/* Per-kernel device time measured with HIP events on the ctx stream (used by bench.py for the roofline
 * figure; ≙ the BaselineMetrics elapsed_compute timers of physical-plan/src/metrics/baseline.rs:47-56).
 * dfgpu_profile_read writes lines "kernel_name launches total_ms\n" into buf, then clears the records.
 * dfgpu_profile_select restricts the timers to one kernel name (NULL = all): an event pair costs ~10 us of stream
 * bubble, so a timed region is measured with only its dominant kernel bracketed. */
/* Device-time spans on the ctx stream (one HIP event pair each): what the plan layer fills elapsed_compute / build_time / join_time /
 * repartition_time with (≙ metrics::Time of BaselineMetrics / BuildProbeJoinMetrics / RepartitionMetrics, physical-plan/src/metrics/baseline.rs:47,
 * joins/utils.rs:1368, repartition/mod.rs:312-349 -- measured on the device instead of on the host thread).  span_elapsed_ns waits for the
 * span's end, returns its device time and frees it. */
DFGPU_API dfgpu_status dfgpu_span_begin(dfgpu_ctx *ctx, int64_t *out_span);
DFGPU_API dfgpu_status dfgpu_span_end(dfgpu_ctx *ctx, int64_t span);
DFGPU_API dfgpu_status dfgpu_span_elapsed_ns(dfgpu_ctx *ctx, int64_t span, int64_t *out_ns);
DFGPU_API dfgpu_status dfgpu_profile_enable(dfgpu_ctx *ctx, int32_t on);
DFGPU_API dfgpu_status dfgpu_profile_select(dfgpu_ctx *ctx, const char *kernel_name);
DFGPU_API dfgpu_status dfgpu_profile_read(dfgpu_ctx *ctx, char *buf, int64_t capacity);

/* ------------------------------------------------------------------ arrays */
/* Copy a host column to HBM (PCIe).  ≙ a RecordBatch column entering the GPU operator. */
DFGPU_API dfgpu_status dfgpu_array_import_host(dfgpu_ctx *ctx, const dfgpu_array_desc *host, dfgpu_array **out);
/* Wrap device memory owned by the caller (zero copy; caller keeps it alive until release).  The wrapped memory must not change while the array (or any array derived from it) lives: arrays are immutable, and
 * statistics measured on one (null count, sortedness of a key column) are kept with it. */
DFGPU_API dfgpu_status dfgpu_array_wrap_device(dfgpu_ctx *ctx, const dfgpu_array_desc *dev, dfgpu_array **out);
/* The same with ownership handed over (the Arrow C Data Interface's release callback, ArrowArray::release): the library calls release(cookie)
 * exactly once, when the last array, slice or plan node that refers to the memory is gone -- also when this call fails.  What a Rust shim
 * passes is a leaked Arc<Buffer> (cookie) and a function that drops it; operators below the C boundary (the C++ plan layer keeps raw device
 * pointers inside MemoryExec / HashJoinExec build sides) then cannot outlive the memory they read. */
DFGPU_API dfgpu_status dfgpu_array_wrap_device_owned(dfgpu_ctx *ctx, const dfgpu_array_desc *dev, void (*release)(void *cookie), void *cookie, dfgpu_array **out);
/* Device pointers + metadata of an array (for RCCL / torch interop). `out->dictionary` points
 * into storage owned by the array. */
DFGPU_API dfgpu_status dfgpu_array_describe(const dfgpu_array *a, dfgpu_array_desc *out);
/* Copy to caller-allocated host buffers sized from dfgpu_array_describe (values / validity /
 * offsets may be NULL to skip).  Synchronises the ctx stream. */
DFGPU_API dfgpu_status dfgpu_array_export_host(dfgpu_ctx *ctx, const dfgpu_array *a, void *values, uint8_t *validity, int32_t *offsets);
DFGPU_API dfgpu_status dfgpu_array_import_arrow(dfgpu_ctx *ctx, struct ArrowArray *array, struct ArrowSchema *schema, dfgpu_array **out);
DFGPU_API dfgpu_status dfgpu_array_export_arrow(dfgpu_ctx *ctx, const dfgpu_array *a, struct ArrowArray *out_array, struct ArrowSchema *out_schema);
DFGPU_API void dfgpu_array_retain(dfgpu_array *a);
DFGPU_API void dfgpu_array_release(dfgpu_array *a);
DFGPU_API int64_t dfgpu_array_length(const dfgpu_array *a);
/* 1 when the array is an index array KNOWN to be 0, 1, .., length - 1 (dfgpu_mask_to_indices over a mask that keeps every row; the probe
 * indices of dfgpu_join_probe when every probe row matched exactly once): dfgpu_take through it returns the values array itself, and
 * a caller composing gathers can skip it.  0 says nothing (the array may still happen to be the identity). */
DFGPU_API int32_t dfgpu_array_is_identity(const dfgpu_array *a);
/* UInt32 0, 1, .., length - 1 (flagged as the identity): group ids of rows that are one group each, in order */
DFGPU_API dfgpu_status dfgpu_array_iota(dfgpu_ctx *ctx, int64_t length, dfgpu_array **out);
DFGPU_API int64_t dfgpu_array_null_count(dfgpu_ctx *ctx, const dfgpu_array *a);   /* computes if unknown */
/* RecordBatch::slice: zero copy when offset % 64 == 0 (batch_size 8192 chunks), otherwise a copy. */
DFGPU_API dfgpu_status dfgpu_array_slice(dfgpu_ctx *ctx, const dfgpu_array *a, int64_t offset, int64_t length, dfgpu_array **out);
/* concat_batches per column (hash_join.rs:764, coalesce_batches.rs:198-260, sorts/sort.rs:505). */
DFGPU_API dfgpu_status dfgpu_concat(dfgpu_ctx *ctx, const dfgpu_array *const *arrays, int32_t n, dfgpu_array **out);
/* Lists of fixed-width values (the List<T> state column of COUNT(DISTINCT), physical-expr/src/aggregate/count_distinct/native.rs:state()) travel in the Utf8 layout:
 * offsets are BYTE offsets into the packed values, so every operation that moves a Utf8 column (take, filter, concat, slice, partition, exchange, Arrow export) moves a
 * list column; an Arrow ListArray is the same buffers with the offsets divided by the value width.  dfgpu_list_from_counts: row i gets the next counts[i] (Int64) values;
 * dfgpu_list_flatten: the values of all rows back to back (typed value_type) and, per value, the row it belongs to (UInt32) -- what merging such a state needs.
 * Lists of STRINGS (values / value_type Utf8: COUNT(DISTINCT) over a Utf8 argument, count_distinct/bytes.rs:47-75): a row holds its strings back to back, each as a 4-byte
 * little-endian length followed by its bytes; flatten gives the strings back as a Utf8 array.  List values carry no NULLs. */
DFGPU_API dfgpu_status dfgpu_list_from_counts(dfgpu_ctx *ctx, const dfgpu_array *counts, const dfgpu_array *values, dfgpu_array **out);
DFGPU_API dfgpu_status dfgpu_list_flatten(dfgpu_ctx *ctx, const dfgpu_array *list, int32_t value_type, int32_t precision, int32_t scale, dfgpu_array **out_values, dfgpu_array **out_row_of);
/* fixed-width column of `length` zeros, no validity (e.g. the single group id of an AggregateExec without GROUP BY) */
DFGPU_API dfgpu_status dfgpu_array_new_zeros(dfgpu_ctx *ctx, int32_t type, int32_t precision, int32_t scale, int64_t length, dfgpu_array **out);
/* DictionaryArray::try_new(keys, values) without copying: an integer array of codes (its validity = NULL codes) over a values array.
 * A gather take(values, keys) that is never materialised is exactly this array -- group-by and comparison kernels then work on the
 * codes (a join output column taken from a small dimension table: GROUP BY n_name). */
DFGPU_API dfgpu_status dfgpu_array_make_dictionary(dfgpu_ctx *ctx, const dfgpu_array *keys, const dfgpu_array *values, dfgpu_array **out);
/* new_null_array (joins/utils.rs:1214) */
DFGPU_API dfgpu_status dfgpu_array_new_null(dfgpu_ctx *ctx, int32_t type, int32_t precision, int32_t scale, int64_t length, dfgpu_array **out);

/* ------------------------------------------------------------------ a1: create_hashes */
/* ≙ create_hashes (common/src/hash_utils.rs:357-417): per-row u64 of k key columns, first column
 * assigns, later columns combine_hashes (:38-41); NULL leaves the running hash (0) unchanged;
 * dictionaries hash their values (:182-213).  `seed`: 0 for join/repartition (fixed seeds,
 * joins/hash_join.rs:329, repartition/mod.rs:115).  out: UINT64 array. */
DFGPU_API dfgpu_status dfgpu_hash_columns(dfgpu_ctx *ctx, const dfgpu_array *const *cols, int32_t k, uint64_t seed, dfgpu_array **out);

/* ------------------------------------------------------------------ arrow-select */
/* ≙ arrow::compute::take (joins/utils.rs:1216,1224; sorts/sort.rs:605; repartition/mod.rs:202).
 * indices: UINT32 / UINT64 / INT32 / INT64 array; a NULL index yields a NULL row. */
DFGPU_API dfgpu_status dfgpu_take(dfgpu_ctx *ctx, const dfgpu_array *values, const dfgpu_array *indices, dfgpu_array **out);
/* `take` of n columns through ONE index array (build_batch_from_indices / sort_batch gather every output column through the same indices,
 * joins/utils.rs:1180-1230, sorts/sort.rs:605): out[c] = take(values[c], indices); values[c] may be NULL (out[c] = NULL).  Same results as n
 * dfgpu_take calls; fixed-width columns without NULLs travel together as row-major records when many rows are gathered from a large source
 * (one memory sector per gathered row instead of one per row and column). */
DFGPU_API dfgpu_status dfgpu_take_multi(dfgpu_ctx *ctx, const dfgpu_array *const *values, int32_t n, const dfgpu_array *indices, dfgpu_array **out);
/* ≙ arrow::compute::filter with a BooleanArray mask (filter.rs:325): keeps rows whose mask is
 * valid AND true, input order preserved. */
DFGPU_API dfgpu_status dfgpu_filter(dfgpu_ctx *ctx, const dfgpu_array *values, const dfgpu_array *mask, dfgpu_array **out);
/* Selection vector of a mask (ascending UINT32 row numbers) -- FilterBuilder::optimize analogue. */
DFGPU_API dfgpu_status dfgpu_mask_to_indices(dfgpu_ctx *ctx, const dfgpu_array *mask, dfgpu_array **out);

/* ------------------------------------------------------------------ a12: PhysicalExpr kernels */
enum { DFGPU_OP_ADD = 0, DFGPU_OP_SUB = 1, DFGPU_OP_MUL = 2, DFGPU_OP_DIV = 3, DFGPU_OP_REM = 4,
       DFGPU_OP_EQ = 10, DFGPU_OP_NEQ = 11, DFGPU_OP_LT = 12, DFGPU_OP_LTEQ = 13, DFGPU_OP_GT = 14,
       DFGPU_OP_GTEQ = 15, DFGPU_OP_DISTINCT = 16, DFGPU_OP_NOT_DISTINCT = 17,
       DFGPU_OP_AND = 20, DFGPU_OP_OR = 21 };
/* ≙ BinaryExpr::evaluate (physical-expr/src/expressions/binary.rs:259-315) -> datum::apply /
 * apply_cmp (datum.rs:28-58) -> and_kleene / or_kleene (binary.rs:563-586).  A scalar operand is a
 * length-1 array with *_is_scalar = 1 (arrow Datum / ColumnarValue::Scalar). */
DFGPU_API dfgpu_status dfgpu_binary(dfgpu_ctx *ctx, int32_t op, const dfgpu_array *lhs, int32_t lhs_is_scalar,
                                    const dfgpu_array *rhs, int32_t rhs_is_scalar, dfgpu_array **out);
/* x op_outer (scalar op_inner y)  [or with either pair swapped: scalar_on_left / inner_on_left] in one pass over x and y: the value
 * and result type of dfgpu_binary(op_outer, x, dfgpu_binary(op_inner, scalar, y)) without the intermediate column (TPC-H's
 * l_extendedprice * (1 - l_discount)).  Decimal128 or Float64 operands without NULLs; any other shape returns DFGPU_NOT_IMPLEMENTED
 * and the caller evaluates the two nodes separately (BinaryExpr::evaluate, physical-expr/src/expressions/binary.rs:259-315). */
DFGPU_API dfgpu_status dfgpu_binary_fused2(dfgpu_ctx *ctx, int32_t op_outer, const dfgpu_array *x, int32_t op_inner, const dfgpu_array *scalar, const dfgpu_array *y,
                                           int32_t scalar_on_left, int32_t inner_on_left, dfgpu_array **out);
DFGPU_API dfgpu_status dfgpu_not(dfgpu_ctx *ctx, const dfgpu_array *a, dfgpu_array **out);                 /* not.rs:71 */
DFGPU_API dfgpu_status dfgpu_is_null(dfgpu_ctx *ctx, const dfgpu_array *a, int32_t negate, dfgpu_array **out); /* is_null.rs:74 / is_not_null.rs */
DFGPU_API dfgpu_status dfgpu_negative(dfgpu_ctx *ctx, const dfgpu_array *a, dfgpu_array **out);            /* negative.rs:79 */
DFGPU_API dfgpu_status dfgpu_cast(dfgpu_ctx *ctx, const dfgpu_array *a, int32_t to_type, int32_t precision, int32_t scale, dfgpu_array **out); /* cast.rs:121 */
DFGPU_API dfgpu_status dfgpu_in_list(dfgpu_ctx *ctx, const dfgpu_array *a, const dfgpu_array *list, int32_t negated, dfgpu_array **out);      /* in_list.rs:349 */

/* ------------------------------------------------------------------ a2-a6: HashJoinExec */
enum { DFGPU_JOIN_INNER = 0, DFGPU_JOIN_LEFT = 1, DFGPU_JOIN_RIGHT = 2, DFGPU_JOIN_FULL = 3,
       DFGPU_JOIN_LEFT_SEMI = 4, DFGPU_JOIN_RIGHT_SEMI = 5, DFGPU_JOIN_LEFT_ANTI = 6,
       DFGPU_JOIN_RIGHT_ANTI = 7 };
/* ≙ collect_left_input + update_hash + JoinHashMap (joins/hash_join.rs:678-815, joins/utils.rs:121-229).
 * keys: the build key columns of the concatenated build batch, rows in ORIGINAL input order (the
 * index space of the returned build indices; dfgpu_join_final_indices documents how the shim
 * restores the reference's reversed-concat order).  opt_mask: optional BOOL array, rows not
 * selected are not inserted (fused upstream FilterExec). */
DFGPU_API dfgpu_status dfgpu_join_build(dfgpu_ctx *ctx, const dfgpu_array *const *keys, int32_t nkeys,
                                        const dfgpu_array *opt_mask, int32_t null_equals_null, dfgpu_join_table **out);
DFGPU_API void dfgpu_join_table_free(dfgpu_join_table *t);
DFGPU_API int64_t dfgpu_join_table_num_rows(const dfgpu_join_table *t);
DFGPU_API int64_t dfgpu_join_table_memory(const dfgpu_join_table *t);   /* bytes, for MemoryReservation::try_grow */
/* ≙ lookup_join_hashmap = get_matched_indices_with_limit_offset (limit = whole batch) +
 * equal_rows_arr (joins/hash_join.rs:1024-1118): all (build, probe) row pairs with equal keys,
 * ordered by probe row, then by build input order (hash_join.rs:161-197, asserted :1593-1594).
 * out_build_idx: UINT64, out_probe_idx: UINT32 (both without nulls). */
DFGPU_API dfgpu_status dfgpu_join_probe(dfgpu_ctx *ctx, const dfgpu_join_table *t, const dfgpu_array *const *probe_keys,
                                        int32_t nkeys, const dfgpu_array *opt_mask,
                                        dfgpu_array **out_build_idx, dfgpu_array **out_probe_idx);
/* dfgpu_join_probe whose build indices may be left for later: when the table locates a build row from the key alone (a unique, rank-indexed build: one integer key column, the
 * probe column of the same type without NULLs) *out_build_idx comes back NULL and only the matched probe rows are returned; dfgpu_join_lookup then gives the build rows for any
 * subset of them -- `rows` = UInt32 probe rows known to match (NULL = every row of `probe_keys`) -- so a consumer that keeps few of the join's rows (a later semi join, a
 * selective filter) never pays for the rest.  Any other table answers exactly as dfgpu_join_probe does (*out_build_idx set). */
DFGPU_API dfgpu_status dfgpu_join_probe_deferred(dfgpu_ctx *ctx, const dfgpu_join_table *table, const dfgpu_array *const *probe_keys, int32_t nkeys,
                                                 const dfgpu_array *opt_mask, dfgpu_array **out_build_idx, dfgpu_array **out_probe_idx);
/* The probe of an Inner join whose build rows nobody asks for, as a SELECTION over the probe batch: *out_selection = Boolean column, bit i = probe row i is selected by
 * opt_mask and finds its key in the build.  Only for a table dfgpu_join_probe_deferred would defer (unique keys: a set bit is exactly one output row, in probe order);
 * any other table answers DFGPU_NOT_IMPLEMENTED before doing any work and the caller probes the ordinary way.  The join's output is then the probe batch under that
 * selection -- no index vector is built and nothing is read back (≙ build_batch_from_indices with probe indices = the set bits, joins/utils.rs:1180-1230). */
DFGPU_API dfgpu_status dfgpu_join_probe_selection(dfgpu_ctx *ctx, const dfgpu_join_table *table, const dfgpu_array *const *probe_keys, int32_t nkeys,
                                                  const dfgpu_array *opt_mask, dfgpu_array **out_selection);
DFGPU_API dfgpu_status dfgpu_join_lookup(dfgpu_ctx *ctx, const dfgpu_join_table *table, const dfgpu_array *const *probe_keys, int32_t nkeys, const dfgpu_array *rows,
                                         dfgpu_array **out_build_idx);
/* ≙ visited_left_side.set_bit for every joined build index (hash_join.rs:1274-1278). */
DFGPU_API dfgpu_status dfgpu_join_mark_visited(dfgpu_ctx *ctx, dfgpu_join_table *t, const dfgpu_array *build_idx);
/* ≙ adjust_indices_by_join_type over the alignment range [range_start, range_end)
 * (joins/utils.rs:1234-1364; hash_join.rs:1297-1316). */
DFGPU_API dfgpu_status dfgpu_join_adjust_indices(dfgpu_ctx *ctx, const dfgpu_array *build_idx, const dfgpu_array *probe_idx,
                                                 int64_t range_start, int64_t range_end, int32_t join_type,
                                                 dfgpu_array **out_build_idx, dfgpu_array **out_probe_idx);
/* ≙ build_join_indices of NestedLoopJoinExec (joins/nested_loop_join.rs:405-432) for a run of left rows: the candidate pairs of left rows
 * [first_left, first_left + count_left) with every right row [0, n_right), left-major.  The side that is collected (the inner table) gets UInt64
 * indices, the streamed side UInt32, like the build / probe indices of dfgpu_join_probe: left_is_u64 = 1 when the left input is the collected side
 * (Right / RightSemi / RightAnti / Full, nested_loop_join.rs:373-378).  At most 2^31 pairs per call. */
DFGPU_API dfgpu_status dfgpu_cross_join_indices(dfgpu_ctx *ctx, int64_t first_left, int64_t count_left, int64_t n_right, int32_t left_is_u64,
                                                dfgpu_array **out_left, dfgpu_array **out_right);
/* ≙ get_final_indices_from_bit_map (joins/utils.rs:1119-1141): ascending build indices that are
 * unmatched (Left/Full/LeftAnti) or matched (LeftSemi); probe side is all NULL. */
DFGPU_API dfgpu_status dfgpu_join_final_indices(dfgpu_ctx *ctx, const dfgpu_join_table *t, int32_t join_type, dfgpu_array **out_build_idx);

/* ------------------------------------------------------------------ a8: GroupValues */
/* ≙ new_group_values (aggregates/group_values/mod.rs:55-85); one implementation covers
 * GroupValuesPrimitive / GroupValuesRows / GroupValuesByes. */
DFGPU_API dfgpu_status dfgpu_groups_new(dfgpu_ctx *ctx, int32_t nkeys, dfgpu_groups **out);
DFGPU_API void dfgpu_groups_free(dfgpu_groups *g);
/* ≙ GroupValues::intern (primitive.rs:112-149, row.rs:94-146, bytes.rs:44-73): out_group_ids is a
 * UINT32 array, ids dense and (option first_seen_group_order) assigned in first-seen order; NULL keys
 * form their own group.  opt_mask: rows not selected get id 0xFFFFFFFF and are skipped by accumulators. */
DFGPU_API dfgpu_status dfgpu_groups_intern(dfgpu_ctx *ctx, dfgpu_groups *g, const dfgpu_array *const *cols, int32_t nkeys,
                                           const dfgpu_array *opt_mask, dfgpu_array **out_group_ids);
/* Same call, but the ids may come back DEFERRED: when every key column is dictionary-encoded over a small composite domain (the dense
 * map of dfgpu_groups_intern) the ids are a pure function of the code columns, and writing 4 B per row only for the accumulators to
 * read them back is the largest avoidable transfer of a TPC-H Q1 style aggregate.  A deferred id array is valid ONLY as the group_ids
 * argument of dfgpu_acc_update_batch / _multi / _fused / dfgpu_acc_merge_batch and of dfgpu_array_export_host (which write the ids
 * out on demand; _fused computes them inside its pass); groups, ids and emitted keys are identical to dfgpu_groups_intern. */
DFGPU_API dfgpu_status dfgpu_groups_intern_deferred(dfgpu_ctx *ctx, dfgpu_groups *g, const dfgpu_array *const *key_columns, int32_t nkeys,
                                                    const dfgpu_array *opt_mask, dfgpu_array **out_group_ids);
DFGPU_API int64_t dfgpu_groups_len(const dfgpu_groups *g);                   /* GroupValues::len */
DFGPU_API int64_t dfgpu_groups_size(const dfgpu_groups *g);                  /* GroupValues::size (bytes) */
/* ≙ GroupValues::emit(EmitTo::All) (primitive.rs:163-209): key columns in group id order. */
DFGPU_API dfgpu_status dfgpu_groups_emit(dfgpu_ctx *ctx, dfgpu_groups *g, dfgpu_array **out_cols /* nkeys */);
/* GroupValues::emit(EmitTo::All) for a caller that gathers lazily: when the stored keys are still "key column c at row out_rows[g]" -- the first batch of a clustered input,
 * whose groups are run numbers -- the columns (retained) and the UInt32 first rows come back instead of gathered key columns: group g's key c is out_sources[c][out_rows[g]].
 * DFGPU_NOT_IMPLEMENTED when the keys are stored already (then dfgpu_groups_emit).  The table keeps its state; a later call that needs stored keys gathers them.
 * (TPC-H Q18: HAVING keeps a few thousand of 150 M groups; only their keys are ever read.) */
DFGPU_API dfgpu_status dfgpu_groups_emit_deferred(dfgpu_ctx *ctx, dfgpu_groups *g, dfgpu_array **out_sources, dfgpu_array **out_rows);
/* ≙ GroupValues::emit(EmitTo::First(n)) (expr/src/groups_accumulator.rs:25-57, group_values/row.rs:176-212, primitive.rs:151-190): the keys of the first n groups
 * leave, the remaining groups are renumbered from 0 (group id g becomes g - n); ids handed out earlier refer to the old numbering.  n >= len emits everything. */
DFGPU_API dfgpu_status dfgpu_groups_emit_first(dfgpu_ctx *ctx, dfgpu_groups *g, int64_t n, dfgpu_array **out_cols /* nkeys */);

/* ------------------------------------------------------------------ a9: GroupsAccumulator */
enum { DFGPU_AGG_SUM = 0, DFGPU_AGG_AVG = 1, DFGPU_AGG_COUNT = 2, DFGPU_AGG_MIN = 3, DFGPU_AGG_MAX = 4 };
/* ≙ AggregateExpr::create_groups_accumulator (sum.rs, average.rs, count.rs, min_max.rs).
 * Input dtype rules as the reference: SUM over Int64/UInt64/Float64/Decimal128 (sum.rs:75-86),
 * AVG over Float64/Decimal128 (average.rs), MIN/MAX over primitive types. */
DFGPU_API dfgpu_status dfgpu_acc_new(dfgpu_ctx *ctx, int32_t kind, int32_t in_type, int32_t in_precision, int32_t in_scale, dfgpu_acc **out);
DFGPU_API void dfgpu_acc_free(dfgpu_acc *a);
/* ≙ GroupsAccumulator::update_batch (expr/src/groups_accumulator.rs:98-104): values may be NULL for
 * COUNT(*); group_ids UINT32 (0xFFFFFFFF = skip); opt_filter BOOL (NULL/false rows skipped). */
DFGPU_API dfgpu_status dfgpu_acc_update_batch(dfgpu_ctx *ctx, dfgpu_acc *a, const dfgpu_array *values, const dfgpu_array *group_ids,
                                              const dfgpu_array *opt_filter, int64_t total_num_groups);
/* The update_batch calls of one group_aggregate_batch (row_hash.rs:560-600: for every accumulator acc.update_batch(values, group_indices,
 * opt_filter, total_num_groups)) handed over together; same results as calling dfgpu_acc_update_batch per accumulator in order.
 * Neighbouring SUM / AVG accumulators over <= 8 groups with the same filter share one pass over the group ids (values[i] == NULL is
 * allowed for COUNT(*); filters may be NULL or hold NULL entries). */
DFGPU_API dfgpu_status dfgpu_acc_update_batch_multi(dfgpu_ctx *ctx, dfgpu_acc *const *accs, const dfgpu_array *const *values, const dfgpu_array *const *filters,
                                                    int32_t n_accs, const dfgpu_array *group_ids, int64_t total_num_groups);
/* The same group_aggregate_batch step with the accumulators' ARGUMENT EXPRESSIONS handed over instead of their evaluated values
 * (row_hash.rs:540-560 evaluates aggregate_expressions per batch, then calls update_batch per accumulator): one pass reads every input
 * column once, evaluates the expression DAG per row in registers and feeds all accumulators.  nodes[k] is a full-length column
 * (op DFGPU_NODE_COLUMN, lhs = index into cols), a literal (DFGPU_NODE_SCALAR, lhs = index into cols of a 1-row array, ≙
 * ColumnarValue::Scalar) or DFGPU_OP_ADD / SUB / MUL over EARLIER nodes lhs, rhs with the result types and checked arithmetic of
 * dfgpu_binary; acc_nodes[i] is accumulator i's argument node (-1 for COUNT(*)).  Taken for SUM / AVG / COUNT over <= 8 groups with
 * all-Float64 or all-Decimal128 non-nullable columns; any other shape returns DFGPU_NOT_IMPLEMENTED and the caller evaluates node by
 * node (dfgpu_binary) and calls dfgpu_acc_update_batch_multi -- the results are identical.  The kernel is compiled for the expression
 * at hand on first use (hiprtc) and cached for the process. */
typedef struct dfgpu_expr_node { int32_t op; int32_t lhs; int32_t rhs; } dfgpu_expr_node;
#define DFGPU_NODE_COLUMN (-1)
#define DFGPU_NODE_SCALAR (-2)
DFGPU_API dfgpu_status dfgpu_acc_update_batch_fused(dfgpu_ctx *ctx, dfgpu_acc *const *accs, const int32_t *acc_nodes, int32_t n_accs,
                                                    const dfgpu_expr_node *nodes, int32_t n_nodes, const dfgpu_array *const *cols, int32_t n_cols,
                                                    const dfgpu_array *group_ids, const dfgpu_array *opt_filter, int64_t total_num_groups);
/* Build check without a device: compiles the fused-aggregate kernel text for a representative Float64 and Decimal128 shape for `arch`
 * (NULL = "gfx950"); the compiler log lands in `log` on failure. */
DFGPU_API dfgpu_status dfgpu_jit_selftest(const char *arch, char *log, int64_t log_capacity);
/* ≙ GroupsAccumulator::merge_batch (:136-142): states as produced by dfgpu_acc_state. */
DFGPU_API dfgpu_status dfgpu_acc_merge_batch(dfgpu_ctx *ctx, dfgpu_acc *a, const dfgpu_array *const *states, int32_t nstates,
                                             const dfgpu_array *group_ids, const dfgpu_array *opt_filter, int64_t total_num_groups);
/* Partial aggregation of ONE batch ahead of intern / merge_batch (≙ AggregateMode::Partial applied inside the operator, aggregates/mod.rs:64-100,
 * row_hash.rs:524-613): the batch is reduced to one row per group -- out_keys[c] = column c of the group keys (type of keys[c]) in first-seen order of
 * their groups, out_states[2 i], out_states[2 i + 1] = the state arrays of aggregate i exactly as dfgpu_acc_state returns them (COUNT: Int64;
 * SUM / MIN / MAX: one array; AVG: UInt64 counts, sums; the second entry is NULL unless AVG).  The caller interns out_keys and calls
 * dfgpu_acc_merge_batch with the states: results equal updating the accumulators with the batch row by row (Float64 sums within 1e-9
 * relative: the additions are reassociated).  Rows are hash-partitioned on the key so that every partition's groups fit one LDS table
 * (csrc/pagg.hip); a key may come back in more than one row when a partition held more groups than the table -- the merge adds them up.
 * Taken for one 4- / 8-byte integer key column (or dictionary codes of that width), or for 1..4 integer / Date32 key columns with or without NULLs
 * whose value ranges multiply to < 2^62 (they travel as ONE packed 64-bit key, NULL = a value of its own, and are unpacked again: out_keys[0 ..
 * nkeys)); SUM / MIN / MAX over Int64 / UInt64, SUM / AVG over Float64 and over Decimal128 (exact 128-bit sums; state Decimal128(min(38, p + 10), s)),
 * COUNT; value columns may hold NULLs (a NULL takes no part, a group without a value has a NULL state, COUNT(x) / AVG count the values: accumulate.rs:126-233) -- up to 8
 * nullable columns, each costs one of the 6 accumulator cells; no per-aggregate filter, a batch of >= option "agg_partitioned_min_rows" rows whose keys are neither clustered
 * nor few (a sample decides); any other shape returns DFGPU_NOT_IMPLEMENTED and the caller updates the accumulators the ordinary way.
 * values[i] may be NULL for COUNT(*).  opt_mask: BOOL selection, unselected rows do not take part.
 * out_keys == NULL asks for the verdict only (DFGPU_OK = a following call with the same key column and selection will be taken, as far as the
 * key decides; kinds / values are ignored): the plan layer asks before it evaluates computed aggregate arguments. */
DFGPU_API dfgpu_status dfgpu_agg_preaggregate(dfgpu_ctx *ctx, const dfgpu_array *const *keys, int32_t nkeys, const int32_t *kinds, const dfgpu_array *const *values, int32_t n_aggs,
                                              const dfgpu_array *opt_mask, dfgpu_array **out_keys, dfgpu_array **out_states);
/* The same with flags.  DFGPU_PREAGG_ANY_ORDER: the caller does not depend on the order of the partial rows (the reference's hash aggregation promises none:
 * AggregateExec::output_ordering is None for unordered input, aggregates/mod.rs:560-600; its stream happens to emit groups in first-seen order) -- the rows leave in
 * partition order and the pass that restores first-seen order (a third of the call at 20 M groups) is skipped.  The plan layer sets it for an AggregateExec whose
 * consumer is a SortExec over all of its group columns, where the order of the input rows cannot show in the output. */
#define DFGPU_PREAGG_ANY_ORDER 1
/* value_casts (optional, one entry per aggregate): DFGPU_FLOAT64 = the aggregate's argument is CAST(values[i] AS DOUBLE) of the Int32 or Int64 column values[i],
 * which is how SUM / AVG over an integer column reach the operator after type coercion (AVG: physical-expr/src/aggregate/average.rs:96-110 takes Float64 / Decimal128
 * only); the column is converted while it is partitioned (arrow-cast's `as f64`) instead of by a cast pass of its own, the states are those of the Float64 argument.
 * 0 = the argument is values[i] itself; any other value declines (DFGPU_NOT_IMPLEMENTED). */
DFGPU_API dfgpu_status dfgpu_agg_preaggregate_flags(dfgpu_ctx *ctx, const dfgpu_array *const *keys, int32_t nkeys, const int32_t *kinds, const dfgpu_array *const *values, const int32_t *value_casts,
                                                    int32_t n_aggs, const dfgpu_array *opt_mask, int32_t flags, dfgpu_array **out_keys, dfgpu_array **out_states);
/* ≙ evaluate(EmitTo::All) / state(EmitTo::All) (:106-134).  out_states holds up to 2 arrays. */
DFGPU_API dfgpu_status dfgpu_acc_evaluate(dfgpu_ctx *ctx, dfgpu_acc *a, dfgpu_array **out);
DFGPU_API dfgpu_status dfgpu_acc_state(dfgpu_ctx *ctx, dfgpu_acc *a, dfgpu_array **out_states, int32_t *n_states);
/* ≙ GroupsAccumulator::evaluate / state with EmitTo::First(n) (prim_op.rs:129-140, average.rs:432-470, count.rs:172-190): as_state = 0 -> out[0] = the final values of the
 * first n groups; as_state = 1 -> out[0 .. *n_out) = their state arrays; the remaining groups keep their state under ids lowered by n. */
DFGPU_API dfgpu_status dfgpu_acc_emit_first(dfgpu_ctx *ctx, dfgpu_acc *a, int64_t n, int32_t as_state, dfgpu_array **out, int32_t *n_out);
DFGPU_API int64_t dfgpu_acc_size(const dfgpu_acc *a);

/* ------------------------------------------------------------------ a13: SortExec */
/* ≙ lexsort_to_indices (sorts/sort.rs:599) with per column SortOptions (physical-expr/src/sort_expr.rs:34-74);
 * fetch < 0 = none.  Stable (ties keep input order; the reference leaves tie order unspecified).
 * out: UINT32 indices. */
DFGPU_API dfgpu_status dfgpu_sort_to_indices(dfgpu_ctx *ctx, const dfgpu_array *const *cols, const uint8_t *descending,
                                             const uint8_t *nulls_first, int32_t k, int64_t fetch, dfgpu_array **out);
/* dfgpu_sort_to_indices plus a by-product: out_sorted[c] (k entries) = cols[c] in the sorted order (== take(cols[c], *out)) when the sort could rebuild it from its
 * packed keys -- fixed-width key columns without NULLs on the packed-key path -- else NULL; sort_batch's take() (sorts/sort.rs:598-603) of such a column is then free. */
DFGPU_API dfgpu_status dfgpu_sort_to_indices_keys(dfgpu_ctx *ctx, const dfgpu_array *const *cols, const uint8_t *descending, const uint8_t *nulls_first, int32_t k, int64_t fetch,
                                                  dfgpu_array **out, dfgpu_array **out_sorted);
/* sort_batch in one call (sorts/sort.rs:584-609: lexsort_to_indices, then take() of every column): dfgpu_sort_to_indices_keys plus the batch's other columns.
 * out_payload[c] (n_payload entries) = take(payload[c], *out) when the sort's last pass could gather it while it writes the result -- a fixed-width column of 4 / 8 / 16
 * bytes without NULLs, at most four of them, on the one-launch-per-pass path (keys and row number in one word, 2^20 rows or more) -- else NULL and the caller takes the
 * column through *out as before.  Context option "sort_payload_in_last_pass" (default 0): measured on MI355X the fused gather is slower than the separate one (100 M rows, one
 * 8-byte column: 6.36 ms for passes + gather against 3.89 + 2.13 ms), so by default every entry of out_payload is NULL; the entry point stays so that a caller states the
 * whole of sort_batch in one call. */
DFGPU_API dfgpu_status dfgpu_sort_take(dfgpu_ctx *ctx, const dfgpu_array *const *cols, const uint8_t *descending, const uint8_t *nulls_first, int32_t k, int64_t fetch,
                                       const dfgpu_array *const *payload, int32_t n_payload, dfgpu_array **out, dfgpu_array **out_sorted, dfgpu_array **out_payload);

/* ------------------------------------------------------------------ a14: RepartitionExec */
/* ≙ BatchPartitioner::partition_iter, Hash(exprs, n) (repartition/mod.rs:148-221): destination =
 * create_hashes(keys) % n; out_indices (UINT32) lists the rows grouped by destination, input
 * order kept inside each destination (:196-214); counts_host[n] receives rows per destination. */
DFGPU_API dfgpu_status dfgpu_hash_partition(dfgpu_ctx *ctx, const dfgpu_array *const *keys, int32_t nkeys, int32_t num_partitions,
                                            dfgpu_array **out_indices, int64_t *counts_host);

/* The same partitioning with the `take` of every column folded in (repartition/mod.rs:196-214): ONE pass reads key and payload columns and
 * writes them grouped by destination.  out_cols[c] = column c of all destinations back to back (slice it by counts_host) for fixed-width
 * columns without NULLs; NULL for any other column (Utf8, dictionary, Boolean, nullable), which the caller gathers through out_indices
 * (always produced: the original row numbers grouped by destination, input order kept inside a destination).  cols[c] may be NULL
 * (a column the caller keeps lazy).  opt_mask: BOOL selection, unselected rows are dropped.  1..256 partitions (else NOT_IMPLEMENTED:
 * use dfgpu_hash_partition). */
DFGPU_API dfgpu_status dfgpu_partition_columns(dfgpu_ctx *ctx, const dfgpu_array *const *keys, int32_t nkeys, int32_t num_partitions,
                                               const dfgpu_array *const *cols, int32_t ncols, const dfgpu_array *opt_mask,
                                               dfgpu_array **out_cols, dfgpu_array **out_indices, int64_t *counts_host);

/* ------------------------------------------------------------------ a14 across GPUs: the exchange of RepartitionExec */
/* One process (or ctx) per GPU.  A dfgpu_comm is the job's communicator: RCCL over xGMI (dfgpu_comm_create_rccl; rank 0 obtains the 128-byte
 * id from dfgpu_comm_unique_id and hands it to the other ranks by any means, as with ncclCommInitRank), or a transport the caller provides as
 * two callbacks (dfgpu_comm_create_custom: an MPI host, or the CPU tests over torch.distributed's gloo backend). */
typedef struct dfgpu_comm dfgpu_comm;
typedef struct dfgpu_comm_vtable {
  void *user; int32_t rank, world;
  /* every rank contributes `bytes` HOST bytes; recv (world * bytes, host) receives the contributions in rank order */
  int32_t (*all_gather_host)(void *user, const void *send, int64_t bytes, void *recv);
  /* DEVICE buffers: segment p of `send` (send_off[p], send_bytes[p]) goes to rank p, segment p of `recv` receives what rank p sends here.
   * Returns (0 = ok) when `recv` is complete. */
  int32_t (*all_to_all_v)(void *user, const void *send, const int64_t *send_off, const int64_t *send_bytes, void *recv, const int64_t *recv_off, const int64_t *recv_bytes);
} dfgpu_comm_vtable;
DFGPU_API dfgpu_status dfgpu_comm_unique_id(uint8_t *out_id128);
DFGPU_API dfgpu_status dfgpu_comm_create_rccl(dfgpu_ctx *ctx, const uint8_t *id128, int32_t rank, int32_t world, dfgpu_comm **out);
DFGPU_API dfgpu_status dfgpu_comm_create_custom(const dfgpu_comm_vtable *vtable, dfgpu_comm **out);
DFGPU_API void dfgpu_comm_free(dfgpu_comm *comm);
DFGPU_API int32_t dfgpu_comm_rank(const dfgpu_comm *comm);
DFGPU_API int32_t dfgpu_comm_world(const dfgpu_comm *comm);
/* ≙ RepartitionExec with Partitioning::Hash(keys, world) between processes (repartition/mod.rs:442-580, :684-760 -- the in-process channels
 * become an all-to-all): every rank hash-partitions its rows (create_hashes % world, identical on all ranks; dfgpu_partition_columns), the
 * row-count matrix is exchanged, then ONE grouped collective moves every column buffer.  out_cols[c] = the rows of column c this rank owns
 * afterwards, ordered by source rank (row i of every column belongs to one row).  Any column type RepartitionExec moves: fixed width, Boolean,
 * Utf8 (lengths + value bytes as two lanes) and dictionary-encoded columns (sent as their values), with or without NULLs.  opt_mask: unselected
 * rows are not sent.  out_counts (optional, 2 * world): rows sent to / received from every rank.  Collective: every rank of the communicator must
 * call it with the same number of columns; a rank whose input produced no batch passes keys = cols = NULL and learns the column types from the
 * others (out_cols stay NULL when no rank had rows).  A column that is nullable on any rank arrives with a validity bitmap on every rank.
 * Failure is collective too: a rank whose local step fails (bad argument, allocation) still joins the metadata all-gather with a status word, and
 * EVERY rank returns an error without entering the data collective -- no rank is left waiting.  Under a ctx memory limit the receive-buffer
 * allocation is agreed on in a second status round.  A transport error inside the data collective closes the RCCL group before it is reported,
 * so the communicator stays usable. */
DFGPU_API dfgpu_status dfgpu_exchange(dfgpu_ctx *ctx, dfgpu_comm *comm, const dfgpu_array *const *keys, int32_t nkeys, const dfgpu_array *const *cols, int32_t ncols,
                                      const dfgpu_array *opt_mask, dfgpu_array **out_cols, int64_t *out_counts);

/* ------------------------------------------------------------------ scan: Parquet column chunks -> Arrow columns in HBM */
/* ≙ what ParquetExec's stream does per row group (core/src/datasource/physical_plan/parquet/mod.rs: ParquetOpener::open :417-560 ->
 * ParquetRecordBatchStreamBuilder of the `parquet` crate (arrow-rs 50, not part of the reference tree) -> RecordBatches), for flat schemas:
 * the footer and the page headers are parsed on the host, pages are decompressed (UNCOMPRESSED, SNAPPY, ZSTD, LZ4_RAW) and decoded (PLAIN, PLAIN_DICTIONARY /
 * RLE_DICTIONARY, RLE definition levels; data pages v1 and v2) by device kernels.  Column types follow the crate's parquet -> arrow rules:
 * BOOLEAN, INT32 / INT64 with their INT(8..64, signed / unsigned), DATE and DECIMAL annotations, FLOAT, DOUBLE, BYTE_ARRAY (STRING / UTF8),
 * FIXED_LEN_BYTE_ARRAY (DECIMAL, <= 16 bytes).  Other columns (nested, INT96, timestamps, binary) report type 0 and fail to read with
 * DFGPU_NOT_IMPLEMENTED -- the file's other columns stay readable (projection).
 * dfgpu_parquet_open: `file_bytes` is the whole file in host memory (kept by the caller until close); `device_bytes` (optional) is the same image
 * already resident in HBM (a GPUDirect read, or a cached file) -- pages are then decoded in place, otherwise each column chunk read is copied
 * to the device first, on a copy stream of the context's own, one event per column: the decode kernels of a column start when its chunks have
 * arrived, while the next columns are still on the wire (≙ the byte-range prefetch of ParquetOpener's AsyncFileReader).  These copies run at PCIe
 * speed and beside the kernels only from page-locked memory: dfgpu_parquet_open_file maps `path` and page-locks the mapping (stage_on_device = 1
 * copies the image to HBM once instead); a caller of dfgpu_parquet_open who wants the same passes page-locked `file_bytes` (hipHostMalloc /
 * hipHostRegister) -- pageable memory works, through the runtime's bounce buffer.
 * Option "utf8_dictionary" (default 1): Utf8 columns are handed over as Dictionary(Int32, Utf8) -- the page indices plus the row group's base in
 * the concatenated dictionaries; nothing is expanded, and dictionary predicates / the canonical-id group-by take the column as it is.  Chunks that
 * fell back to PLAIN pages get identity keys, so the column type is the same in every batch.  0 = plain Utf8 columns, as the reference decodes. */
typedef struct dfgpu_parquet dfgpu_parquet;
DFGPU_API dfgpu_status dfgpu_parquet_open(dfgpu_ctx *ctx, const uint8_t *file_bytes, int64_t len, const uint8_t *device_bytes, dfgpu_parquet **out);
DFGPU_API dfgpu_status dfgpu_parquet_open_file(dfgpu_ctx *ctx, const char *path, int32_t stage_on_device, dfgpu_parquet **out);
DFGPU_API void dfgpu_parquet_close(dfgpu_parquet *file);
DFGPU_API dfgpu_status dfgpu_parquet_set_option(dfgpu_parquet *file, const char *key, int64_t value);
DFGPU_API int64_t dfgpu_parquet_num_rows(const dfgpu_parquet *file);
DFGPU_API int32_t dfgpu_parquet_num_row_groups(const dfgpu_parquet *file);
DFGPU_API int32_t dfgpu_parquet_num_columns(const dfgpu_parquet *file);              /* leaf columns == column chunks per row group */
DFGPU_API int64_t dfgpu_parquet_row_group_rows(const dfgpu_parquet *file, int32_t row_group);
DFGPU_API const char *dfgpu_parquet_column_name(const dfgpu_parquet *file, int32_t column);
/* type: DFGPU_* of the column as read (DFGPU_DICTIONARY for Utf8 under "utf8_dictionary"; 0 = unsupported); value_type: the logical type */
DFGPU_API dfgpu_status dfgpu_parquet_column_type(const dfgpu_parquet *file, int32_t column, int32_t *type, int32_t *value_type, int32_t *precision, int32_t *scale, int32_t *nullable);
/* Row-group statistics of an integer / date column for pruning (≙ parquet/row_groups.rs: prune_row_groups_by_statistics): has_min_max = 0 when the
 * chunk carries none (or the column's order is not the physical one); null_count -1 = unknown */
DFGPU_API dfgpu_status dfgpu_parquet_column_stats(const dfgpu_parquet *file, int32_t row_group, int32_t column, int64_t *min_value, int64_t *max_value, int64_t *null_count, int32_t *has_min_max);
DFGPU_API int64_t dfgpu_parquet_column_chunk_bytes(const dfgpu_parquet *file, int32_t row_group, int32_t column, int32_t uncompressed);
/* Decode `ncols` columns (leaf indices) of row groups [first_row_group, first_row_group + num_row_groups) into one array per column, rows of
 * consecutive row groups back to back.  A column without NULLs in these row groups comes without a validity bitmap. */
DFGPU_API dfgpu_status dfgpu_parquet_read(dfgpu_ctx *ctx, dfgpu_parquet *file, int32_t first_row_group, int32_t num_row_groups, const int32_t *columns, int32_t ncols, dfgpu_array **out);

/* ≙ what CsvExec's stream does per file (core/src/datasource/physical_plan/csv.rs: CsvOpener -> the `arrow-csv` reader of arrow-rs 50, not part of the reference tree): records
 * are lines (LF or CRLF; a line feed inside a quoted field does not end a record; blank lines are skipped), fields are separated by `delimiter`, a field may be wrapped in `quote` characters with a
 * doubled quote standing for one; an empty field of a non-string column is NULL, of a Utf8 column the empty string.  The schema is the caller's: `columns` = ascending indices
 * of the wanted file columns, `types` = (DFGPU_* type, precision, scale) per wanted column -- Int8 .. UInt64, Float64 (inputs of up to 15 significant digits and exponents of
 * at most 22: the exactly rounded path; longer ones raise an error instead of a guess), Boolean, Date32 (YYYY-MM-DD), Decimal128, Utf8.  A field that does not parse as its
 * column's type, a record with too few fields or broken quoting fail the call with DFGPU_EXECUTION.  `bytes` is the file image in host memory, or already in HBM
 * (bytes_on_device = 1); at most 4 GB per call.  escape (0 = none; CsvExec::escape, csv.rs:59, :304): inside a quoted field `escape` followed by any byte stands for that byte.
 * has_header = 1 skips the first record.  out_rows (optional) = records read. */
DFGPU_API dfgpu_status dfgpu_csv_read(dfgpu_ctx *ctx, const uint8_t *bytes, int64_t len, int32_t bytes_on_device, int32_t delimiter, int32_t quote, int32_t escape, int32_t has_header, int32_t ncols_file,
                                      const int32_t *columns, const int32_t *types, int32_t ncols, dfgpu_array **out, int64_t *out_rows);

#ifdef __cplusplus
}
#endif
#endif
