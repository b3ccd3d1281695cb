/*
 * dfgpu_exec.h -- C ABI of the C++ host layer (datafusion-upstream_amd/csrc/exec/) that mirrors the reference's
 * operator interface for the hot path on top of the kernel-level ABI of dfgpu.h:
 *
 *   trait PhysicalExpr   datafusion/physical-expr/src/physical_expr.rs:96-123
 *   trait ExecutionPlan  datafusion/physical-plan/src/lib.rs:115-405   (execute(partition, ctx) -> RecordBatchStream)
 *   MemoryExec memory.rs:40 | FilterExec filter.rs:56 | ProjectionExec projection.rs:52 | CoalesceBatchesExec
 *   coalesce_batches.rs | CoalescePartitionsExec coalesce_partitions.rs | RepartitionExec repartition/mod.rs:232 |
 *   HashJoinExec joins/hash_join.rs:283 | AggregateExec aggregates/mod.rs:242 | SortExec sorts/sort.rs:719
 *
 * A Rust shim can bind either level: dfgpu.h (one call per reference inner function) or this one (one handle per plan
 * node; `dfgpu_plan_execute` + `dfgpu_stream_next` are `ExecutionPlan::execute` + `Stream::poll_next`).
 * Ownership: constructors take shared ownership of their inputs (reference counted inside); every handle returned to
 * the caller is released with the matching *_free.  Errors: dfgpu_status + dfgpu_exec_last_error() (thread local).
 */
#ifndef DFGPU_EXEC_H
#define DFGPU_EXEC_H
#include "dfgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dfgpu_expr dfgpu_expr;       /* Arc<dyn PhysicalExpr> */
typedef struct dfgpu_plan dfgpu_plan;       /* Arc<dyn ExecutionPlan> */
typedef struct dfgpu_batch dfgpu_batch;     /* RecordBatch resident in HBM */
typedef struct dfgpu_stream dfgpu_stream;   /* SendableRecordBatchStream */

DFGPU_API const char *dfgpu_exec_last_error(void);

/* ---- RecordBatch */
DFGPU_API dfgpu_status dfgpu_batch_new(const char *const *names, const dfgpu_array *const *columns, int32_t ncols, dfgpu_batch **out);
DFGPU_API void dfgpu_batch_free(dfgpu_batch *b);
DFGPU_API int32_t dfgpu_batch_num_columns(const dfgpu_batch *b);
/* rows after applying the selection mask (materialises nothing but the mask popcount) */
DFGPU_API dfgpu_status dfgpu_batch_num_rows(dfgpu_ctx *ctx, dfgpu_batch *b, int64_t *out);
DFGPU_API const char *dfgpu_batch_column_name(const dfgpu_batch *b, int32_t i);
/* applies the selection and executes every pending gather of the batch at once: columns that go through the same index array are gathered
 * together (dfgpu_take_multi).  Optional: dfgpu_batch_column does the same for one column at a time. */
DFGPU_API dfgpu_status dfgpu_batch_materialize(dfgpu_ctx *ctx, dfgpu_batch *b);
/* column i, selection applied and lazy gathers executed; caller releases the array */
DFGPU_API dfgpu_status dfgpu_batch_column(dfgpu_ctx *ctx, dfgpu_batch *b, int32_t i, dfgpu_array **out);

/* ---- PhysicalExpr (physical-expr/src/expressions/) */
DFGPU_API dfgpu_status dfgpu_expr_column(const char *name, int32_t index, dfgpu_expr **out);                       /* column.rs */
DFGPU_API dfgpu_status dfgpu_expr_literal(const dfgpu_array *scalar_len1, dfgpu_expr **out);                      /* literal.rs */
DFGPU_API dfgpu_status dfgpu_expr_binary(const dfgpu_expr *l, int32_t op /* DFGPU_OP_* */, const dfgpu_expr *r, dfgpu_expr **out); /* binary.rs */
DFGPU_API dfgpu_status dfgpu_expr_not(const dfgpu_expr *e, dfgpu_expr **out);
DFGPU_API dfgpu_status dfgpu_expr_is_null(const dfgpu_expr *e, int32_t negated, dfgpu_expr **out);
DFGPU_API dfgpu_status dfgpu_expr_negative(const dfgpu_expr *e, dfgpu_expr **out);
DFGPU_API dfgpu_status dfgpu_expr_cast(const dfgpu_expr *e, int32_t to_type, int32_t precision, int32_t scale, dfgpu_expr **out);
DFGPU_API dfgpu_status dfgpu_expr_in_list(const dfgpu_expr *e, const dfgpu_array *list, int32_t negated, dfgpu_expr **out);
DFGPU_API void dfgpu_expr_free(dfgpu_expr *e);

/* ---- ExecutionPlan nodes */
/* MemoryExec: batches of all partitions back to back, partition_sizes[p] batches each */
DFGPU_API dfgpu_status dfgpu_plan_memory(const dfgpu_batch *const *batches, const int32_t *partition_sizes, int32_t npartitions, dfgpu_plan **out);
/* Replaces the batches of a MemoryExec in place (same schema): the input slot of a plan that is built once and executed many times
 * -- e.g. the segment above an exchange, whose input is whatever the collective delivered in this execution. */
DFGPU_API dfgpu_status dfgpu_plan_memory_replace(dfgpu_plan *memory_exec, const dfgpu_batch *const *batches, const int32_t *partition_sizes, int32_t npartitions);
/* ParquetExec (core/src/datasource/physical_plan/parquet/mod.rs:78): scan of one open file (include/dfgpu.h dfgpu_parquet_*; it must stay open while
 * the plan lives).  columns = leaf indices (the projection); the file's row groups are dealt to `npartitions` output partitions in contiguous runs;
 * every batch holds `row_groups_per_batch` row groups.  dfgpu_plan_parquet_prune: skip row groups whose statistics of an integer / date column lie
 * outside [min_value, max_value] (≙ parquet/row_groups.rs; conservative -- keep the FilterExec); dfgpu_plan_parquet_pruned = row groups skipped so far
 * (≙ ParquetFileMetrics::row_groups_pruned, parquet/metrics.rs). */
DFGPU_API dfgpu_status dfgpu_plan_parquet(dfgpu_parquet *file, const int32_t *columns, int32_t ncols, int32_t npartitions, int32_t row_groups_per_batch, dfgpu_plan **out);
/* CsvExec (core/src/datasource/physical_plan/csv.rs:53): scan of one file image held by the caller (host memory; it must outlive the plan) under the table's
 * schema: names[ncols_file], types[3 * ncols_file] = (DFGPU type, precision, scale) of EVERY file column, columns[ncols] = the projection (ascending file column
 * indices).  The image is cut at record boundaries into pieces of about batch_bytes (0 = 256 MiB; <= 4 GB) -- one batch each, parsed on the device by
 * dfgpu_csv_read -- and the pieces are dealt to npartitions in contiguous runs (the byte-range file groups of FileScanConfig, csv.rs:362-420). */
DFGPU_API dfgpu_status dfgpu_plan_csv(const uint8_t *bytes, int64_t len, int32_t delimiter, int32_t quote, int32_t escape, int32_t has_header, const char *const *names, const int32_t *types,
                                      int32_t ncols_file, const int32_t *columns, int32_t ncols, int32_t npartitions, int64_t batch_bytes, dfgpu_plan **out);
DFGPU_API dfgpu_status dfgpu_plan_parquet_prune(dfgpu_plan *parquet_exec, int32_t column, int64_t min_value, int64_t max_value);
DFGPU_API int64_t dfgpu_plan_parquet_pruned(const dfgpu_plan *parquet_exec);
DFGPU_API dfgpu_status dfgpu_plan_filter(const dfgpu_expr *predicate, const dfgpu_plan *input, dfgpu_plan **out);
DFGPU_API dfgpu_status dfgpu_plan_projection(const dfgpu_expr *const *exprs, const char *const *names, int32_t n, const dfgpu_plan *input, dfgpu_plan **out);
DFGPU_API dfgpu_status dfgpu_plan_coalesce_batches(const dfgpu_plan *input, int64_t target_batch_size, dfgpu_plan **out);
DFGPU_API dfgpu_status dfgpu_plan_coalesce_partitions(const dfgpu_plan *input, dfgpu_plan **out);
/* Partitioning::Hash(exprs, n) when nexprs > 0, else Partitioning::RoundRobinBatch(n) */
DFGPU_API dfgpu_status dfgpu_plan_repartition(const dfgpu_plan *input, const dfgpu_expr *const *exprs, int32_t nexprs, int32_t num_partitions, dfgpu_plan **out);
/* HashJoinExec::try_new(left, right, on, filter, join_type, mode, null_equals_null); mode 0 = CollectLeft, 1 = Partitioned.
 * JoinFilter: filter == NULL for none; filter_sides[i] 0 = left, 1 = right; filter_indices[i] = column of that side. */
DFGPU_API dfgpu_status dfgpu_plan_hash_join(const dfgpu_plan *left, const dfgpu_plan *right, const dfgpu_expr *const *on_left, const dfgpu_expr *const *on_right,
                                            int32_t non, const dfgpu_expr *filter, const int32_t *filter_sides, const int32_t *filter_indices, int32_t nfilter_cols,
                                            int32_t join_type, int32_t mode, int32_t null_equals_null, dfgpu_plan **out);
/* SortMergeJoinExec::try_new(left, right, on, filter, join_type, sort_options, null_equals_null) (joins/sort_merge_join.rs:95-160): inputs sorted on the keys and
 * partitioned alike; output rows in streamed-side order (left; right for JoinType::Right), each with its matches in buffered order, unmatched outer rows in place.
 * Inner, Left, Right, Full, LeftSemi, LeftAnti, RightAnti; RightSemi is refused as the reference refuses it.  Full adds the buffered rows no streamed row matched as a
 * last batch.  JoinFilter as for dfgpu_plan_hash_join, for Inner / Left / Right / Full, with the reference's own semantics (:1156-1300): it is applied to the joined pairs,
 * and an outer join emits every FAILING pair NULL-joined (once for the streamed side; for Full once more for the buffered side) -- sort_merge_join.slt:137-147 pins this.
 * LeftSemi / LeftAnti / RightAnti with a filter answer DFGPU_NOT_IMPLEMENTED (the reference's freeze_streamed builds no buffered columns for them, :1133-1135, so a filter over the
 * buffered side has nothing to be evaluated on there either).  The sort options do not enter: row order follows the inputs' own order. */
DFGPU_API dfgpu_status dfgpu_plan_sort_merge_join(const dfgpu_plan *left, const dfgpu_plan *right, const dfgpu_expr *const *on_left, const dfgpu_expr *const *on_right, int32_t non,
                                                  const dfgpu_expr *filter, const int32_t *filter_sides, const int32_t *filter_indices, int32_t nfilter_cols,
                                                  int32_t join_type, int32_t null_equals_null, dfgpu_plan **out);
/* NestedLoopJoinExec::try_new(left, right, filter, join_type) (joins/nested_loop_join.rs:102-127): no equi-join keys; the side named by left_is_build_side
 * (:373-378) is collected, the other side streams and decides the output partitioning.  JoinFilter as for dfgpu_plan_hash_join (NULL = cross join). */
DFGPU_API dfgpu_status dfgpu_plan_nested_loop_join(const dfgpu_plan *left, const dfgpu_plan *right, const dfgpu_expr *filter, const int32_t *filter_sides, const int32_t *filter_indices,
                                                   int32_t nfilter_cols, int32_t join_type, dfgpu_plan **out);
/* Aggregate kinds of the plan layer beyond DFGPU_AGG_* (include/dfgpu.h): COUNT(DISTINCT x) (aggregate/count_distinct/) -- every mode for fixed-width arguments: the Partial state is the reference's List of distinct values per group, carried in the
 * Utf8 layout (dfgpu_list_from_counts / dfgpu_list_flatten, include/dfgpu.h; a ListArray is the same buffers with the offsets divided by the value width); Utf8 arguments in Single /
 * SinglePartitioned modes only.  MIN / MAX
 * over Utf8 (input type DFGPU_UTF8) are also served by the plan layer, in every mode. */
enum { DFGPU_AGG_COUNT_DISTINCT = 5 };
/* AggregateExec::try_new(mode, group_by, aggr_expr, input): mode 0 Partial, 1 Final, 2 FinalPartitioned, 3 Single,
 * 4 SinglePartitioned.  Aggregate i: kind DFGPU_AGG_*, argument expr (NULL = COUNT(*)), optional FILTER expr, output name,
 * argument data type (type, precision, scale) as the AggregateExpr knows it in every mode. */
DFGPU_API dfgpu_status dfgpu_plan_aggregate(int32_t mode, const dfgpu_expr *const *group_exprs, const char *const *group_names, int32_t ngroups,
                                            const int32_t *agg_kinds, const dfgpu_expr *const *agg_args, const dfgpu_expr *const *agg_filters, const char *const *agg_names,
                                            const int32_t *agg_arg_types /* 3 per aggregate */, int32_t naggs, const dfgpu_plan *input, dfgpu_plan **out);
/* PhysicalGroupBy with grouping sets (aggregates/mod.rs:103-160; GROUPING SETS / CUBE / ROLLUP): null_exprs[i] is the typed NULL
 * literal that stands in for group expression i, groups[s * nkeys + i] != 0 says set s replaces key i by it.  Every input batch is
 * grouped once per set into the same GroupValues (evaluate_group_by, :1161-1200).  Call before the plan first executes. */
/* InputOrderMode of an AggregateExec (physical-plan/src/ordering.rs:33-44; GroupOrdering, aggregates/order/{mod,full,partial}.rs): 0 Linear (default), 1 PartiallySorted --
 * the input is sorted on the group keys order_indices[0..n) (indices into group_exprs, in sort order), 2 Sorted -- on all of them.  After every input batch the groups
 * that can receive no more rows leave as an output batch (EmitTo::First(n), row_hash.rs:455-461): all but the last group (Sorted) / all groups in front of the latest sort
 * prefix (PartiallySorted); the group table holds only the open groups.  The concatenated output is the same as in Linear mode. */
DFGPU_API dfgpu_status dfgpu_plan_aggregate_input_order(dfgpu_plan *aggregate, int32_t input_order_mode, const int32_t *order_indices, int32_t n);
DFGPU_API dfgpu_status dfgpu_plan_aggregate_grouping_sets(dfgpu_plan *aggregate, const dfgpu_expr *const *null_exprs, int32_t nkeys, const uint8_t *groups, int32_t nsets);
/* SortExec::new(expr, input).with_fetch(fetch).with_preserve_partitioning(..); fetch < 0 = none */
DFGPU_API dfgpu_status dfgpu_plan_sort(const dfgpu_expr *const *exprs, const uint8_t *descending, const uint8_t *nulls_first, int32_t n, int64_t fetch,
                                       int32_t preserve_partitioning, const dfgpu_plan *input, dfgpu_plan **out);
/* SortPreservingMergeExec::new(expr, input).with_fetch(fetch) (sorts/sort_preserving_merge.rs:67-120): merges the input's sorted
 * partitions into one sorted partition; equal keys keep partition order (lower partition first); one input partition is passed
 * through; fetch < 0 = none.  The inputs must already be sorted on `exprs`. */
DFGPU_API dfgpu_status dfgpu_plan_sort_preserving_merge(const dfgpu_expr *const *exprs, const uint8_t *descending, const uint8_t *nulls_first, int32_t n, int64_t fetch,
                                                        const dfgpu_plan *input, dfgpu_plan **out);
DFGPU_API void dfgpu_plan_free(dfgpu_plan *p);
/* ExecutionPlan::with_new_children(self, same children, recursively) (physical-plan/src/lib.rs:198-201): a copy of the plan tree
 * without run-once state -- HashJoinExec's OnceAsync build side (joins/utils.rs:736-776; with_new_children hash_join.rs:559),
 * RepartitionExec's pulled input (repartition/mod.rs:422) -- so one plan description can be executed again from scratch. */
DFGPU_API dfgpu_status dfgpu_plan_with_fresh_state(const dfgpu_plan *p, dfgpu_plan **out);
DFGPU_API int32_t dfgpu_plan_partition_count(const dfgpu_plan *p);               /* output_partitioning().partition_count() */
DFGPU_API int32_t dfgpu_plan_schema_len(const dfgpu_plan *p);
DFGPU_API const char *dfgpu_plan_schema_name(const dfgpu_plan *p, int32_t i);
DFGPU_API const char *dfgpu_plan_name(const dfgpu_plan *p);                      /* DisplayAs: "HashJoinExec", ... */

/* ExecutionPlan::metrics() of every node of the tree (physical-plan/src/lib.rs:385-390; DisplayableExecutionPlan::with_metrics), collected while
 * the ctx option "collect_metrics" was 1 when the plan's streams were created.  One line per node, pre-order:
 *   "<depth> <NodeName> output_rows=<n> output_batches=<n> elapsed_compute=<ns> [build_time=<ns> join_time=<ns>] [repartition_time=<ns>]"
 * -- the names of BaselineMetrics (metrics/baseline.rs:47), BuildProbeJoinMetrics (joins/utils.rs:1368) and RepartitionMetrics
 * (repartition/mod.rs:312-349).  Times are device times of what the operator enqueued (HIP events on the ctx stream), elapsed_compute without
 * the children's share.  Reading the metrics waits for the recorded work. */
DFGPU_API dfgpu_status dfgpu_plan_metrics(const dfgpu_plan *p, char *buf, int64_t capacity);

/* ---- execution */
/* ExecutionPlan::execute(partition, TaskContext{session_config.batch_size}) */
DFGPU_API dfgpu_status dfgpu_plan_execute(const dfgpu_plan *p, int32_t partition, dfgpu_ctx *ctx, int64_t batch_size, dfgpu_stream **out);
/* Stream::poll_next: *out = NULL at end of stream */
DFGPU_API dfgpu_status dfgpu_stream_next(dfgpu_stream *s, dfgpu_batch **out);
DFGPU_API void dfgpu_stream_free(dfgpu_stream *s);

#ifdef __cplusplus
}
#endif
#endif
