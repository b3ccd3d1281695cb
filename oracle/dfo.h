/*
 * dfo.h -- CPU ORACLE for the DataFusion 36 hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This directory is a plain-C restatement of the reference's CPU algorithms for the
 * path named in BASELINE.json (HashJoinExec / AggregateExec / Filter+Projection
 * expression evaluation / SortExec / RepartitionExec).  It exists so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg can CHECK (and time) the
 * HIP product path.  Nothing in datafusion-upstream_amd/ may include, link or call it.
 *
 * Pinning: every function cites the reference file:line it restates; the restatement is
 * pinned against the reference's own known-answer tests transcribed under tests/golden/
 * (see tests/test_oracle_golden.py).  The Rust reference itself cannot be compiled in
 * the build container (no cargo/rustc), so there is no oracle/_ref.
 *
 * Hash VALUES are not part of the contract: the reference hashes with ahash (not in
 * tree, platform dependent, never asserted by its tests -- hash_utils.rs:448-728 check
 * relations only).  The oracle therefore uses the same documented splitmix64-based hash
 * as the product (DESIGN.md "hash function") while keeping the reference's structure:
 * per-column hash, combine_hashes, NULL leaves the running hash unchanged.
 */
#ifndef DFO_H
#define DFO_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Type ids: numerically identical to dfgpu_type in include/dfgpu.h so that the Python
 * test harness marshals columns once for both sides. */
enum {
  DFO_BOOL = 1, DFO_INT8 = 2, DFO_INT16 = 3, DFO_INT32 = 4, DFO_INT64 = 5,
  DFO_UINT8 = 6, DFO_UINT16 = 7, DFO_UINT32 = 8, DFO_UINT64 = 9,
  DFO_FLOAT32 = 10, DFO_FLOAT64 = 11, DFO_DATE32 = 12, DFO_DECIMAL128 = 13,
  DFO_UTF8 = 14, DFO_DICTIONARY = 15
};

/* Arrow-layout column (host memory, bit offset 0). Same field order as dfgpu_array_desc. */
typedef struct dfo_array {
  int32_t type;
  int32_t precision;            /* DECIMAL128 */
  int32_t scale;                /* DECIMAL128 */
  int32_t key_type;             /* DICTIONARY: integer type of the `values` (keys) buffer */
  int64_t length;
  int64_t null_count;           /* -1 = unknown */
  const void *values;           /* fixed width values | utf8 bytes | dictionary keys | bool bits */
  const uint8_t *validity;      /* LSB-first bitmap, NULL = all valid */
  const int32_t *offsets;       /* UTF8: length+1 offsets */
  int64_t values_bytes;         /* UTF8: size of `values` in bytes */
  const struct dfo_array *dictionary; /* DICTIONARY: value array */
} dfo_array;

/* Growable output column owned by the oracle (free with dfo_builder_free). */
typedef struct dfo_builder {
  dfo_array arr;        /* view on the storage below, valid until next mutation */
  uint8_t *vals; int64_t vals_cap;
  uint8_t *valid; int64_t valid_cap;  /* always materialised; arr.validity NULL if no nulls */
  int32_t *offs; int64_t offs_cap;
  int64_t nbytes;       /* utf8 bytes used */
} dfo_builder;

dfo_builder *dfo_builder_new(int32_t type, int32_t precision, int32_t scale);
void dfo_builder_free(dfo_builder *b);
const dfo_array *dfo_builder_array(dfo_builder *b);
void dfo_builder_append_cell(dfo_builder *b, const dfo_array *src, int64_t i); /* copies cell (dictionary resolved to value) */
void dfo_builder_append_null(dfo_builder *b);

int dfo_type_width(int32_t type); /* bytes per value for fixed-width types, 0 otherwise */
const char *dfo_last_error(void);

/* ---- a1: create_hashes (datafusion/common/src/hash_utils.rs:357-417) ---- */
void dfo_create_hashes(const dfo_array *const *cols, int k, int64_t n, uint64_t seed,
                       int force_collisions, uint64_t *out);

/* ---- a2-a6: HashJoinExec (physical-plan/src/joins/hash_join.rs, joins/utils.rs) ---- */
enum { DFO_JOIN_INNER = 0, DFO_JOIN_LEFT = 1, DFO_JOIN_RIGHT = 2, DFO_JOIN_FULL = 3,
       DFO_JOIN_LEFT_SEMI = 4, DFO_JOIN_RIGHT_SEMI = 5, DFO_JOIN_LEFT_ANTI = 6,
       DFO_JOIN_RIGHT_ANTI = 7 };

/* JoinFilter callback: evaluate the filter for n candidate pairs (indices into the
 * reference-order concatenated build batch and into probe batch `probe_batch`);
 * keep[i] = 1 keep, 0 drop (NULL filter result drops, arrow-select filter semantics). */
typedef void (*dfo_join_filter_fn)(void *ud, int32_t probe_batch, const int64_t *build_idx,
                                   const int64_t *probe_idx, int64_t n, uint8_t *keep);

typedef struct dfo_join_result {
  int64_t n;               /* output rows */
  int64_t *build_idx;      /* index into concatenated build batch (reference order = reversed
                              input-batch order, hash_join.rs:746,764); -1 = NULL */
  int64_t *probe_idx;      /* row inside probe batch probe_batch[i]; -1 = NULL */
  int32_t *probe_batch;    /* probe batch number; -1 for the final unmatched-build batch */
  int64_t n_batches;       /* output RecordBatch boundaries */
  int64_t *batch_offsets;  /* n_batches+1 */
} dfo_join_result;

/* build_keys[b*nkeys + c] = key column c of build batch b (input order); likewise probe. */
int dfo_hash_join(const dfo_array *const *build_keys, int n_build_batches,
                  const dfo_array *const *probe_keys, int n_probe_batches, int nkeys,
                  int join_type, int null_equals_null, int64_t batch_size,
                  int force_collisions, dfo_join_filter_fn filter, void *filter_ud,
                  dfo_join_result *out);
void dfo_join_result_free(dfo_join_result *r);

/* ---- a8: GroupValues (aggregates/group_values/{primitive,row,bytes}.rs) ---- */
typedef struct dfo_groups dfo_groups;
dfo_groups *dfo_groups_new(int nkeys, const int32_t *types, const int32_t *precisions,
                           const int32_t *scales);
void dfo_groups_free(dfo_groups *g);
/* intern: ids assigned in first-seen order; NULL is a key value of its own */
void dfo_groups_intern(dfo_groups *g, const dfo_array *const *cols, int64_t n, int64_t *out_ids);
int64_t dfo_groups_len(const dfo_groups *g);
const dfo_array *dfo_groups_emit(dfo_groups *g, int col); /* EmitTo::All view of key column */

/* ---- a9: GroupsAccumulator (expr/src/groups_accumulator.rs:78-164 + aggregate/{sum,average,count,min_max}.rs) ---- */
enum { DFO_AGG_SUM = 0, DFO_AGG_AVG = 1, DFO_AGG_COUNT = 2, DFO_AGG_MIN = 3, DFO_AGG_MAX = 4 };
typedef struct dfo_acc dfo_acc;
dfo_acc *dfo_acc_new(int kind, int32_t in_type, int32_t in_precision, int32_t in_scale);
void dfo_acc_free(dfo_acc *a);
/* values may be NULL for COUNT(*); filter is an optional BOOL array (NULL/false => skip row) */
int dfo_acc_update_batch(dfo_acc *a, const dfo_array *values, const int64_t *group_ids,
                         const dfo_array *opt_filter, int64_t n, int64_t total_num_groups);
int dfo_acc_merge_batch(dfo_acc *a, const dfo_array *const *states, int nstates,
                        const int64_t *group_ids, const dfo_array *opt_filter, int64_t n,
                        int64_t total_num_groups);
/* evaluate/state: EmitTo::All; returned arrays are owned by the accumulator */
int dfo_acc_evaluate(dfo_acc *a, const dfo_array **out);
int dfo_acc_state(dfo_acc *a, const dfo_array **out0, const dfo_array **out1, int *nstates);

/* ---- a12: PhysicalExpr evaluation kernels (expressions/binary.rs:259-315, datum.rs:28-58) ---- */
enum { DFO_OP_ADD = 0, DFO_OP_SUB = 1, DFO_OP_MUL = 2, DFO_OP_DIV = 3, DFO_OP_REM = 4,
       DFO_OP_EQ = 10, DFO_OP_NEQ = 11, DFO_OP_LT = 12, DFO_OP_LTEQ = 13, DFO_OP_GT = 14,
       DFO_OP_GTEQ = 15, DFO_OP_DISTINCT = 16, DFO_OP_NOT_DISTINCT = 17,
       DFO_OP_AND = 20, DFO_OP_OR = 21 };
/* scalar operands are length-1 arrays with *_scalar=1 (arrow Datum) */
int dfo_binary(int op, const dfo_array *l, int l_scalar, const dfo_array *r, int r_scalar,
               dfo_builder **out);
int dfo_not(const dfo_array *a, dfo_builder **out);
int dfo_is_null(const dfo_array *a, int negate, dfo_builder **out);
int dfo_negative(const dfo_array *a, dfo_builder **out);
int dfo_cast(const dfo_array *a, int32_t to_type, int32_t precision, int32_t scale, dfo_builder **out);
int dfo_in_list(const dfo_array *a, const dfo_array *list, int negated, dfo_builder **out);

/* ---- a13: sort_batch / lexsort_to_indices (sorts/sort.rs:584-609) ---- */
int dfo_lexsort_to_indices(const dfo_array *const *cols, int k, const uint8_t *descending,
                           const uint8_t *nulls_first, int64_t n, int64_t fetch /* -1 none */,
                           uint32_t *out, int64_t *n_out);

/* ---- a14: BatchPartitioner::partition_iter (repartition/mod.rs:148-221) ---- */
int dfo_hash_partition(const dfo_array *const *cols, int k, int64_t n, int num_partitions,
                       int force_collisions, uint32_t *indices_out, int64_t *counts_out);

/* ---- arrow-select kernels used by the path (take / filter) ---- */
int dfo_take(const dfo_array *a, const int64_t *indices /* -1 => null */, int64_t n, dfo_builder **out);
int dfo_filter(const dfo_array *a, const dfo_array *mask, dfo_builder **out);

/* ---- whole-query restatements used as cpu_baseline (dfo_tpch.c) ---- */
typedef struct dfo_q3_input {
  int64_t n_customer; const int64_t *c_custkey; const int8_t *c_mktsegment; int8_t segment_code;
  int64_t n_orders; const int64_t *o_orderkey; const int64_t *o_custkey; const int32_t *o_orderdate;
  const int32_t *o_shippriority; int32_t date_cut;
  int64_t n_lineitem; const int64_t *l_orderkey; const __int128 *l_extendedprice;
  const __int128 *l_discount; const int32_t *l_shipdate;
} dfo_q3_input;
typedef struct dfo_q3_output {
  int64_t n; int64_t *l_orderkey; __int128 *revenue; int32_t *o_orderdate; int32_t *o_shippriority;
} dfo_q3_output;
int dfo_tpch_q3(const dfo_q3_input *in, int target_partitions, int64_t batch_size, dfo_q3_output *out);
void dfo_q3_output_free(dfo_q3_output *o);

#ifdef __cplusplus
}
#endif
#endif
