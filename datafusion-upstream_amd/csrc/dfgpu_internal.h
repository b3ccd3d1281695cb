// dfgpu_internal.h -- host-side plumbing shared by the HIP translation units of libdfgpu.so.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/dfgpu.h"

namespace dfgpu {

struct Error : std::exception {
  dfgpu_status code; std::string msg;
  Error(dfgpu_status c, std::string m) : code(c), msg(std::move(m)) {}
  const char* what() const noexcept override { return msg.c_str(); }
};
[[noreturn]] void fail(dfgpu_status code, const char* fmt, ...);

#define HIP_CHECK(expr)                                                                         \
  do { hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) ::dfgpu::fail(e_ == hipErrorOutOfMemory ? DFGPU_RESOURCES_EXHAUSTED : DFGPU_INTERNAL, \
                                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define KERNEL_CHECK() HIP_CHECK(hipGetLastError())

}  // namespace dfgpu

// Device error flags raised by kernels (checked by the op that launched them).
enum : uint32_t { DFGPU_FLAG_DIV_ZERO = 1, DFGPU_FLAG_OVERFLOW = 2, DFGPU_FLAG_CAST = 4, DFGPU_FLAG_OOB = 8, DFGPU_FLAG_TABLE_FULL = 16, DFGPU_FLAG_STALLED = 32 };      // STALLED: a workgroup gave up waiting for a word another workgroup publishes (one-sweep sort passes)

namespace dfgpu { struct Buffer; }
struct dfgpu_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t copy_stream = nullptr;   // host -> device staging of the Parquet reader (created on first use): copies of the next column chunks run beside the decode kernels on `stream`
  std::string err;
  bool force_hash_collisions = false;
  bool first_seen_group_order = true;
  bool join_rank_index = true;
  bool join_lazy_build_rows = true;         // plan layer: an Inner HashJoinExec over a unique rank-indexed build looks the build rows up when a build-side column is read (dfgpu_join_probe_deferred)
  bool join_selection_output = true;        // plan layer: such a join whose build side contributes key columns only answers with the probe batch under a selection (dfgpu_join_probe_selection)
  bool join_rank_index_unsorted = true;     // unique integer keys over a dense domain in ANY order (a repartitioned or filtered primary-key column): bitmap + rank -> build row
  bool join_key_packing = true;
  bool group_run_detection = true;
  bool group_lazy_keys = true;              // run-mode GroupValues: the first batch's group keys stay (key columns, first rows) until somebody needs them stored (groups.hip)
  bool group_dictionary_canon = true;
  bool join_swap_small_semi = true;
  // radix-partitioned hash join (pjoin.hip): on/off, smallest build / probe batch that takes it, build rows per partition (<= 14000)
  uint64_t join_partitioned_hash_mask = ~0ull;       // tests: AND-ed onto the key hashes of the hashed mode, so that different keys collide
  bool join_partitioned_hashed = true;       // builds the integer mode of the partitioned join does not take (several key columns that do not pack, Utf8 / dictionary keys, null_equals_null) go through it on 64-bit key hashes
  bool join_partitioned_big = true;                            // builds beyond 2048 x join_partition_rows rows: up to 4096 partitions (else they decline to the global table)
  // membership-bitmap probes of unclustered keys through a key-range partition (pjoin.hip bp_probe).  OFF: measured slower than the random probes it replaces (TPC-H Q3 over
  // shuffled tables at SF100: partition 6.9 ms + probe 2.0 ms against 6.8 ms of random probes -- the LDS-staged scatter moves ~120 G rows/s whatever the row width)
  bool join_bitmap_partitioned = false; int64_t join_bitmap_partitioned_min_rows = 1 << 24;
  bool join_partitioned = true; int64_t join_partitioned_min_build = 1 << 20, join_partitioned_min_probe = 1 << 22, join_partition_rows = 14000;
  int64_t fused_aggregate_min_rows = 1 << 20;
  bool sort_packed_keys = true;     // large sorts over fixed-width keys: range-packed u64 keys + stable one-pass partition per digit (sort.hip)
  bool agg_partitioned = true, agg_partitioned_force = false; int64_t agg_partitioned_min_rows = 1 << 22;      // partitioned pre-aggregation (pagg.hip)
  bool agg_order_inverse_map = true;   // pre-aggregation with millions of partial rows: first-seen order through an inverse map over the input rows (pagg.hip) instead of a stable sort
  int64_t agg_pack_estimate_min_rows = 1 << 22;   // packed group keys: batches of at least this many rows take their value ranges from a sample (checked row by row while packing)
  bool pa_last_distinct = false;    // the last dfgpu_agg_preaggregate call emitted every key once
  const void* pa_sample_key = nullptr; const void* pa_sample_mask = nullptr; int64_t pa_sample_n = 0; uint64_t pa_sample[3] = {0, 0, 0};   // sample of a verdict-only dfgpu_agg_preaggregate call
  // 2..4 key columns packed into one u64 by that verdict-only call (kept for the call that follows on the same columns): key = sum((v - min + nullable) * stride), 0 in a nullable column's digit = NULL
  std::shared_ptr<dfgpu::Buffer> pa_pack; int pa_pack_n = 0; int64_t pa_pack_min[4] = {0, 0, 0, 0}; uint64_t pa_pack_stride[4] = {0, 0, 0, 0}, pa_pack_range[4] = {0, 0, 0, 0}; bool pa_pack_nullable[4] = {false, false, false, false};
  // row selection of the running operator (dfgpu_ctx_set_row_selection): expression kernels evaluate every row of full-length
  // columns but raise errors only for selected rows
  std::shared_ptr<dfgpu::Buffer> row_selection; int64_t row_selection_len = 0;
  // kernel error flags (overflow, divide by zero, cast range, index bounds): checked after the raising call, or -- inside one
  // poll of a plan's output stream -- once before the batch is handed out (saves a stream sync per kernel-level call)
  int defer_flag_checks = 0; bool flags_pending = false; std::string flags_what;
  uint32_t* d_flags = nullptr;      // device word for kernel error flags
  uint64_t* d_scratch64 = nullptr;  // 64 x u64 device scratch for counters / totals
  uint64_t* h_pinned = nullptr;     // 64 x u64 pinned host mirror (+ the mailbox's sequence word, a cache line of its own, at [MAIL_SEQ])
  // Host read-backs: a one-workgroup kernel posts the words to the pinned mirror and then a sequence number (system-scope release); the host polls the
  // sequence word.  The answer arrives a microsecond or two after the producing kernel retires -- no blit dispatch, no completion-signal wake-up of a stream
  // synchronisation (~20 us of idle device per read-back in the SF12.5 trace, profiles/r04_a_gaps_q3_sf12.5.txt).  Option "mailbox_readback" = 0: copy + hipStreamSynchronize.
  static constexpr int MAIL_SEQ = 384, MAIL_WORDS = 256, PINNED_WORDS = 512;       // words [0, 64): the scratch mirror; [0, MAIL_WORDS): the landing area of fetch_to_host
  bool mailbox_readback = true; uint64_t mail_seq = 0;
  int num_cus = 256;
  // optional per-kernel timing with HIP events on ctx->stream (bench.py roofline leg)
  // stream-ordered caching allocator: freed blocks are reused by later work on the same stream without going
  // back to the driver (hipMallocAsync pool growth cost ~200 ms per step at SF100); size classes 2^k and 1.5*2^k
  std::mutex* alloc_mu = nullptr;
  std::vector<std::pair<size_t, void*>>* free_blocks = nullptr;   // (class bytes, ptr)
  size_t cached_bytes = 0, live_bytes = 0;
  int64_t sort_packed_min_rows = 1 << 20;                      // smallest input the packed-key sort takes (below: byte planes of the encoded keys)
  int64_t sort_topk_words_min_rows = 1 << 23;                  // SortExec with fetch <= n / 16: from this many rows on (keys packing into a word with the row number) the radix select runs on the packed words; below, on byte planes
  int64_t sort_one_block_max_rows = 8192;                      // byte-plane sorts of at most this many rows (<= 8192): every pass inside one launch of one workgroup; 0 = off
  int64_t sort_onesweep_min_rows = 1 << 20;
  bool partition_two_round_staging = false;                    // radix partition into 513 .. 2048 partitions: every column staged in two rounds of half a tile (74 KB of LDS: two workgroups per CU).  Off: measured slower -- twice the barriers and predicated loads cost more than the second workgroup brings (ClickBench uniform: scatter 1.79 -> 2.44 ms; three-key Decimal128 group-by 1.49 -> 1.99 ms; 1 M Int64 groups 1.26 -> 1.43 ms)
  bool sort_payload_in_last_pass = false;                      // dfgpu_sort_take: the last one-sweep pass gathers the payload columns (else the caller's take() does, afterwards).  Off: measured slower (100 M rows, one 8-byte payload column: passes + gather 6.36 ms inside the last pass against 3.89 + 2.13 ms apart; both sides are bound by memory requests, the random reads in the scatter phase only slow the pass down)
  bool sort_onesweep_fused_finish = true;                      // the last one-sweep pass writes row numbers and rebuilt key columns instead of the words (no k_pk_finish pass)
  int sort_onesweep_rows = 16;                                  // word-mode sorts of 2^20 .. 2^30 rows: one launch per pass (look-back over published tile counts, sort.hip); rows per lane of a tile (8 or 16), 0 = the three-launch passes
  bool sort_fused_small_passes = true;                         // inputs below 2^20 rows: the per-pass scan is folded into the scatter (two launches per varying key byte instead of three)
  bool sort_estimate_ranges = true;                            // packed-key sort of >= 2^22 rows: value ranges from a sample, checked while encoding
  int64_t spm_merge_rows = (int64_t)1 << 25;                   // SortPreservingMergeExec: rows loaded over all inputs per merge step
  int64_t sort_spill_bytes = 0, sort_spill_ranges = 16;        // SortExec: bytes of input batches kept on the device before they are sorted and spilled to host memory as a run (0 = never); key ranges a run is cut into
  int64_t agg_spill_state_bytes = 0, agg_spill_ranges = 16;    // AggregateExec: state size (group table + accumulators) above which the state is spilled to host memory (0 = never); key ranges a spill is cut into
  int64_t memory_limit = 0;         // bytes of live device memory this ctx may hold (0 = none): ≙ a bounded MemoryPool (execution/src/memory_pool/pool.rs GreedyMemoryPool)
  bool collect_metrics = false;     // the plan layer meters its operators (device-time spans + row counts)
  struct Span { hipEvent_t start = nullptr, stop = nullptr; };
  std::vector<Span> spans;          // dfgpu_span_*
  std::mutex span_mu;               // a plan's partitions share one ctx and meter concurrently
  bool profile = false; std::string profile_only;
  std::vector<std::pair<std::string, int64_t>> sync_counts;   // host<->stream synchronisation points by cause (profiling only)
  void count_sync(const char* why) { if (!profile) return; for (auto& kv : sync_counts) if (kv.first == why) { kv.second++; return; } sync_counts.emplace_back(why, 1); }
  struct ProfRec { const char* name; hipEvent_t start, stop; };
  std::vector<ProfRec> prof;
};

namespace dfgpu {

struct Buffer {
  void* ptr = nullptr; size_t bytes = 0; dfgpu_ctx* ctx = nullptr; bool owned = true;
  std::shared_ptr<Buffer> parent;   // keeps a sliced parent alive
  std::shared_ptr<void> owner;      // borrowed memory: the caller's release callback fires when the last buffer / view over it is gone
  ~Buffer();
};
using BufferPtr = std::shared_ptr<Buffer>;
BufferPtr alloc_buffer(dfgpu_ctx* ctx, size_t bytes, bool zero = false);
BufferPtr borrow_buffer(const void* ptr, size_t bytes);

}  // namespace dfgpu

namespace dfgpu { struct DeferredIds;
// Order statistics of an integer column without NULLs (arrays are immutable, so they are a memo like null_count): known once a pass has looked (k_check_increasing), or derived --
// a gather of a sorted column through strictly ascending indices is sorted and lies inside the source's bounds (exact = false: lo / hi bound the values, not necessarily tight).
// The join build's rank index asks for them; a base-table key column answers from the memo, its filtered / joined descendants from the derivation.
// lo / hi: the first and the last element as measured (order_stats_measure) -- the minimum and maximum of a SORTED column, bounds of everything gathered from it by ascending
// row numbers; for an unsorted column they bound nothing (using them as the domain of an unsorted rank index was tried in round 4 and faulted: the selected keys' own min / max
// are read back there)
struct OrderStats { bool sorted = false, repeats = false, exact = false; int64_t lo = 0, hi = 0; };
}
struct dfgpu_array {
  std::atomic<int64_t> refs{1};
  dfgpu_ctx* ctx = nullptr;
  int32_t type = 0, precision = 0, scale = 0, key_type = 0;
  int64_t length = 0;
  int64_t null_count = -1;
  dfgpu::BufferPtr values, validity, offsets;
  int64_t values_bytes = 0;
  dfgpu_array* dictionary = nullptr;   // retained
  // host mirror of a length-1 array (scalar Datum) so kernels can take it by value
  bool has_host_scalar = false; unsigned char host_scalar[16] = {0}; bool host_scalar_valid = true;
  dfgpu_array_desc dict_desc{};        // storage for describe()
  // group ids of dfgpu_groups_intern_deferred whose values are not written yet: the recipe that computes them (dense dictionary keys).
  // The accumulator entry points either compute the ids inside their own pass or write them out first (materialize_ids).
  std::shared_ptr<dfgpu::DeferredIds> deferred_ids;
  // an index array known to be 0, 1, .., length - 1 (a compaction that kept every row, the probe indices of a join whose probe rows
  // all matched once): gathering through it is the identity
  bool identity = false;
  bool base_column = false;     // imported / wrapped by the caller rather than computed by an operator
  std::shared_ptr<const dfgpu::OrderStats> order_stats;      // read / written through order_stats_get / order_stats_set (a lock: plan partitions share arrays)
};

namespace dfgpu {

inline int type_width(int32_t t) {
  switch (t) {
    case DFGPU_INT8: case DFGPU_UINT8: return 1;
    case DFGPU_INT16: case DFGPU_UINT16: return 2;
    case DFGPU_INT32: case DFGPU_UINT32: case DFGPU_FLOAT32: case DFGPU_DATE32: return 4;
    case DFGPU_INT64: case DFGPU_UINT64: case DFGPU_FLOAT64: return 8;
    case DFGPU_DECIMAL128: return 16;
    default: return 0;
  }
}
inline bool is_signed_int(int32_t t) { return t == DFGPU_INT8 || t == DFGPU_INT16 || t == DFGPU_INT32 || t == DFGPU_INT64; }
inline bool is_unsigned_int(int32_t t) { return t == DFGPU_UINT8 || t == DFGPU_UINT16 || t == DFGPU_UINT32 || t == DFGPU_UINT64; }
inline bool is_float(int32_t t) { return t == DFGPU_FLOAT32 || t == DFGPU_FLOAT64; }
inline size_t bitmap_bytes(int64_t n) { return (size_t)((n + 63) / 64) * 8; }   // always whole u64 words
inline int32_t logical_type(const dfgpu_array* a) { return a->type == DFGPU_DICTIONARY ? a->dictionary->type : a->type; }

dfgpu_array* new_array(dfgpu_ctx* ctx, int32_t type, int64_t length, int32_t precision = 0, int32_t scale = 0);
dfgpu_array* new_fixed(dfgpu_ctx* ctx, int32_t type, int64_t length, int32_t precision = 0, int32_t scale = 0, bool with_validity = false);
struct ArrayHolder {   // RAII release on exception paths
  dfgpu_array* a = nullptr;
  ArrayHolder() = default; explicit ArrayHolder(dfgpu_array* x) : a(x) {}
  ~ArrayHolder() { if (a) dfgpu_array_release(a); }
  dfgpu_array* release() { dfgpu_array* x = a; a = nullptr; return x; }
  dfgpu_array* get() const { return a; }
  ArrayHolder(ArrayHolder&& o) noexcept : a(o.a) { o.a = nullptr; }
  ArrayHolder(const ArrayHolder&) = delete; ArrayHolder& operator=(const ArrayHolder&) = delete;
};

// Read back the kernel error flags (synchronises) and raise the matching DataFusionError analogue.
void check_flags(dfgpu_ctx* ctx, const char* what);
void flush_flags(dfgpu_ctx* ctx);                          // the deferred check, now
uint64_t read_scratch(dfgpu_ctx* ctx, int slot);          // sync + D2H of d_scratch64[slot]
const uint64_t* read_scratch_range(dfgpu_ctx* ctx, int first, int count);      // the same for `count` consecutive slots: one copy, one wait
void zero_scratch(dfgpu_ctx* ctx);
std::shared_ptr<const OrderStats> order_stats_get(const dfgpu_array* a);
void order_stats_set(const dfgpu_array* a, const OrderStats& st);
std::shared_ptr<const OrderStats> order_stats_measure(dfgpu_ctx* ctx, const dfgpu_array* a);      // join.hip: one streaming pass + one read-back; nullptr for columns that are not plain integers without NULLs
void order_stats_through_take(dfgpu_ctx* ctx, const dfgpu_array* values, const dfgpu_array* indices, dfgpu_array* out);      // select.hip
// `bytes` (a multiple of 4, <= 512) of device memory to ctx->h_pinned + h_word (64-bit words), then wait until they are there: the one way the host reads a device word back
void fetch_to_pinned(dfgpu_ctx* ctx, int h_word, const void* d_src, size_t bytes);
// `bytes` of device memory into host memory `dst`, waited for: through the mailbox up to 2 KB (a multiple of 4), by copy + stream synchronisation beyond
void fetch_to_host(dfgpu_ctx* ctx, void* dst, const void* d_src, size_t bytes);

// Device view of a column passed to kernels by value.
struct ColView {
  int32_t type; int32_t width;          // logical type (dictionary resolved) and byte width
  const void* values; const uint64_t* validity;   // validity as u64 words (may be null)
  const int32_t* offsets;               // utf8
  const void* keys; const uint64_t* key_validity; int32_t key_type;  // dictionary indirection (keys != null)
  int32_t precision, scale;
};
ColView make_view(const dfgpu_array* a);

constexpr int MAX_KEYS = 8;
struct KeySet { int32_t n; ColView c[MAX_KEYS]; };
// dense composite group map over dictionary key columns (groups.hip): code -> canonical id of the dictionary value -> sum(id * stride) -> map
struct DenseCol { const void* keys; const uint64_t* key_valid; const uint32_t* canon; const uint64_t* dict_valid; int64_t dict_len; uint32_t n_ids, stride; int32_t key_type; };
struct DenseCols { int32_t n; DenseCol c[MAX_KEYS]; };
struct DeferredIds {
  int kind = 0;          // 0: dense dictionary keys (dc, dense_map); 1: run numbers of a clustered batch (heads, prefix, base)
  DenseCols dc{}; int32_t key_type = 0; BufferPtr mask, dense_map; std::vector<BufferPtr> keep;      // keep: code columns and canonical id tables
  BufferPtr heads, prefix; uint32_t base = 0;          // id of row i = base + prefix[i / 64] + popcount(heads[i / 64] up to bit i % 64) - 1
};
void materialize_ids(dfgpu_ctx* ctx, const dfgpu_array* ids);         // no-op for ordinary arrays
KeySet make_keyset(const dfgpu_array* const* cols, int32_t n);

// mask (BOOL array) -> effective selection bitmap words (values & validity), null if no mask
BufferPtr effective_mask(dfgpu_ctx* ctx, const dfgpu_array* mask, int64_t expect_len);

// primitives implemented in scan.hip
void exclusive_scan_u32(dfgpu_ctx* ctx, const uint32_t* in, uint64_t* out, int64_t n, uint64_t* d_total /*device, may be null*/);
void exclusive_scan_u32_inplace32(dfgpu_ctx* ctx, uint32_t* data, int64_t n, uint64_t* d_total);

// select.hip helpers reused by other ops
dfgpu_array* take_impl(dfgpu_ctx* ctx, const dfgpu_array* values, const void* idx, int idx_width, const uint64_t* idx_validity, int64_t n_out);
dfgpu_array* mask_to_indices_impl(dfgpu_ctx* ctx, const uint64_t* bits, int64_t n);
dfgpu_array* mask_to_indices_checked(dfgpu_ctx* ctx, const uint64_t* bits, int64_t n, int check_slot, uint64_t* check_value);
dfgpu_array* mask_to_indices_uncounted(dfgpu_ctx* ctx, const uint64_t* bits, int64_t n, uint64_t* d_count = nullptr);      // d_count: device word that receives the number of entries written      // n entries, the first popcount(bits) written; no read-back (internal rank -> row tables)
int64_t count_set_bits(dfgpu_ctx* ctx, const uint64_t* bits, int64_t n);
inline const uint64_t* row_selection_words(dfgpu_ctx* ctx, int64_t n) { return ctx->row_selection && ctx->row_selection_len == n ? (const uint64_t*)ctx->row_selection->ptr : nullptr; }

void launch_iota_u32(dfgpu_ctx* ctx, uint32_t* out, int64_t n, uint32_t start);
void launch_set_bits_prefix(dfgpu_ctx* ctx, uint64_t* bits, int64_t m);   // bits[0..m) = 1

// hash.hip
void hash_keys_device(dfgpu_ctx* ctx, const dfgpu_array* const* cols, int32_t k, uint64_t seed, uint64_t* out);

// sort.hip: stable sort of (u32 key, u32 value) pairs on `bits` low bits of the key
void radix_sort_pairs_u32(dfgpu_ctx* ctx, uint32_t* keys, uint32_t* vals, int64_t n, int bits);

// RAII: brackets the kernel launches in its scope with HIP events when profiling is enabled
struct KernelTimer {
  dfgpu_ctx* c; int idx = -1;
  KernelTimer(dfgpu_ctx* ctx, const char* name) : c(ctx) {
    if (!c->profile || (!c->profile_only.empty() && c->profile_only != name)) return;
    dfgpu_ctx::ProfRec r{name, nullptr, nullptr};
    if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
    (void)hipEventRecord(r.start, c->stream); c->prof.push_back(r); idx = (int)c->prof.size() - 1;
  }
  ~KernelTimer() { if (idx >= 0) (void)hipEventRecord(c->prof[idx].stop, c->stream); }
};

template <typename F>
dfgpu_status guard(dfgpu_ctx* ctx, F&& f, bool ctx_optional = false) {
  if (!ctx && !ctx_optional) return DFGPU_INVALID_ARGUMENT;          // every entry point that needs its ctx: a NULL handle is an argument error, never a crash of the host process
  try { f(); return DFGPU_OK; }
  catch (const Error& e) { if (ctx) ctx->err = e.msg; return e.code; }
  catch (const std::bad_alloc&) { if (ctx) ctx->err = "host allocation failed"; return DFGPU_RESOURCES_EXHAUSTED; }
  catch (const std::exception& e) { if (ctx) ctx->err = e.what(); return DFGPU_INTERNAL; }
}

// One workgroup per `block` elements.  Kernels that index one element per lane pass no cap (gridDim.x may be
// up to 2^31-1); only kernels written as grid-stride loops pass `max_blocks`.
inline int grid_for(int64_t n, int block, int64_t max_blocks = 0x7FFFFFFF) {
  int64_t g = (n + block - 1) / block; if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  if (g > 0x7FFFFFFFll) fail(DFGPU_NOT_IMPLEMENTED, "launch of %lld workgroups exceeds the grid limit", (long long)g);
  return (int)g;
}

}  // namespace dfgpu
