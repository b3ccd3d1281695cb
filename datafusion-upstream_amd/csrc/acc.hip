// acc.hip -- a9: GroupsAccumulator (SUM / AVG / COUNT / MIN / MAX) for AggregateExec on gfx950.
//
// Reference: expr/src/groups_accumulator.rs:78-164; physical-expr/src/aggregate/{sum,average,count,min_max}.rs;
// groups_accumulator/{prim_op,accumulate}.rs; aggregate/utils.rs:55-125 (DecimalAverager).
//
// State lives in HBM, one slot per group: value/sum (8 B, or 16 B lo|hi for Decimal128), count (AVG, COUNT),
// seen byte (NullState).  Two update paths:
//   * few groups (<= 8, e.g. TPC-H Q1's 4): every lane keeps all group partials in registers over a
//     grid-stride loop (compare-select, no divergence, no atomics in the loop), then wave shuffle
//     reduce -> LDS -> one global atomic per workgroup and group.
//   * many groups (Q3: 1e6, Q18: 1.5e8): one device-scope atomic per row; Decimal128 wrapping add is two
//     64-bit atomics with the carry taken from the value the low add returned (exact mod 2^128).
// Float64 SUM is order dependent in the reference too (sequential add in row order, prim_op.rs:101-109):
// the contract is <= 1e-9 relative.
#include "int128.h"
#include <algorithm>
#include <tuple>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace dfgpu {
constexpr uint32_t GID_NONE = 0xFFFFFFFFu;
enum { CLS_I64 = 0, CLS_U64 = 1, CLS_F64 = 2, CLS_I128 = 3 };
}
using namespace dfgpu;

struct dfgpu_acc {
  dfgpu_ctx* ctx = nullptr; int kind = 0;
  int32_t in_type = 0, in_precision = 0, in_scale = 0;
  int32_t state_type = 0, state_precision = 0, state_scale = 0;
  int32_t out_type = 0, out_precision = 0, out_scale = 0;
  int cls = 0; int width = 8;
  int64_t n = 0, cap = 0;
  BufferPtr vals, counts, seen;     // vals: width B/group; counts: 8 B/group (AVG u64, COUNT i64); seen: 1 B/group
  // states adopted from a fully pre-aggregated first batch (acc_adopt_identity) and not copied yet: the accumulator holds n groups whose values (SUM / MIN / MAX) or counts
  // (COUNT) are the adopted array's buffer, every group seen.  Evaluating it hands that buffer out again; the first update / merge copies it into vals / counts / seen
  bool lazy = false; BufferPtr lazy_vals, lazy_counts;
};

namespace dfgpu {

__device__ inline bool filter_pass(const uint64_t* fbits, const uint64_t* fvalid, int64_t i) {
  return fbits == nullptr || (bit_get(fbits, i) && valid_at(fvalid, i));
}
__device__ inline i128 acc_cell_int(const ColView& c, int64_t r) {
  switch (c.type) {
    case DFGPU_INT8: return ((const int8_t*)c.values)[r]; case DFGPU_INT16: return ((const int16_t*)c.values)[r];
    case DFGPU_INT32: case DFGPU_DATE32: return ((const int32_t*)c.values)[r]; case DFGPU_INT64: return ((const int64_t*)c.values)[r];
    case DFGPU_UINT8: return ((const uint8_t*)c.values)[r]; case DFGPU_UINT16: return ((const uint16_t*)c.values)[r];
    case DFGPU_UINT32: return ((const uint32_t*)c.values)[r]; case DFGPU_UINT64: return (i128)((const uint64_t*)c.values)[r];
    case DFGPU_DECIMAL128: return load_i128(c.values, r);
    default: return 0;
  }
}
__device__ inline double acc_cell_f64(const ColView& c, int64_t r) { return c.type == DFGPU_FLOAT32 ? (double)((const float*)c.values)[r] : ((const double*)c.values)[r]; }

__device__ inline void atomic_add_i128(uint64_t* slot, i128 v) {
  uint64_t lo = (uint64_t)(u128)v, hi = (uint64_t)((u128)v >> 64);
  uint64_t old = atomicAdd((unsigned long long*)&slot[0], (unsigned long long)lo);
  uint64_t carry = (uint64_t)(old + lo < old);
  if (hi + carry) atomicAdd((unsigned long long*)&slot[1], (unsigned long long)(hi + carry));
}
__device__ inline void atomic_min_f64(double* p, double x, bool is_min) {
  unsigned long long* q = (unsigned long long*)p; unsigned long long old = *q;
  for (;;) { double cur = __longlong_as_double((long long)old); bool repl = is_min ? (cur > x) : (cur < x); if (!repl) return;
    unsigned long long prev = atomicCAS(q, old, (unsigned long long)__double_as_longlong(x)); if (prev == old) return; old = prev; }
}

// ------------------------------------------------------------ many groups: one atomic per row
__global__ void __launch_bounds__(BLOCK) k_acc_update(int kind, int cls, ColView v, int has_values, const uint32_t* gids, const uint64_t* fbits, const uint64_t* fvalid,
                                                      int64_t n, int64_t total, void* vals, uint64_t* counts, uint8_t* seen, int count_only_valid, uint32_t* flags) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    uint32_t g = gids ? gids[i] : 0u;
    if (g == GID_NONE || !filter_pass(fbits, fvalid, i)) continue;
    if ((int64_t)g >= total) { atomicOr(flags, DFGPU_FLAG_OOB); continue; }
    int64_t r = i; bool ok = !has_values || cell_resolve(v, i, &r);
    if (kind == DFGPU_AGG_COUNT) { if (ok || !count_only_valid) atomicAdd((unsigned long long*)&counts[g], 1ull); continue; }
    if (!ok) continue;
    seen[g] = 1;
    if (kind == DFGPU_AGG_AVG) atomicAdd((unsigned long long*)&counts[g], 1ull);
    if (kind == DFGPU_AGG_SUM || kind == DFGPU_AGG_AVG) {
      if (cls == CLS_F64) unsafeAtomicAdd((double*)vals + g, acc_cell_f64(v, r));
      else if (cls == CLS_I128) atomic_add_i128((uint64_t*)vals + 2 * (int64_t)g, acc_cell_int(v, r));
      else atomicAdd((unsigned long long*)vals + g, (unsigned long long)(uint64_t)acc_cell_int(v, r));     // wrapping i64 / u64
    } else {
      bool is_min = kind == DFGPU_AGG_MIN;
      if (cls == CLS_F64) atomic_min_f64((double*)vals + g, acc_cell_f64(v, r), is_min);
      // read first: an extreme only moves one way, so a (possibly stale) value that already beats x proves the atomic is not needed --
      // with r rows per group only ~ln r of them are records
      else if (cls == CLS_U64) { unsigned long long x = (unsigned long long)(uint64_t)acc_cell_int(v, r), cur = ((const unsigned long long*)vals)[g];
                                 if (is_min) { if (x < cur) atomicMin((unsigned long long*)vals + g, x); } else if (x > cur) atomicMax((unsigned long long*)vals + g, x); }
      else if (cls == CLS_I64) { long long x = (long long)acc_cell_int(v, r), cur = ((const long long*)vals)[g];
                                 if (is_min) { if (x < cur) atomicMin((long long*)vals + g, x); } else if (x > cur) atomicMax((long long*)vals + g, x); }
      else { long long hi = (long long)(acc_cell_int(v, r) >> 64); if (is_min) atomicMin((long long*)vals + 2 * (int64_t)g + 1, hi); else atomicMax((long long*)vals + 2 * (int64_t)g + 1, hi); }   // i128 pass 1: high word
    }
  }
}
// Decimal128 MIN/MAX pass 2: among rows whose high word equals the group's extreme high word, extreme of the low word
__global__ void __launch_bounds__(BLOCK) k_acc_minmax128_lo(int is_min, ColView v, const uint32_t* gids, const uint64_t* fbits, const uint64_t* fvalid, int64_t n, uint64_t* vals) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    uint32_t g = gids ? gids[i] : 0u; int64_t r;
    if (g == GID_NONE || !filter_pass(fbits, fvalid, i) || !cell_resolve(v, i, &r)) continue;
    i128 x = acc_cell_int(v, r);
    if ((uint64_t)((u128)x >> 64) != vals[2 * (int64_t)g + 1]) continue;
    unsigned long long lo = (unsigned long long)(uint64_t)(u128)x;
    if (is_min) atomicMin((unsigned long long*)&vals[2 * (int64_t)g], lo); else atomicMax((unsigned long long*)&vals[2 * (int64_t)g], lo);
  }
}
__global__ void k_minmax128_prepare(int is_min, uint64_t* vals, const uint64_t* old_hi, int64_t total) {   // reset low words whose high word moved
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < total && vals[2 * g + 1] != old_hi[g]) vals[2 * g] = is_min ? ~0ull : 0ull;
}
__global__ void k_add_count_delta(uint64_t* dst, const uint64_t* now, const uint64_t* before, int64_t total) {
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; if (g < total) dst[g] += now[g] - before[g];
}
__global__ void k_copy_hi(const uint64_t* vals, uint64_t* hi, int64_t total) { int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; if (g < total) hi[g] = vals[2 * g + 1]; }

// ------------------------------------------------------------ few groups: register partials
template <typename T> __device__ inline T wave_sum_any(T v) { return wave_sum(v); }
template <> __device__ inline i128 wave_sum_any<i128>(i128 v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    uint64_t lo = __shfl_xor((unsigned long long)(uint64_t)(u128)v, d, 64), hi = __shfl_xor((unsigned long long)(uint64_t)((u128)v >> 64), d, 64);
    v = (i128)((u128)v + (((u128)hi << 64) | lo));
  }
  return v;
}
constexpr int SMALL_G = 8;
template <typename T, int CLS>
__global__ void __launch_bounds__(BLOCK) k_acc_small(int kind, ColView v, int has_values, const uint32_t* gids, const uint64_t* fbits, const uint64_t* fvalid,
                                                     int64_t n, int G, void* vals, uint64_t* counts, uint8_t* seen, int count_only_valid) {
  T acc[SMALL_G]; uint64_t cnt[SMALL_G];
#pragma unroll
  for (int k = 0; k < SMALL_G; k++) { acc[k] = 0; cnt[k] = 0; }
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    uint32_t g = gids ? gids[i] : 0u;
    if (g == GID_NONE || !filter_pass(fbits, fvalid, i)) continue;
    int64_t r = i; bool ok = !has_values || cell_resolve(v, i, &r);
    if (kind == DFGPU_AGG_COUNT) { if (!(ok || !count_only_valid)) continue; }
    else if (!ok) continue;
    T x = 0;
    if (kind != DFGPU_AGG_COUNT) { if constexpr (CLS == CLS_F64) x = acc_cell_f64(v, r); else x = (T)acc_cell_int(v, r); }
#pragma unroll
    for (int k = 0; k < SMALL_G; k++) { bool m = g == (uint32_t)k; acc[k] += m ? x : (T)0; cnt[k] += m ? 1u : 0u; }
  }
  __shared__ T s_acc[BLOCK / WAVE][SMALL_G]; __shared__ uint64_t s_cnt[BLOCK / WAVE][SMALL_G];
  int wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < SMALL_G; k++) { T a = wave_sum_any<T>(acc[k]); uint64_t c = wave_sum(cnt[k]); if (lane_id() == 0) { s_acc[wave][k] = a; s_cnt[wave][k] = c; } }
  __syncthreads();
  if (threadIdx.x < G) {
    int k = threadIdx.x; T a = 0; uint64_t c = 0;
    for (int w = 0; w < BLOCK / WAVE; w++) { a += s_acc[w][k]; c += s_cnt[w][k]; }
    if (c) {
      if (kind == DFGPU_AGG_COUNT) { atomicAdd((unsigned long long*)&counts[k], (unsigned long long)c); return; }
      seen[k] = 1;
      if (kind == DFGPU_AGG_AVG) atomicAdd((unsigned long long*)&counts[k], (unsigned long long)c);
      if constexpr (CLS == CLS_F64) unsafeAtomicAdd((double*)vals + k, a);
      else if constexpr (CLS == CLS_I128) atomic_add_i128((uint64_t*)vals + 2 * k, a);
      else atomicAdd((unsigned long long*)vals + k, (unsigned long long)a);
    }
  }
}

// ------------------------------------------------------------ few groups, several SUM / AVG accumulators in ONE pass
// TPC-H Q1 updates 7 sum-like accumulators over the same 6 groups: one pass per accumulator re-reads the group ids and the selection
// 7 times and SUM(x) / AVG(x) read x twice.  dfgpu_acc_update_batch_multi hands all accumulators of a batch over at once; runs of
// compatible ones (same value class, plain fixed-width values, same filter, <= 8 groups) share a pass: ids and filter are read once,
// a value column that repeats in the next slot is read once, every (slot, group) partial lives in registers.
constexpr int MULTI_MAX = 4;
struct MultiArgs { const void* vals[MULTI_MAX]; const uint64_t* valid[MULTI_MAX]; void* out_vals[MULTI_MAX]; uint64_t* out_counts[MULTI_MAX]; uint8_t* out_seen[MULTI_MAX]; int kind[MULTI_MAX]; };
// PLAIN: no value validity and no filter validity anywhere -> the rows' counts are the same for every slot (one counter set), and the
// loop body is branch-free with index-clamped loads, R rows per lane in flight; a load behind a branch waits for the branch's own loads
// first (s_waitcnt), which makes the generic one-row loop latency bound
template <typename T, int CLS, int NACC, bool PLAIN>
__global__ void __launch_bounds__(BLOCK) k_acc_small_multi(MultiArgs a, const uint32_t* gids, const uint64_t* fbits, const uint64_t* fvalid, int64_t n, int G) {
  constexpr int NC = PLAIN ? 1 : NACC;
  T acc[NACC][SMALL_G]; uint32_t cnt[NC][SMALL_G];
#pragma unroll
  for (int s = 0; s < NACC; s++)
#pragma unroll
    for (int k = 0; k < SMALL_G; k++) { acc[s][k] = (T)0; if (s < NC) cnt[s][k] = 0; }
  if constexpr (PLAIN) {
    constexpr int R = 2;
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t base = (int64_t)blockIdx.x * BLOCK + threadIdx.x; base < n; base += stride * R) {
      uint32_t g[R]; T x[R][NACC];
#pragma unroll
      for (int r = 0; r < R; r++) {
        int64_t i = base + r * stride; bool in = i < n; int64_t ii = in ? i : n - 1;
        uint32_t gg = gids[ii];
        bool pass = in && (fbits == nullptr || ((fbits[ii >> 6] >> (ii & 63)) & 1ull));
        g[r] = pass ? gg : GID_NONE;
#pragma unroll
        for (int s = 0; s < NACC; s++) {
          if (s > 0 && a.vals[s] == a.vals[s - 1]) x[r][s] = x[r][s - 1];
          else if constexpr (CLS == CLS_I128) x[r][s] = load_i128(a.vals[s], ii); else x[r][s] = ((const T*)a.vals[s])[ii];
        }
      }
#pragma unroll
      for (int r = 0; r < R; r++)
#pragma unroll
        for (int k = 0; k < SMALL_G; k++) {
          bool m = g[r] == (uint32_t)k; cnt[0][k] += m ? 1u : 0u;
#pragma unroll
          for (int s = 0; s < NACC; s++) acc[s][k] += m ? x[r][s] : (T)0;
        }
    }
  } else
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    uint32_t g = gids[i];
    if (g == GID_NONE || !filter_pass(fbits, fvalid, i)) continue;
    T x[NACC]; bool ok[NACC];
#pragma unroll
    for (int s = 0; s < NACC; s++) {
      if (s > 0 && a.vals[s] == a.vals[s - 1]) x[s] = x[s - 1];               // SUM(x), AVG(x): one load
      else if constexpr (CLS == CLS_I128) x[s] = load_i128(a.vals[s], i); else x[s] = ((const T*)a.vals[s])[i];
      ok[s] = valid_at(a.valid[s], i);
    }
#pragma unroll
    for (int s = 0; s < NACC; s++)
#pragma unroll
      for (int k = 0; k < SMALL_G; k++) { bool m = ok[s] && g == (uint32_t)k; acc[s][k] += m ? x[s] : (T)0; cnt[s][k] += m ? 1u : 0u; }
  }
  __shared__ T s_acc[BLOCK / WAVE][NACC][SMALL_G]; __shared__ uint32_t s_cnt[BLOCK / WAVE][NACC][SMALL_G];
  int wave = threadIdx.x >> 6;
#pragma unroll
  for (int s = 0; s < NACC; s++)
#pragma unroll
    for (int k = 0; k < SMALL_G; k++) { T v = wave_sum_any<T>(acc[s][k]); uint32_t c = wave_sum(cnt[s < NC ? s : 0][k]); if (lane_id() == 0) { s_acc[wave][s][k] = v; s_cnt[wave][s][k] = c; } }
  __syncthreads();
  if (threadIdx.x < NACC * SMALL_G) {
    int s = threadIdx.x / SMALL_G, k = threadIdx.x % SMALL_G;
    if (k < G) {
      T v = (T)0; uint64_t c = 0;
      for (int w = 0; w < BLOCK / WAVE; w++) { v += s_acc[w][s][k]; c += s_cnt[w][s][k]; }
      if (c) {
        a.out_seen[s][k] = 1;
        if (a.kind[s] == DFGPU_AGG_AVG) atomicAdd((unsigned long long*)&a.out_counts[s][k], (unsigned long long)c);
        if constexpr (CLS == CLS_F64) unsafeAtomicAdd((double*)a.out_vals[s] + k, v);
        else if constexpr (CLS == CLS_I128) atomic_add_i128((uint64_t*)a.out_vals[s] + 2 * k, v);
        else atomicAdd((unsigned long long*)a.out_vals[s] + k, (unsigned long long)v);
      }
    }
  }
}

// ------------------------------------------------------------ many groups, sum-like kinds: runs of equal adjacent group ids are combined first
// Clustered input (GROUP BY over a fact table in key order, ids from groups.hip's run numbering) puts the rows of a group in
// neighbouring lanes: a segmented wave scan adds them up and only the last lane of each run issues the atomic -- 4x fewer atomics
// for TPC-H Q18's sub-aggregate (600 M rows, 150 M groups: 35 -> ms).  A wave without two equal neighbours skips the scan.
template <typename T> __device__ inline T shfl_up_any(T v, int d) { return __shfl_up(v, d, 64); }
template <> __device__ inline i128 shfl_up_any<i128>(i128 v, int d) {
  uint64_t lo = __shfl_up((unsigned long long)(uint64_t)(u128)v, d, 64), hi = __shfl_up((unsigned long long)(uint64_t)((u128)v >> 64), d, 64);
  return (i128)(((u128)hi << 64) | lo);
}
template <typename T, int CLS>
__global__ void __launch_bounds__(BLOCK) k_acc_update_add(int kind, ColView v, int has_values, const uint32_t* gids, const uint64_t* fbits, const uint64_t* fvalid,
                                                          int64_t n, int64_t total, void* vals, uint64_t* counts, uint8_t* seen, int count_only_valid, uint32_t* flags) {
  int lane = lane_id();
  int64_t n_round = (n + WAVE - 1) / WAVE * WAVE;                  // whole waves enter every iteration: the shuffles need all lanes
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * BLOCK) {
    uint32_t g = GID_NONE; bool act = false; T x = (T)0;
    if (i < n) {
      g = gids[i];
      act = g != GID_NONE && filter_pass(fbits, fvalid, i);
      if (act && (int64_t)g >= total) { atomicOr(flags, DFGPU_FLAG_OOB); act = false; }
      if (act) {
        int64_t r = i; bool ok = !has_values || cell_resolve(v, i, &r);
        if (kind == DFGPU_AGG_COUNT) act = ok || !count_only_valid;
        else { act = ok; if (ok) { if constexpr (CLS == CLS_F64) x = acc_cell_f64(v, r); else x = (T)acc_cell_int(v, r); } }
      }
    }
    uint32_t c = act ? 1u : 0u;
    uint32_t gp = __shfl_up(g, 1, 64); int ap = __shfl_up((int)act, 1, 64);
    bool head = !(lane > 0 && act && ap && gp == g);                 // inactive lanes are their own (empty) runs
    uint64_t hm = ballot64(head);
    if (hm != ~0ull) {                                               // some run has >= 2 lanes
      int start = 63 - __clzll((long long)(hm & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull))));     // first lane of my run
#pragma unroll
      for (int d = 1; d < WAVE; d <<= 1) { T ox = shfl_up_any<T>(x, d); uint32_t oc = __shfl_up(c, d, 64); if (lane - d >= start) { x += ox; c += oc; } }
    }
    uint32_t gn = __shfl_down(g, 1, 64); int an = __shfl_down((int)act, 1, 64);
    bool tail = act && (lane == 63 || !an || gn != g);
    if (!tail) continue;
    if (kind == DFGPU_AGG_COUNT) { atomicAdd((unsigned long long*)&counts[g], (unsigned long long)c); continue; }
    seen[g] = 1;
    if (kind == DFGPU_AGG_AVG) atomicAdd((unsigned long long*)&counts[g], (unsigned long long)c);
    if constexpr (CLS == CLS_F64) unsafeAtomicAdd((double*)vals + g, x);
    else if constexpr (CLS == CLS_I128) atomic_add_i128((uint64_t*)vals + 2 * (int64_t)g, x);
    else atomicAdd((unsigned long long*)vals + g, (unsigned long long)x);
  }
}

// The same segmented combine for the plain case -- no filter, values of exactly the state type without validity or dictionary -- with
// ADD_ROWS slabs of 64 rows per wave whose loads are all issued up front, unconditionally (rows past the end re-read the last row and
// are dropped by their group id): the general kernel's one conditional row per lane per iteration leaves HBM latency exposed.
constexpr int ADD_ROWS = 4;
// RUNS: the ids are run numbers that were never written out (DeferredIds kind 1): row i's id = base + prefix[i / 64] + popcount of the
// head bits up to i -- one 8-B word and one 4-B prefix per 64-row slab instead of 4 B per row.
struct RunIds { const uint64_t* heads; const uint32_t* prefix; uint32_t base; };
template <typename T, int CLS, bool HAS_VALUES, bool RUNS>
__global__ void __launch_bounds__(BLOCK) k_acc_add_plain(int kind, const T* values, const uint32_t* gids, RunIds ri, int64_t n, int64_t total, void* vals, uint64_t* counts, uint8_t* seen, uint32_t* flags) {
  int lane = lane_id();
  int64_t base = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * (WAVE * ADD_ROWS);
  if (base >= n) return;                                             // wave-uniform
  uint32_t gs[ADD_ROWS]; T xs[ADD_ROWS];
#pragma unroll
  for (int r = 0; r < ADD_ROWS; r++) {
    int64_t i = base + r * WAVE + lane, ic = i < n ? i : n - 1;
    if constexpr (RUNS) { int64_t w = ic >> 6; gs[r] = ri.base + ri.prefix[w] + (uint32_t)__popcll(ri.heads[w] & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull))) - 1u; }      // base is a multiple of 64: bit = lane
    else gs[r] = gids[ic];
    if constexpr (HAS_VALUES) xs[r] = values[ic]; else xs[r] = (T)0;
    if (i >= n) gs[r] = GID_NONE;
  }
#pragma unroll
  for (int r = 0; r < ADD_ROWS; r++) {
    uint32_t g = gs[r]; T x = xs[r];
    bool act = g != GID_NONE;
    if (act && (int64_t)g >= total) { atomicOr(flags, DFGPU_FLAG_OOB); act = false; }
    if (!act) x = (T)0;
    uint32_t c = act ? 1u : 0u;
    uint32_t gp = __shfl_up(g, 1, 64); int ap = __shfl_up((int)act, 1, 64);
    bool head = !(lane > 0 && act && ap && gp == g);
    uint64_t hm = ballot64(head);
    if (hm != ~0ull) {
      int start = 63 - __clzll((long long)(hm & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull))));
#pragma unroll
      for (int d = 1; d < WAVE; d <<= 1) {
        if (ballot64(lane - d >= start) == 0) break;                 // every run of the slab is shorter than d lanes (wave-uniform): short runs stop early
        T ox = shfl_up_any<T>(x, d); uint32_t oc = __shfl_up(c, d, 64); if (lane - d >= start) { x += ox; c += oc; }
      }
    }
    uint32_t gn = __shfl_down(g, 1, 64); int an = __shfl_down((int)act, 1, 64);
    bool tail = act && (lane == 63 || !an || gn != g);
    // (measured: storing whole runs -- a piece that starts on a head bit and ends where its run ends -- without atomics made TPC-H Q18's 150 M runs slower, 2.8 -> 3.1 ms:
    // the L2 atomics were not the bound, the extra head-bit loads and three partial-line stores per group cost more)
    if (tail) {
      if (kind == DFGPU_AGG_COUNT) atomicAdd((unsigned long long*)&counts[g], (unsigned long long)c);
      else {
        seen[g] = 1;
        if (kind == DFGPU_AGG_AVG) atomicAdd((unsigned long long*)&counts[g], (unsigned long long)c);
        if constexpr (CLS == CLS_F64) unsafeAtomicAdd((double*)vals + g, x);
        else if constexpr (CLS == CLS_I128) atomic_add_i128((uint64_t*)vals + 2 * (int64_t)g, x);
        else atomicAdd((unsigned long long*)vals + g, (unsigned long long)x);
      }
    }
  }
}

// ------------------------------------------------------------ repeated groups: per-workgroup LDS cache in front of the global atomics
// Atomics on one address serialise in L2 (~2 ns each): 10 M rows of a Zipf(1.1) key spent 36 ms in k_acc_update, 1000 uniform
// groups 4 ms.  Each workgroup therefore keeps a 4096-entry, 4-probe cache (group id tag, partial value, row count) in LDS:
// the first groups a workgroup meets claim entries and accumulate with LDS atomics; a group that finds both probes taken
// falls through to the global atomic.  Hot groups are met first with overwhelming probability, so their traffic stays in
// LDS and reaches HBM as one atomic per (workgroup, group) at the end.  Chosen by launch_update when the batch has >= 8 rows
// per known group on average (uniform high-cardinality batches keep the direct path).
constexpr int ACC_CACHE = 4096;        // entries per workgroup: 64 KB of LDS for 8-byte states, 96 KB for i128
constexpr int ACC_PROBES = 4;
constexpr uint32_t TAG_EMPTY = 0xFFFFFFFFu;
enum { OP_ADD = 0, OP_MIN = 1, OP_MAX = 2 };

template <typename T, int OP> __device__ inline T op_identity() {
  if constexpr (OP == OP_ADD) return (T)0;
  else if constexpr (std::is_same<T, double>::value) return OP == OP_MIN ? __longlong_as_double(0x7FF0000000000000ll) : __longlong_as_double((long long)0xFFF0000000000000ull);
  else if constexpr (std::is_same<T, long long>::value) return OP == OP_MIN ? INT64_MAX : INT64_MIN;
  else return OP == OP_MIN ? (T)~0ull : (T)0;
}
template <typename T, int OP> __device__ inline void lds_apply(T* p, T x) {
  if constexpr (std::is_same<T, i128>::value) {        // OP_ADD only: two 64-bit adds with carry, as in HBM
    uint64_t* q = (uint64_t*)p; uint64_t lo = (uint64_t)(u128)x, hi = (uint64_t)((u128)x >> 64);
    uint64_t old = atomicAdd((unsigned long long*)&q[0], (unsigned long long)lo); uint64_t carry = (uint64_t)(old + lo < old);
    if (hi + carry) atomicAdd((unsigned long long*)&q[1], (unsigned long long)(hi + carry));
  } else if constexpr (OP == OP_ADD) {
    if constexpr (std::is_same<T, double>::value) unsafeAtomicAdd(p, x); else atomicAdd((unsigned long long*)p, (unsigned long long)x);
  } else if constexpr (std::is_same<T, double>::value) atomic_min_f64(p, x, OP == OP_MIN);
  else { if (OP == OP_MIN) atomicMin(p, x); else atomicMax(p, x); }
}
template <typename T, int CLS, int OP> __device__ inline void global_apply(int kind, uint32_t g, T x, uint64_t c, void* vals, uint64_t* counts, uint8_t* seen) {
  if (kind == DFGPU_AGG_COUNT) { atomicAdd((unsigned long long*)&counts[g], (unsigned long long)c); return; }
  seen[g] = 1;
  if (kind == DFGPU_AGG_AVG) atomicAdd((unsigned long long*)&counts[g], (unsigned long long)c);
  if constexpr (CLS == CLS_I128) atomic_add_i128((uint64_t*)vals + 2 * (int64_t)g, x);
  else if constexpr (OP == OP_ADD) { if constexpr (CLS == CLS_F64) unsafeAtomicAdd((double*)vals + g, x); else atomicAdd((unsigned long long*)vals + g, (unsigned long long)x); }
  else if constexpr (CLS == CLS_F64) atomic_min_f64((double*)vals + g, x, OP == OP_MIN);
  else { T cur = ((const T*)vals)[g]; if (OP == OP_MIN) { if (x < cur) atomicMin((T*)vals + g, x); } else if (x > cur) atomicMax((T*)vals + g, x); }     // read first, as in k_acc_update
}
// PLAIN: no filter and values of exactly T without validity or dictionary (or COUNT(*)): CACHED_ROWS rows per lane whose group ids and
// values are loaded together, unconditionally, before any of them touches the cache.
constexpr int CACHED_ROWS = 4;
template <typename T, int CLS, int OP, bool PLAIN>
__global__ void __launch_bounds__(BLOCK) k_acc_cached(int kind, ColView v, int has_values, const uint32_t* gids, const uint64_t* fbits, const uint64_t* fvalid,
                                                      int64_t n, int64_t total, void* vals, uint64_t* counts, uint8_t* seen, int count_only_valid, uint32_t* flags) {
  __shared__ uint32_t s_tag[ACC_CACHE]; __shared__ uint32_t s_cnt[ACC_CACHE]; __shared__ T s_val[ACC_CACHE];
  for (int s = threadIdx.x; s < ACC_CACHE; s += BLOCK) { s_tag[s] = TAG_EMPTY; s_cnt[s] = 0; s_val[s] = op_identity<T, OP>(); }
  __syncthreads();
  if constexpr (PLAIN) {
    const T* vp = (const T*)v.values;
    for (int64_t base = (int64_t)blockIdx.x * BLOCK * CACHED_ROWS + threadIdx.x; base < n; base += (int64_t)gridDim.x * BLOCK * CACHED_ROWS) {
      uint32_t gq[CACHED_ROWS]; T xq[CACHED_ROWS];
#pragma unroll
      for (int q = 0; q < CACHED_ROWS; q++) {
        int64_t i = base + (int64_t)q * BLOCK, ic = i < n ? i : n - 1;
        gq[q] = gids[ic]; xq[q] = has_values ? vp[ic] : op_identity<T, OP>();
        if (i >= n) gq[q] = GID_NONE;
      }
#pragma unroll
      for (int q = 0; q < CACHED_ROWS; q++) {
        uint32_t g = gq[q]; T x = xq[q];
        if (g == GID_NONE) continue;
        if ((int64_t)g >= total) { atomicOr(flags, DFGPU_FLAG_OOB); continue; }
        uint32_t h = (g * 0x9E3779B1u) >> 20; int slot = -1;
#pragma unroll
        for (int p = 0; p < ACC_PROBES && slot < 0; p++) {
          int s = (int)((h + p) & (ACC_CACHE - 1)); uint32_t t = s_tag[s];
          if (t == TAG_EMPTY) { t = atomicCAS(&s_tag[s], TAG_EMPTY, g); if (t == TAG_EMPTY) t = g; }
          if (t == g) slot = s;
        }
        if (slot >= 0) { if (kind != DFGPU_AGG_COUNT) lds_apply<T, OP>(&s_val[slot], x); atomicAdd(&s_cnt[slot], 1u); }
        else global_apply<T, CLS, OP>(kind, g, x, 1, vals, counts, seen);
      }
    }
  } else
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    uint32_t g = gids[i];
    if (g == GID_NONE || !filter_pass(fbits, fvalid, i)) continue;
    if ((int64_t)g >= total) { atomicOr(flags, DFGPU_FLAG_OOB); continue; }
    int64_t r = i; bool ok = !has_values || cell_resolve(v, i, &r);
    if (kind == DFGPU_AGG_COUNT) { if (!(ok || !count_only_valid)) continue; }
    else if (!ok) continue;
    T x = op_identity<T, OP>();
    if (kind != DFGPU_AGG_COUNT) { if constexpr (CLS == CLS_F64) x = acc_cell_f64(v, r); else x = (T)acc_cell_int(v, r); }
    uint32_t h = (g * 0x9E3779B1u) >> 20; int slot = -1;
#pragma unroll
    for (int p = 0; p < ACC_PROBES && slot < 0; p++) {
      int s = (int)((h + p) & (ACC_CACHE - 1)); uint32_t t = s_tag[s];
      if (t == TAG_EMPTY) { t = atomicCAS(&s_tag[s], TAG_EMPTY, g); if (t == TAG_EMPTY) t = g; }
      if (t == g) slot = s;
    }
    if (slot >= 0) { if (kind != DFGPU_AGG_COUNT) lds_apply<T, OP>(&s_val[slot], x); atomicAdd(&s_cnt[slot], 1u); }
    else global_apply<T, CLS, OP>(kind, g, x, 1, vals, counts, seen);
  }
  __syncthreads();
  for (int s = threadIdx.x; s < ACC_CACHE; s += BLOCK) { uint32_t g = s_tag[s], c = s_cnt[s]; if (g != TAG_EMPTY && c) global_apply<T, CLS, OP>(kind, g, s_val[s], c, vals, counts, seen); }
}

// ------------------------------------------------------------ emit
__global__ void __launch_bounds__(BLOCK) k_seen_to_bits(const uint8_t* seen, int64_t n, uint64_t* bits) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  uint64_t m = ballot64(i < n && seen[i]);
  if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) bits[i >> 6] = m;
}
__global__ void k_narrow(const uint64_t* vals, int64_t n, int32_t type, void* out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  switch (type) {
    case DFGPU_INT8: case DFGPU_UINT8: ((uint8_t*)out)[i] = (uint8_t)vals[i]; break;
    case DFGPU_INT16: case DFGPU_UINT16: ((uint16_t*)out)[i] = (uint16_t)vals[i]; break;
    case DFGPU_INT32: case DFGPU_UINT32: case DFGPU_DATE32: ((uint32_t*)out)[i] = (uint32_t)vals[i]; break;
    case DFGPU_FLOAT32: ((float*)out)[i] = (float)__longlong_as_double((long long)vals[i]); break;
    default: ((uint64_t*)out)[i] = vals[i];
  }
}
__global__ void k_avg_f64(const double* sums, const uint64_t* counts, const uint8_t* seen, int64_t n, double* out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = seen[i] ? sums[i] / (double)counts[i] : 0.0;     // average.rs:166
}
// DecimalAverager::avg (aggregate/utils.rs:108-124)
__global__ void k_avg_dec(const uint64_t* sums, const uint64_t* counts, const uint8_t* seen, int64_t n, i128 factor, int precision, uint64_t* out, uint32_t* flags) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  i128 v = 0;
  if (seen[i]) {
    i128 prod;
    if (!mul128_checked(load_i128(sums, i), factor, &prod)) atomicOr(flags, DFGPU_FLAG_OVERFLOW);
    else { v = sdiv128(prod, (i128)counts[i], nullptr); if (!decimal_fits(v, precision)) { atomicOr(flags, DFGPU_FLAG_OVERFLOW); v = 0; } }
  }
  store_i128(out, i, v);
}
__global__ void k_fill64(uint64_t* p, int64_t from, int64_t to, uint64_t v, int stride, int off) {
  int64_t i = from + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < to) p[i * stride + off] = v;
}

static int imin(int a, int b) { return a < b ? a : b; }

// the three state arrays of the new groups cleared in ONE launch (a query's result-sized accumulators: three hipMemsetAsync calls were three dispatches of ~5 us each)
__global__ void __launch_bounds__(BLOCK) k_acc_clear(uint64_t* vals, int64_t vwords, uint64_t* counts, int64_t cwords, uint8_t* seen, int64_t sbytes) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < vwords + cwords + sbytes; i += (int64_t)gridDim.x * BLOCK) {
    if (i < vwords) vals[i] = 0; else if (i < vwords + cwords) counts[i - vwords] = 0; else seen[i - vwords - cwords] = 0;
  }
}

static void acc_materialize(dfgpu_acc* a) {
  if (!a->lazy) return;
  dfgpu_ctx* ctx = a->ctx; const int64_t total = a->n;
  a->vals = alloc_buffer(ctx, (size_t)total * a->width); a->counts = alloc_buffer(ctx, (size_t)total * 8); a->seen = alloc_buffer(ctx, (size_t)total); a->cap = total;
  if (a->lazy_vals) HIP_CHECK(hipMemcpyAsync(a->vals->ptr, a->lazy_vals->ptr, (size_t)total * a->width, hipMemcpyDeviceToDevice, ctx->stream));
  else HIP_CHECK(hipMemsetAsync(a->vals->ptr, 0, (size_t)total * a->width, ctx->stream));
  if (a->lazy_counts) HIP_CHECK(hipMemcpyAsync(a->counts->ptr, a->lazy_counts->ptr, (size_t)total * 8, hipMemcpyDeviceToDevice, ctx->stream));
  else HIP_CHECK(hipMemsetAsync(a->counts->ptr, 0, (size_t)total * 8, ctx->stream));
  HIP_CHECK(hipMemsetAsync(a->seen->ptr, 1, (size_t)total, ctx->stream));
  a->lazy = false; a->lazy_vals.reset(); a->lazy_counts.reset();
}
static void acc_resize(dfgpu_acc* a, int64_t total) {
  dfgpu_ctx* ctx = a->ctx;
  acc_materialize(a);                     // every update / merge path comes through here before it touches the state arrays
  if (total <= a->n) return;
  if (total > a->cap) {
    int64_t nc = a->cap ? a->cap : 1024; while (nc < total) nc *= 2;
    BufferPtr nv = alloc_buffer(ctx, (size_t)nc * a->width), ncnt = alloc_buffer(ctx, (size_t)nc * 8), ns = alloc_buffer(ctx, (size_t)nc);
    if (a->n) { HIP_CHECK(hipMemcpyAsync(nv->ptr, a->vals->ptr, (size_t)a->n * a->width, hipMemcpyDeviceToDevice, ctx->stream));
                HIP_CHECK(hipMemcpyAsync(ncnt->ptr, a->counts->ptr, (size_t)a->n * 8, hipMemcpyDeviceToDevice, ctx->stream));
                HIP_CHECK(hipMemcpyAsync(ns->ptr, a->seen->ptr, (size_t)a->n, hipMemcpyDeviceToDevice, ctx->stream)); }
    a->vals = nv; a->counts = ncnt; a->seen = ns; a->cap = nc;
  }
  int64_t add = total - a->n;
  const bool minmax = a->kind == DFGPU_AGG_MIN || a->kind == DFGPU_AGG_MAX;
  if (!minmax && add <= ((int64_t)1 << 22)) {         // small: one launch for all three arrays
    const int64_t vw = add * (a->width / 8), cw = add, sb = add;
    hipLaunchKernelGGL(k_acc_clear, dim3(grid_for(vw + cw + sb, BLOCK, ctx->num_cus * 8)), dim3(BLOCK), 0, ctx->stream, (uint64_t*)((char*)a->vals->ptr + a->n * a->width), vw,
                       (uint64_t*)((char*)a->counts->ptr + a->n * 8), cw, (uint8_t*)a->seen->ptr + a->n, sb);
    KERNEL_CHECK();
    a->n = total;
    return;
  }
  HIP_CHECK(hipMemsetAsync((char*)a->counts->ptr + a->n * 8, 0, (size_t)add * 8, ctx->stream));
  HIP_CHECK(hipMemsetAsync((char*)a->seen->ptr + a->n, 0, (size_t)add, ctx->stream));
  if (a->kind == DFGPU_AGG_MIN || a->kind == DFGPU_AGG_MAX) {
    bool mn = a->kind == DFGPU_AGG_MIN; uint64_t init;
    // starting values: NATIVE::MAX for MIN, NATIVE::MIN for MAX (min_max.rs:102-139); integer state is widened to 64 bit
    if (a->cls == CLS_F64) { double d = a->state_type == DFGPU_FLOAT32 ? 3.40282346638528859812e+38 : 1.7976931348623157e308; if (!mn) d = -d; memcpy(&init, &d, 8); }
    else if (a->cls == CLS_U64) { uint64_t mx = a->state_type == DFGPU_UINT8 ? 0xFFull : a->state_type == DFGPU_UINT16 ? 0xFFFFull : a->state_type == DFGPU_UINT32 ? 0xFFFFFFFFull : ~0ull; init = mn ? mx : 0; }
    else { int bits = a->cls == CLS_I128 ? 64 : type_width(a->state_type) * 8; int64_t mx = bits == 64 ? INT64_MAX : ((1ll << (bits - 1)) - 1); int64_t mnv = bits == 64 ? INT64_MIN : -(1ll << (bits - 1)); init = (uint64_t)(mn ? mx : mnv); }
    dim3 grid(grid_for(add, BLOCK)), block(BLOCK);
    if (a->cls == CLS_I128) { hipLaunchKernelGGL(k_fill64, grid, block, 0, ctx->stream, (uint64_t*)a->vals->ptr, a->n, total, init, 2, 1);
                              hipLaunchKernelGGL(k_fill64, grid, block, 0, ctx->stream, (uint64_t*)a->vals->ptr, a->n, total, mn ? ~0ull : 0ull, 2, 0); }
    else hipLaunchKernelGGL(k_fill64, grid, block, 0, ctx->stream, (uint64_t*)a->vals->ptr, a->n, total, init, 1, 0);
    KERNEL_CHECK();
  } else HIP_CHECK(hipMemsetAsync((char*)a->vals->ptr + a->n * a->width, 0, (size_t)add * a->width, ctx->stream));
  a->n = total;
}

static void launch_update(dfgpu_acc* a, int kind, int cls, const dfgpu_array* values, const dfgpu_array* gids, const dfgpu_array* filt, int64_t total, void* vals) {
  dfgpu_ctx* ctx = a->ctx; int64_t n = gids->length;
  if (gids->type != DFGPU_UINT32) fail(DFGPU_INVALID_ARGUMENT, "group ids must be a UINT32 array");
  // run numbers that were never written out are derived inside the plain accumulate kernel; every other kernel reads stored ids
  std::shared_ptr<DeferredIds> runs = gids->deferred_ids && gids->deferred_ids->kind == 1 ? gids->deferred_ids : nullptr;
  {
    bool sl = kind == DFGPU_AGG_SUM || kind == DFGPU_AGG_AVG || kind == DFGPU_AGG_COUNT;
    bool plain = sl && !filt && total > SMALL_G && n < 8 * total &&
                 (kind == DFGPU_AGG_COUNT ? (!values || !values->validity)
                                          : (values && !values->validity && values->type == (cls == CLS_I128 ? DFGPU_DECIMAL128 : cls == CLS_F64 ? DFGPU_FLOAT64 : values->type) &&
                                             (cls == CLS_I128 || cls == CLS_F64 || values->type == DFGPU_INT64 || values->type == DFGPU_UINT64)));
    if (!plain) runs.reset();
    if (!runs) materialize_ids(ctx, gids);
  }
  if (values && values->length != n) fail(DFGPU_INVALID_ARGUMENT, "values (%lld rows) and group ids (%lld rows) differ in length", (long long)values->length, (long long)n);
  if (filt && (filt->type != DFGPU_BOOL || filt->length != n)) fail(DFGPU_INVALID_ARGUMENT, "opt_filter must be a Boolean array of the batch length");
  if (!n) return;
  ColView v{}; if (values) v = make_view(values);
  const uint64_t* fb = filt ? (const uint64_t*)filt->values->ptr : nullptr; const uint64_t* fv = filt && filt->validity ? (const uint64_t*)filt->validity->ptr : nullptr;
  const uint32_t* g = (const uint32_t*)gids->values->ptr;
  int blocks = grid_for(n, BLOCK * 8, ctx->num_cus * 8);
  KernelTimer kt_(ctx, "k_acc_update");
  bool sumlike = kind == DFGPU_AGG_SUM || kind == DFGPU_AGG_AVG || kind == DFGPU_AGG_COUNT;
  if (sumlike && total <= SMALL_G) {
#define SMALL(T, C) hipLaunchKernelGGL((k_acc_small<T, C>), dim3(blocks), dim3(BLOCK), 0, ctx->stream, kind, v, values ? 1 : 0, g, fb, fv, n, (int)total, vals, (uint64_t*)a->counts->ptr, (uint8_t*)a->seen->ptr, 1)
    if (cls == CLS_F64) SMALL(double, CLS_F64); else if (cls == CLS_I128) SMALL(i128, CLS_I128); else SMALL(uint64_t, CLS_U64);
#undef SMALL
  } else if (n >= 8 * total && !(cls == CLS_I128 && !sumlike)) {
    int cblocks = grid_for(n, BLOCK * 32, ctx->num_cus * 4);         // >= 8192 rows per workgroup amortise the cache flush
    // plain: the values are stored exactly as the kernel's T (MIN / MAX state keeps the input type; SUM / AVG of a narrower integer widens per row)
    bool cplain = !filt && (!values ? kind == DFGPU_AGG_COUNT
                                    : (kind == DFGPU_AGG_COUNT ? !values->validity
                                       : (!values->validity && values->type != DFGPU_DICTIONARY &&
                                          (cls == CLS_F64 ? values->type == DFGPU_FLOAT64 : cls == CLS_I128 ? values->type == DFGPU_DECIMAL128 : cls == CLS_U64 ? values->type == DFGPU_UINT64 : values->type == DFGPU_INT64))));
    const bool cvals = values && kind != DFGPU_AGG_COUNT;
#define CACHED(T, C, O) do { if (cplain) hipLaunchKernelGGL((k_acc_cached<T, C, O, true>), dim3(cblocks), dim3(BLOCK), 0, ctx->stream, kind, v, cvals ? 1 : 0, g, fb, fv, n, total, vals, (uint64_t*)a->counts->ptr, (uint8_t*)a->seen->ptr, 1, ctx->d_flags); \
                             else hipLaunchKernelGGL((k_acc_cached<T, C, O, false>), dim3(cblocks), dim3(BLOCK), 0, ctx->stream, kind, v, values ? 1 : 0, g, fb, fv, n, total, vals, (uint64_t*)a->counts->ptr, (uint8_t*)a->seen->ptr, 1, ctx->d_flags); } while (0)
    if (sumlike) { if (cls == CLS_F64) CACHED(double, CLS_F64, OP_ADD); else if (cls == CLS_I128) CACHED(i128, CLS_I128, OP_ADD); else CACHED(unsigned long long, CLS_U64, OP_ADD); }
    else if (kind == DFGPU_AGG_MIN) { if (cls == CLS_F64) CACHED(double, CLS_F64, OP_MIN); else if (cls == CLS_U64) CACHED(unsigned long long, CLS_U64, OP_MIN); else CACHED(long long, CLS_I64, OP_MIN); }
    else { if (cls == CLS_F64) CACHED(double, CLS_F64, OP_MAX); else if (cls == CLS_U64) CACHED(unsigned long long, CLS_U64, OP_MAX); else CACHED(long long, CLS_I64, OP_MAX); }
#undef CACHED
  } else if (sumlike && !filt && (kind == DFGPU_AGG_COUNT ? (!values || !values->validity)
                                  : (!values->validity && values->type == (cls == CLS_I128 ? DFGPU_DECIMAL128 : cls == CLS_F64 ? DFGPU_FLOAT64 : values->type) &&
                                     (cls == CLS_I128 || cls == CLS_F64 || values->type == DFGPU_INT64 || values->type == DFGPU_UINT64)))) {
    int pblocks = grid_for(n, BLOCK * ADD_ROWS);
    const void* vp = kind == DFGPU_AGG_COUNT ? nullptr : values->values->ptr;
    RunIds ri{}; if (runs) { ri.heads = (const uint64_t*)runs->heads->ptr; ri.prefix = (const uint32_t*)runs->prefix->ptr; ri.base = runs->base; }
#define PLAIN(T, C, HV) do { if (runs) hipLaunchKernelGGL((k_acc_add_plain<T, C, HV, true>), dim3(pblocks), dim3(BLOCK), 0, ctx->stream, kind, (const T*)vp, g, ri, n, total, vals, (uint64_t*)a->counts->ptr, (uint8_t*)a->seen->ptr, ctx->d_flags); \
                             else hipLaunchKernelGGL((k_acc_add_plain<T, C, HV, false>), dim3(pblocks), dim3(BLOCK), 0, ctx->stream, kind, (const T*)vp, g, ri, n, total, vals, (uint64_t*)a->counts->ptr, (uint8_t*)a->seen->ptr, ctx->d_flags); } while (0)
    if (kind == DFGPU_AGG_COUNT) { PLAIN(unsigned long long, CLS_U64, false); } else if (cls == CLS_F64) { PLAIN(double, CLS_F64, true); } else if (cls == CLS_I128) { PLAIN(i128, CLS_I128, true); } else { PLAIN(unsigned long long, CLS_U64, true); }
#undef PLAIN
  } else if (sumlike) {
#define ADD(T, C) hipLaunchKernelGGL((k_acc_update_add<T, C>), dim3(blocks), dim3(BLOCK), 0, ctx->stream, kind, v, values ? 1 : 0, g, fb, fv, n, total, vals, (uint64_t*)a->counts->ptr, (uint8_t*)a->seen->ptr, 1, ctx->d_flags)
    if (cls == CLS_F64) ADD(double, CLS_F64); else if (cls == CLS_I128) ADD(i128, CLS_I128); else ADD(unsigned long long, CLS_U64);
#undef ADD
  } else {
    hipLaunchKernelGGL(k_acc_update, dim3(blocks), dim3(BLOCK), 0, ctx->stream, kind, cls, v, values ? 1 : 0, g, fb, fv, n, total, vals, (uint64_t*)a->counts->ptr, (uint8_t*)a->seen->ptr, 1, ctx->d_flags);
  }
  KERNEL_CHECK();
}

}  // namespace dfgpu

extern "C" {

dfgpu_status dfgpu_acc_new(dfgpu_ctx* ctx, int32_t kind, int32_t in_type, int32_t p, int32_t s, dfgpu_acc** out) {
  return guard(ctx, [&] {
    std::unique_ptr<dfgpu_acc> a(new dfgpu_acc()); a->ctx = ctx; a->kind = kind; a->in_type = in_type; a->in_precision = p; a->in_scale = s;
    auto cls_of = [](int32_t t) { return t == DFGPU_DECIMAL128 ? CLS_I128 : is_float(t) ? CLS_F64 : is_unsigned_int(t) ? CLS_U64 : CLS_I64; };
    switch (kind) {
      case DFGPU_AGG_COUNT: a->state_type = a->out_type = DFGPU_INT64; a->cls = CLS_I64; break;
      case DFGPU_AGG_SUM:       // sum_return_type, expr/src/type_coercion/aggregates.rs:397-416
        if (is_signed_int(in_type)) a->state_type = DFGPU_INT64; else if (is_unsigned_int(in_type)) a->state_type = DFGPU_UINT64; else if (is_float(in_type)) a->state_type = DFGPU_FLOAT64;
        else if (in_type == DFGPU_DECIMAL128) { a->state_type = DFGPU_DECIMAL128; a->state_precision = imin(38, p + 10); a->state_scale = s; }
        else fail(DFGPU_NOT_IMPLEMENTED, "SUM over type %d", in_type);
        a->out_type = a->state_type; a->out_precision = a->state_precision; a->out_scale = a->state_scale; a->cls = cls_of(a->state_type); break;
      case DFGPU_AGG_AVG:       // avg_return_type / avg_sum_type, aggregates.rs:455-505
        if (in_type == DFGPU_DECIMAL128) { a->state_type = DFGPU_DECIMAL128; a->state_precision = imin(38, p + 10); a->state_scale = s; a->out_type = DFGPU_DECIMAL128; a->out_precision = imin(38, p + 4); a->out_scale = imin(38, s + 4); }
        else if (in_type == DFGPU_FLOAT64) a->state_type = a->out_type = DFGPU_FLOAT64;
        else fail(DFGPU_INVALID_ARGUMENT, "AVG input must be coerced to Float64 or Decimal128 (got %d)", in_type);
        a->cls = cls_of(a->state_type); break;
      case DFGPU_AGG_MIN: case DFGPU_AGG_MAX:
        if (!type_width(in_type)) fail(DFGPU_NOT_IMPLEMENTED, "MIN/MAX over type %d", in_type);
        a->state_type = a->out_type = in_type; a->state_precision = a->out_precision = p; a->state_scale = a->out_scale = s; a->cls = cls_of(in_type); break;
      default: fail(DFGPU_INVALID_ARGUMENT, "unknown aggregate kind %d", kind);
    }
    a->width = a->cls == CLS_I128 ? 16 : 8;
    *out = a.release();
  });
}
void dfgpu_acc_free(dfgpu_acc* a) { delete a; }
int64_t dfgpu_acc_size(const dfgpu_acc* a) { return a ? a->cap * (a->width + 9) : 0; }

dfgpu_status dfgpu_acc_update_batch(dfgpu_ctx* ctx, dfgpu_acc* a, const dfgpu_array* values, const dfgpu_array* gids, const dfgpu_array* filt, int64_t total) {
  return guard(ctx, [&] {
    if (!a || !gids) fail(DFGPU_INVALID_ARGUMENT, "acc_update_batch: null argument");
    if (gids->length == 0) { if (a->lazy && total <= a->n) return;          // nothing to grow and nothing to add: adopted states stay adopted
      acc_resize(a, total); return; }     // self.values.resize(total_num_groups, ..) with nothing to accumulate
    if (a->kind != DFGPU_AGG_COUNT) {
      if (!values) fail(DFGPU_INVALID_ARGUMENT, "acc_update_batch: values required");
      int32_t lt = logical_type(values);
      bool ok = (a->kind == DFGPU_AGG_MIN || a->kind == DFGPU_AGG_MAX) ? lt == a->in_type
              : (a->cls == CLS_I128 ? lt == DFGPU_DECIMAL128 : a->cls == CLS_F64 ? is_float(lt) : a->cls == CLS_U64 ? is_unsigned_int(lt) : is_signed_int(lt));
      if (!ok) fail(DFGPU_INVALID_ARGUMENT, "accumulator created for input type %d got values of type %d", a->in_type, lt);
    }
    acc_resize(a, total);
    bool mm128 = (a->kind == DFGPU_AGG_MIN || a->kind == DFGPU_AGG_MAX) && a->cls == CLS_I128;
    BufferPtr old_hi;
    if (mm128 && total) { old_hi = alloc_buffer(ctx, (size_t)total * 8); hipLaunchKernelGGL(k_copy_hi, dim3(grid_for(total, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)a->vals->ptr, (uint64_t*)old_hi->ptr, total); }
    launch_update(a, a->kind, a->cls, values, gids, filt, total, a->vals ? a->vals->ptr : nullptr);
    if (mm128 && total && gids->length) {
      hipLaunchKernelGGL(k_minmax128_prepare, dim3(grid_for(total, BLOCK)), dim3(BLOCK), 0, ctx->stream, a->kind == DFGPU_AGG_MIN ? 1 : 0, (uint64_t*)a->vals->ptr, (const uint64_t*)old_hi->ptr, total);
      ColView v = make_view(values);
      hipLaunchKernelGGL(k_acc_minmax128_lo, dim3(grid_for(gids->length, BLOCK * 8, ctx->num_cus * 8)), dim3(BLOCK), 0, ctx->stream, a->kind == DFGPU_AGG_MIN ? 1 : 0, v, (const uint32_t*)gids->values->ptr,
                         filt ? (const uint64_t*)filt->values->ptr : nullptr, filt && filt->validity ? (const uint64_t*)filt->validity->ptr : nullptr, gids->length, (uint64_t*)a->vals->ptr);
      KERNEL_CHECK();
    }
    check_flags(ctx, "acc_update_batch");
  });
}

// can accumulator a take part in a shared small-group pass over `values`?
static bool multi_ok(const dfgpu_acc* a, const dfgpu_array* values, int64_t total) {
  if (!a || !values || total > SMALL_G || (a->kind != DFGPU_AGG_SUM && a->kind != DFGPU_AGG_AVG)) return false;
  if (values->type == DFGPU_DICTIONARY) return false;
  return a->cls == CLS_F64 ? values->type == DFGPU_FLOAT64 : a->cls == CLS_I128 ? values->type == DFGPU_DECIMAL128 : (values->type == DFGPU_INT64 || values->type == DFGPU_UINT64);
}
dfgpu_status dfgpu_acc_update_batch_multi(dfgpu_ctx* ctx, dfgpu_acc* const* accs, const dfgpu_array* const* values, const dfgpu_array* const* filters, int32_t n_accs,
                                          const dfgpu_array* gids, int64_t total) {
  if (!accs || !values || !gids || n_accs < 0) { if (ctx) ctx->err = "acc_update_batch_multi: null argument"; return DFGPU_INVALID_ARGUMENT; }
  // accumulators are independent of each other: visit them ordered by (shareable, value class, filter, value column) so that SUM(x) and
  // AVG(x) become neighbours and share both the pass and the load of x
  std::vector<int32_t> order((size_t)n_accs); for (int32_t k = 0; k < n_accs; k++) order[(size_t)k] = k;
  auto key = [&](int32_t k) { bool ok = multi_ok(accs[k], values[k], total); return std::make_tuple(ok ? 0 : 1, ok ? accs[k]->cls : 0, (const void*)(filters ? filters[k] : nullptr), (const void*)(ok ? values[k]->values->ptr : nullptr)); };
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key(a) < key(b); });
  std::vector<dfgpu_acc*> accs_o; std::vector<const dfgpu_array*> values_o, filters_o;
  for (int32_t k : order) { accs_o.push_back(accs[k]); values_o.push_back(values[k]); filters_o.push_back(filters ? filters[k] : nullptr); }
  accs = accs_o.data(); values = values_o.data(); filters = filters_o.data();
  // COUNT(*) (or COUNT of a NULL-free column) counts exactly the rows an AVG over a NULL-free column with the same filter counts: with
  // many groups every state array costs one memory-side atomic per row, so the COUNT takes the AVG's per-group count delta of this
  // batch instead of its own pass over the rows.
  int donor = -1; std::vector<int> followers; BufferPtr before;
  if (total > SMALL_G && gids->length >= (1 << 20)) {
    auto null_free = [](const dfgpu_array* v) { return v && !v->validity && v->type != DFGPU_DICTIONARY; };
    for (int32_t k = 0; k < n_accs && donor < 0; k++) if (accs[k] && accs[k]->kind == DFGPU_AGG_AVG && null_free(values[k]) && values[k]->length == gids->length) donor = k;
    if (donor >= 0) for (int32_t k = 0; k < n_accs; k++)
      if (accs[k] && accs[k]->kind == DFGPU_AGG_COUNT && filters[k] == filters[donor] && (!values[k] || null_free(values[k]))) followers.push_back(k);
    if (!followers.empty()) {
      dfgpu_status st = guard(ctx, [&] {
        acc_resize(accs[donor], total);
        before = alloc_buffer(ctx, (size_t)total * 8);
        HIP_CHECK(hipMemcpyAsync(before->ptr, accs[donor]->counts->ptr, (size_t)total * 8, hipMemcpyDeviceToDevice, ctx->stream));
      });
      if (st != DFGPU_OK) return st;
    }
  }
  auto is_follower = [&](int32_t k) { return std::find(followers.begin(), followers.end(), (int)k) != followers.end(); };
  struct AddDeltas { dfgpu_ctx* ctx; dfgpu_acc* const* accs; const std::vector<int>& fol; int donor; const BufferPtr& before; int64_t total;
    dfgpu_status run() const { if (fol.empty()) return DFGPU_OK; return guard(ctx, [&] {
      for (int k : fol) { acc_resize(accs[k], total);
        hipLaunchKernelGGL(k_add_count_delta, dim3(grid_for(total, BLOCK)), dim3(BLOCK), 0, ctx->stream, (uint64_t*)accs[k]->counts->ptr, (const uint64_t*)accs[donor]->counts->ptr, (const uint64_t*)before->ptr, total); }
      KERNEL_CHECK(); }); } } add_deltas{ctx, accs, followers, donor, before, total};
  int32_t i = 0;
  while (i < n_accs) {
    if (is_follower(i)) { i++; continue; }
    const dfgpu_array* f = filters ? filters[i] : nullptr;
    int32_t j = i;
    int cap = accs[i] && accs[i]->cls == CLS_I128 ? 2 : MULTI_MAX;
    if (gids->length && multi_ok(accs[i], values[i], total))
      while (j + 1 < n_accs && j + 1 - i < cap && multi_ok(accs[j + 1], values[j + 1], total) && accs[j + 1]->cls == accs[i]->cls && (filters ? filters[j + 1] : nullptr) == f &&
             values[j + 1]->length == gids->length) j++;
    if (j == i) {                 // alone: the single-accumulator path
      dfgpu_status st = dfgpu_acc_update_batch(ctx, accs[i], values[i], gids, f, total);
      if (st != DFGPU_OK) return st;
      i++; continue;
    }
    dfgpu_status st = guard(ctx, [&] {
      int64_t n = gids->length;
      if (gids->type != DFGPU_UINT32) fail(DFGPU_INVALID_ARGUMENT, "group ids must be a UINT32 array");
      materialize_ids(ctx, gids);
      if (f && (f->type != DFGPU_BOOL || f->length != n)) fail(DFGPU_INVALID_ARGUMENT, "opt_filter must be a Boolean array of the batch length");
      MultiArgs ma{}; int na = j - i + 1;
      for (int s = 0; s < na; s++) {
        dfgpu_acc* a = accs[i + s]; const dfgpu_array* v = values[i + s];
        if (v->length != n) fail(DFGPU_INVALID_ARGUMENT, "values (%lld rows) and group ids (%lld rows) differ in length", (long long)v->length, (long long)n);
        acc_resize(a, total);
        ma.vals[s] = v->values->ptr; ma.valid[s] = v->validity ? (const uint64_t*)v->validity->ptr : nullptr;
        ma.out_vals[s] = a->vals->ptr; ma.out_counts[s] = (uint64_t*)a->counts->ptr; ma.out_seen[s] = (uint8_t*)a->seen->ptr; ma.kind[s] = a->kind;
      }
      const uint64_t* fb = f ? (const uint64_t*)f->values->ptr : nullptr; const uint64_t* fv = f && f->validity ? (const uint64_t*)f->validity->ptr : nullptr;
      int blocks = grid_for(n, BLOCK * 8, ctx->num_cus * 8);
      KernelTimer kt_(ctx, "k_acc_update");
      bool plain = fv == nullptr; for (int s2 = 0; s2 < na; s2++) plain = plain && ma.valid[s2] == nullptr;
#define MULTI(T, C, N) do { if (plain) hipLaunchKernelGGL((k_acc_small_multi<T, C, N, true>), dim3(blocks), dim3(BLOCK), 0, ctx->stream, ma, (const uint32_t*)gids->values->ptr, fb, fv, n, (int)total); \
                            else hipLaunchKernelGGL((k_acc_small_multi<T, C, N, false>), dim3(blocks), dim3(BLOCK), 0, ctx->stream, ma, (const uint32_t*)gids->values->ptr, fb, fv, n, (int)total); } while (0)
      int cls = accs[i]->cls;
      if (cls == CLS_I128) MULTI(i128, CLS_I128, 2);
      else if (cls == CLS_F64) { if (na == 2) MULTI(double, CLS_F64, 2); else if (na == 3) MULTI(double, CLS_F64, 3); else MULTI(double, CLS_F64, 4); }
      else { if (na == 2) MULTI(unsigned long long, CLS_U64, 2); else if (na == 3) MULTI(unsigned long long, CLS_U64, 3); else MULTI(unsigned long long, CLS_U64, 4); }
#undef MULTI
      KERNEL_CHECK();
    });
    if (st != DFGPU_OK) return st;
    i = j + 1;
  }
  return add_deltas.run();
}

// ------------------------------------------------------------------ fused "evaluate arguments + accumulate" (run-time compiled)
// TPC-H Q1's shape: a handful of groups, eight accumulators whose arguments are arithmetic over four columns.  Operator at a time that
// is ~64 GB of HBM traffic at SF100 (intermediate columns written and re-read, every accumulator pass re-reading its inputs and the
// group ids); here one kernel reads each input column once, evaluates the argument expressions in registers and keeps per-lane partial
// sums for every (accumulator, group) pair.  The kernel text below is fixed except for the straight-line expression body, and is
// compiled for the expression at hand by hiprtc (jit.hip), so that the partials stay in registers.
extern "C++" {
namespace dfgpu {
void* jit_kernel(dfgpu_ctx* ctx, const std::string& source, const char* name);
bool jit_compile_only(const std::string& source, const char* arch, std::string* log);
void decimal_arith_plan(int op, int p1, int s1, int p2, int s2, int* rp, int* rs, i128* lmul, i128* rmul);      // expr.hip

constexpr int FUSED_MAX = 16;
struct FusedArgs { const void* col[FUSED_MAX]; const uint32_t* gids; const uint64_t* fbits; long long n; void* vals[FUSED_MAX]; uint64_t* counts[FUSED_MAX]; uint8_t* seen[FUSED_MAX]; uint32_t* flags;
                   const void* kcode[2]; const uint32_t* canon[2]; const uint32_t* dense_map; const uint64_t* gmask; };

static const char* FUSED_PRELUDE = R"SRC(
typedef __int128 i128; typedef unsigned __int128 u128; typedef unsigned long long u64; typedef unsigned int u32;
#define I128(hi, lo) ((i128)(((u128)(u64)(hi) << 64) | (u128)(u64)(lo)))
#define GID_NONE 0xFFFFFFFFu
struct Args { const void* col[16]; const u32* gids; const u64* fbits; long long n; void* vals[16]; u64* counts[16]; unsigned char* seen[16]; u32* flags;
              const void* kcode[2]; const u32* canon[2]; const u32* dense_map; const u64* gmask; };
__device__ inline bool add128_checked(i128 a, i128 b, i128* out) { i128 r = (i128)((u128)a + (u128)b); if ((a >= 0) == (b >= 0) && (r >= 0) != (a >= 0)) return false; *out = r; return true; }
__device__ inline bool sub128_checked(i128 a, i128 b, i128* out) { i128 r = (i128)((u128)a - (u128)b); if ((a >= 0) != (b >= 0) && (r >= 0) != (a >= 0)) return false; *out = r; return true; }
__device__ inline bool mul128_checked(i128 a, i128 b, i128* out) {
  const long long al = (long long)a, bl = (long long)b;
  if ((i128)al == a && (i128)bl == b) { *out = (i128)al * (i128)bl; return true; }      // both operands fit 64 bits: the product fits 128 (every TPC-H money value)
  bool neg = (a < 0) != (b < 0);
  u128 ua = a < 0 ? (u128)0 - (u128)a : (u128)a, ub = b < 0 ? (u128)0 - (u128)b : (u128)b;
  u64 a0 = (u64)ua, a1 = (u64)(ua >> 64), b0 = (u64)ub, b1 = (u64)(ub >> 64);
  if (a1 && b1) return false;
  u128 lo = (u128)a0 * (u128)b0;
  u128 cross = a1 ? (u128)a1 * (u128)b0 : (u128)b1 * (u128)a0;
  if (cross >> 64) return false;
  u128 r = lo + (cross << 64);
  if (r < lo) return false;
  if (neg) { if (r > ((u128)1 << 127)) return false; *out = (i128)((u128)0 - r); } else { if (r >> 127) return false; *out = (i128)r; }
  return true;
}
// checked decimal arithmetic of arrow-arith (operands rescaled to the result scale first); a row no accumulator sees cannot raise
__device__ inline i128 dec_arith(int op, i128 x, i128 y, i128 lmul, i128 rmul, bool live, u32* flags) {
  bool ok = true; i128 v = 0;
  if (op != 2) ok = mul128_checked(x, lmul, &x) && mul128_checked(y, rmul, &y);
  if (ok) ok = op == 0 ? add128_checked(x, y, &v) : op == 1 ? sub128_checked(x, y, &v) : mul128_checked(x, y, &v);
  if (!ok) { if (live) atomicOr(flags, 2u); v = 0; }
  return v;
}
template <typename E, int N, int AL> struct __attribute__((aligned(AL))) Vec { E v[N]; };
__device__ inline i128 ld_i128(const void* p, long long i) { const u64* q = (const u64*)p + 2 * i; return I128(q[1], q[0]); }
__device__ inline double ld_f64(const void* p, long long i) { return ((const double*)p)[i]; }
__device__ inline u32 wsum(u32 v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64); return v; }
__device__ inline double wsum(double v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64); return v; }
__device__ inline i128 wsum(i128 v) {
  for (int d = 32; d > 0; d >>= 1) { u64 lo = __shfl_xor((u64)(u128)v, d, 64), hi = __shfl_xor((u64)((u128)v >> 64), d, 64); v = (i128)((u128)v + (u128)I128(hi, lo)); }
  return v;
}
__device__ inline void gadd(void* vals, int k, double v) { atomicAdd((double*)vals + k, v); }
__device__ inline void gadd(void* vals, int k, i128 v) {
  u64* slot = (u64*)vals + 2 * k; u64 lo = (u64)(u128)v, hi = (u64)((u128)v >> 64);
  u64 old = atomicAdd(&slot[0], lo); u64 carry = (u64)(old + lo < old);
  if (hi + carry) atomicAdd(&slot[1], hi + carry);
}
)SRC";

static std::string i128_text(i128 v) { char b[96]; snprintf(b, sizeof b, "I128(0x%016llxull, 0x%016llxull)", (unsigned long long)(uint64_t)((u128)v >> 64), (unsigned long long)(uint64_t)(u128)v); return b; }
static std::string f64_text(double d) { uint64_t u; memcpy(&u, &d, 8); char b[64]; snprintf(b, sizeof b, "__longlong_as_double((long long)0x%016llxull)", (unsigned long long)u); return b; }

struct FusedNodeInfo { int p = 0, s = 0; i128 lmul = 1, rmul = 1; };
// The kernel text for one (expression DAG, accumulator list, group count, mask presence) shape
// dense: the group ids come from dictionary code columns (DeferredIds of dfgpu_groups_intern_deferred) instead of a stored id column:
// nk code columns of type kt, canonical-id tables and strides as in groups.hip's dense_composite_fast
struct FusedDense { int nk = 0; int32_t kt = 0; uint32_t stride[2] = {0, 0}; int64_t len[2] = {1, 1}; bool has_mask = false; };
constexpr int64_t FUSED_DENSE_TABLE = 4096;      // code tuple -> group id table held in LDS
static std::string fused_source(bool dec, int G, int R, bool has_mask, const dfgpu_expr_node* nodes, int n_nodes, const std::vector<FusedNodeInfo>& info,
                                const dfgpu_array* const* cols, dfgpu_acc* const* accs, const int32_t* acc_nodes, int n_accs, std::vector<int>* value_nodes,
                                const FusedDense* dense = nullptr) {
  std::string T = dec ? "i128" : "double", LD = dec ? "ld_i128" : "ld_f64";
  std::vector<int> vn;                                  // distinct argument nodes of SUM / AVG accumulators
  std::vector<int> acc_v((size_t)n_accs, -1);
  for (int i = 0; i < n_accs; i++) if (accs[i]->kind != DFGPU_AGG_COUNT) {
    int j = 0; for (; j < (int)vn.size(); j++) if (vn[(size_t)j] == acc_nodes[i]) break;
    if (j == (int)vn.size()) vn.push_back(acc_nodes[i]);
    acc_v[(size_t)i] = j;
  }
  *value_nodes = vn;
  const int NV = (int)vn.size();
  std::vector<int> used_cols;
  for (int k = 0; k < n_nodes; k++) if (nodes[k].op == DFGPU_NODE_COLUMN && std::find(used_cols.begin(), used_cols.end(), nodes[k].lhs) == used_cols.end()) used_cols.push_back(nodes[k].lhs);
  std::string s = FUSED_PRELUDE;
  char b[512];
  snprintf(b, sizeof b, "typedef %s T;\n#define G %d\n#define R %d\n#define NV %d\n", T.c_str(), G, R, NV > 0 ? NV : 1); s += b;
  s += "extern \"C\" __global__ void __launch_bounds__(256) dfgpu_fused_agg(Args a) {\n"
       "  T acc[NV][G]; u32 cnt[G];\n"
       "#pragma unroll\n  for (int k = 0; k < G; k++) { cnt[k] = 0;\n#pragma unroll\n    for (int v = 0; v < NV; v++) acc[v][k] = (T)0; }\n"
       ;
  if (dense) {      // code tuple -> group id, composed once per workgroup from the canonical-id tables and the dense map (codes are in range: Arrow)
    snprintf(b, sizeof b, "  __shared__ u32 gtab[%lld];\n  for (int t = threadIdx.x; t < %lld; t += 256) gtab[t] = a.dense_map[a.canon[0][t / %lld] * %uu", (long long)(dense->len[0] * dense->len[1]),
             (long long)(dense->len[0] * dense->len[1]), (long long)dense->len[1], dense->stride[0]); s += b;
    if (dense->nk == 2) { snprintf(b, sizeof b, " + a.canon[1][t %% %lld] * %uu", (long long)dense->len[1], dense->stride[1]); s += b; }
    s += "];\n  __syncthreads();\n";
  }
  // Each lane owns R CONSECUTIVE rows per iteration: its R values of a column are one or two 16-byte loads, its R group ids (or code
  // bytes) one load, its R mask bits one word -- a wave still covers a contiguous 64 R-row span.  Whole groups of R rows take the vector
  // path; the ragged last group loads element by element with clamped indices.
  const char* kt = dense ? (dense->kt == DFGPU_INT8 ? "signed char" : dense->kt == DFGPU_INT16 ? "short" : "int") : "int";
  snprintf(b, sizeof b, "  typedef Vec<T, R, 16> TV; typedef Vec<u32, R, 4 * R> GV; typedef Vec<%s, R, sizeof(%s) * R> KV;\n", kt, kt); s += b;
  s += "  const long long stride = (long long)gridDim.x * 256, nvec = (a.n + R - 1) / R;\n"
       "  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride) {\n"
       "    const long long i0 = v * R;\n"
       "    u32 g[R];";
  for (int c : used_cols) { snprintf(b, sizeof b, " T c%d[R];", c); s += b; }
  s += "\n    if (i0 + R <= a.n) {\n";
  if (dense) {
    s += "      const KV k0 = *(const KV*)((const " + std::string(kt) + "*)a.kcode[0] + i0);\n";
    if (dense->nk == 2) s += "      const KV k1 = *(const KV*)((const " + std::string(kt) + "*)a.kcode[1] + i0);\n";
    if (dense->has_mask) s += "      const u64 gm = a.gmask[i0 >> 6] >> (i0 & 63);\n";
  } else s += "      const GV gv = *(const GV*)(a.gids + i0);\n";
  if (has_mask) s += "      const u64 fm = a.fbits[i0 >> 6] >> (i0 & 63);\n";
  for (int c : used_cols) { snprintf(b, sizeof b, "      const TV v%d = *(const TV*)((const T*)a.col[%d] + i0);\n", c, c); s += b; }
  s += "#pragma unroll\n      for (int r = 0; r < R; r++) {\n";
  if (dense) {
    snprintf(b, sizeof b, "        u32 gg = gtab[(u32)k0.v[r] * %lldu%s];\n", (long long)dense->len[1], dense->nk == 2 ? " + (u32)k1.v[r]" : ""); s += b;
    if (dense->has_mask) s += "        if (!((gm >> r) & 1ull)) gg = GID_NONE;\n";
  } else s += "        u32 gg = gv.v[r];\n";
  s += has_mask ? "        g[r] = ((fm >> r) & 1ull) ? gg : GID_NONE;\n" : "        g[r] = gg;\n";
  for (int c : used_cols) { snprintf(b, sizeof b, "        c%d[r] = v%d.v[r];\n", c, c); s += b; }
  s += "      }\n    } else {\n#pragma unroll\n      for (int r = 0; r < R; r++) {\n"
       "        const long long i = i0 + r; const bool in = i < a.n; const long long ii = in ? i : a.n - 1;\n";
  if (dense) {
    snprintf(b, sizeof b, "        u32 gg = gtab[(u32)((const %s*)a.kcode[0])[ii] * %lldu", kt, (long long)dense->len[1]); s += b;
    if (dense->nk == 2) { snprintf(b, sizeof b, " + (u32)((const %s*)a.kcode[1])[ii]", kt); s += b; }
    s += "];\n";
    if (dense->has_mask) s += "        if (!((a.gmask[ii >> 6] >> (ii & 63)) & 1ull)) gg = GID_NONE;\n";
  } else s += "        u32 gg = a.gids[ii];\n";
  s += has_mask ? "        const bool pass = in && ((a.fbits[ii >> 6] >> (ii & 63)) & 1ull);\n" : "        const bool pass = in;\n";
  s += "        g[r] = pass ? gg : GID_NONE;\n";
  for (int c : used_cols) { snprintf(b, sizeof b, "        c%d[r] = %s(a.col[%d], ii);\n", c, LD.c_str(), c); s += b; }
  s += "      }\n";
  s += "    }\n#pragma unroll\n    for (int r = 0; r < R; r++) {\n      const bool live = g[r] != GID_NONE; (void)live;\n";
  for (int k = 0; k < n_nodes; k++) {
    const dfgpu_expr_node& nd = nodes[k];
    if (nd.op == DFGPU_NODE_COLUMN) { snprintf(b, sizeof b, "      const T n%d = c%d[r];\n", k, nd.lhs); s += b; }
    else if (nd.op == DFGPU_NODE_SCALAR) {
      const dfgpu_array* sc = cols[nd.lhs];
      std::string lit; if (dec) { i128 v; memcpy(&v, sc->host_scalar, 16); lit = i128_text(v); } else { double d; memcpy(&d, sc->host_scalar, 8); lit = f64_text(d); }
      snprintf(b, sizeof b, "      const T n%d = %s;\n", k, lit.c_str()); s += b;
    } else if (dec) {
      snprintf(b, sizeof b, "      const T n%d = dec_arith(%d, n%d, n%d, %s, %s, live, a.flags);\n", k, nd.op, nd.lhs, nd.rhs, i128_text(info[(size_t)k].lmul).c_str(), i128_text(info[(size_t)k].rmul).c_str()); s += b;
    } else {
      snprintf(b, sizeof b, "      const T n%d = n%d %c n%d;\n", k, nd.lhs, nd.op == DFGPU_OP_ADD ? '+' : nd.op == DFGPU_OP_SUB ? '-' : '*', nd.rhs); s += b;
    }
  }
  s += "#pragma unroll\n      for (int k = 0; k < G; k++) { const bool m = g[r] == (u32)k; cnt[k] += m ? 1u : 0u;\n";
  for (int v = 0; v < NV; v++) { snprintf(b, sizeof b, "        acc[%d][k] += m ? n%d : (T)0;\n", v, vn[(size_t)v]); s += b; }
  s += "      }\n    }\n  }\n"
       "  __shared__ T s_acc[4][NV][G]; __shared__ u32 s_cnt[4][G];\n"
       "  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;\n"
       "#pragma unroll\n  for (int k = 0; k < G; k++) { u32 c = wsum(cnt[k]); if (lane == 0) s_cnt[wave][k] = c;\n"
       "#pragma unroll\n    for (int v = 0; v < NV; v++) { T x = wsum(acc[v][k]); if (lane == 0) s_acc[wave][v][k] = x; } }\n"
       "  __syncthreads();\n"
       "  if (threadIdx.x >= G) return;\n"
       "  const int k = threadIdx.x; u64 c = 0; for (int w = 0; w < 4; w++) c += s_cnt[w][k];\n"
       "  if (c == 0) return;\n"
       "  T tot[NV];\n#pragma unroll\n  for (int v = 0; v < NV; v++) { tot[v] = (T)0; for (int w = 0; w < 4; w++) tot[v] += s_acc[w][v][k]; }\n";
  for (int i = 0; i < n_accs; i++) {
    if (accs[i]->kind == DFGPU_AGG_COUNT) { snprintf(b, sizeof b, "  atomicAdd(&a.counts[%d][k], c);\n", i); s += b; continue; }
    snprintf(b, sizeof b, "  a.seen[%d][k] = 1; gadd(a.vals[%d], k, tot[%d]);\n", i, i, acc_v[(size_t)i]); s += b;
    if (accs[i]->kind == DFGPU_AGG_AVG) { snprintf(b, sizeof b, "  atomicAdd(&a.counts[%d][k], c);\n", i); s += b; }
  }
  s += "}\n";
  return s;
}

}  // namespace dfgpu
}  // extern "C++"

dfgpu_status dfgpu_acc_update_batch_fused(dfgpu_ctx* ctx, dfgpu_acc* const* accs, const int32_t* acc_nodes, int32_t n_accs, const dfgpu_expr_node* nodes, int32_t n_nodes,
                                          const dfgpu_array* const* cols, int32_t n_cols, const dfgpu_array* gids, const dfgpu_array* filt, int64_t total) {
  return guard(ctx, [&] {
    if (!accs || !acc_nodes || n_accs < 1 || !gids || (n_nodes > 0 && (!nodes || !cols))) fail(DFGPU_INVALID_ARGUMENT, "acc_update_batch_fused: null argument");
    if (n_accs > FUSED_MAX || n_cols > FUSED_MAX || n_nodes > 64) fail(DFGPU_NOT_IMPLEMENTED, "acc_update_batch_fused: more than %d accumulators / columns or 64 nodes", FUSED_MAX);
    if (total < 1 || total > SMALL_G) fail(DFGPU_NOT_IMPLEMENTED, "acc_update_batch_fused: %lld groups (register partials hold up to %d)", (long long)total, SMALL_G);
    if (gids->type != DFGPU_UINT32) fail(DFGPU_INVALID_ARGUMENT, "group ids must be a UINT32 array");
    int64_t n = gids->length;
    if (filt && (filt->type != DFGPU_BOOL || filt->length != n)) fail(DFGPU_INVALID_ARGUMENT, "opt_filter must be a Boolean array of the batch length");
    if (filt && filt->validity) fail(DFGPU_NOT_IMPLEMENTED, "acc_update_batch_fused: nullable filter");
    // one arithmetic class for the whole DAG: Float64, or Decimal128 with arrow-arith's result types per node
    int32_t cls_type = 0; std::vector<FusedNodeInfo> info((size_t)n_nodes);
    for (int k = 0; k < n_nodes; k++) {
      const dfgpu_expr_node& nd = nodes[k];
      if (nd.op == DFGPU_NODE_COLUMN || nd.op == DFGPU_NODE_SCALAR) {
        if (nd.lhs < 0 || nd.lhs >= n_cols || !cols[nd.lhs]) fail(DFGPU_INVALID_ARGUMENT, "acc_update_batch_fused: node %d references column %d of %d", k, nd.lhs, n_cols);
        const dfgpu_array* c = cols[nd.lhs];
        if (c->type != DFGPU_FLOAT64 && c->type != DFGPU_DECIMAL128) fail(DFGPU_NOT_IMPLEMENTED, "acc_update_batch_fused: column type %d", c->type);
        if (cls_type && c->type != cls_type) fail(DFGPU_NOT_IMPLEMENTED, "acc_update_batch_fused: mixed Float64 / Decimal128 arguments");
        cls_type = c->type;
        if (nd.op == DFGPU_NODE_COLUMN) { if (c->length != n) fail(DFGPU_INVALID_ARGUMENT, "column (%lld rows) and group ids (%lld rows) differ in length", (long long)c->length, (long long)n);
                                          if (c->validity) fail(DFGPU_NOT_IMPLEMENTED, "acc_update_batch_fused: nullable column"); }
        else if (c->length != 1 || !c->has_host_scalar || !c->host_scalar_valid) fail(DFGPU_NOT_IMPLEMENTED, "acc_update_batch_fused: scalar operand must be a non-null 1-row literal");
        info[(size_t)k].p = c->precision; info[(size_t)k].s = c->scale;
      } else if (nd.op == DFGPU_OP_ADD || nd.op == DFGPU_OP_SUB || nd.op == DFGPU_OP_MUL) {
        if (nd.lhs < 0 || nd.lhs >= k || nd.rhs < 0 || nd.rhs >= k) fail(DFGPU_INVALID_ARGUMENT, "acc_update_batch_fused: node %d must reference earlier nodes", k);
        if (cls_type == DFGPU_DECIMAL128) {
          FusedNodeInfo& o = info[(size_t)k]; const FusedNodeInfo &l = info[(size_t)nd.lhs], &r = info[(size_t)nd.rhs];
          decimal_arith_plan(nd.op, l.p, l.s, r.p, r.s, &o.p, &o.s, &o.lmul, &o.rmul);
        }
      } else fail(DFGPU_NOT_IMPLEMENTED, "acc_update_batch_fused: operator %d", nd.op);
    }
    const bool dec = cls_type == DFGPU_DECIMAL128;
    for (int i = 0; i < n_accs; i++) {
      dfgpu_acc* a = accs[i]; if (!a) fail(DFGPU_INVALID_ARGUMENT, "acc_update_batch_fused: null accumulator");
      if (a->kind == DFGPU_AGG_COUNT) { if (acc_nodes[i] >= n_nodes) fail(DFGPU_INVALID_ARGUMENT, "acc_update_batch_fused: bad node"); continue; }
      if (a->kind != DFGPU_AGG_SUM && a->kind != DFGPU_AGG_AVG) fail(DFGPU_NOT_IMPLEMENTED, "acc_update_batch_fused: aggregate kind %d", a->kind);
      if (acc_nodes[i] < 0 || acc_nodes[i] >= n_nodes) fail(DFGPU_INVALID_ARGUMENT, "acc_update_batch_fused: accumulator %d has no argument node", i);
      if (a->cls != (dec ? CLS_I128 : CLS_F64)) fail(DFGPU_NOT_IMPLEMENTED, "acc_update_batch_fused: accumulator state class differs from the argument class");
    }
    for (int k = 0; k < n_nodes; k++) if (nodes[k].op == DFGPU_NODE_COLUMN && ((uintptr_t)cols[nodes[k].lhs]->values->ptr & 15)) fail(DFGPU_NOT_IMPLEMENTED, "acc_update_batch_fused: column buffer not 16-byte aligned");
    if (((uintptr_t)gids->values->ptr & 15) || (filt && ((uintptr_t)filt->values->ptr & 7))) fail(DFGPU_NOT_IMPLEMENTED, "acc_update_batch_fused: id / filter buffer alignment");
    for (int i = 0; i < n_accs; i++) acc_resize(accs[i], total);
    if (n == 0) return;
    const int R = dec ? 2 : 4;
    std::vector<int> vn;
    FusedDense fd; std::shared_ptr<DeferredIds> di = gids->deferred_ids;       // keeps the recipe's buffers alive over the launch
    if (di && di->kind != 0) { materialize_ids(ctx, gids); di.reset(); }
    if (di) {       // the code tuple table must fit LDS and the code columns must take R-element vector loads
      bool ok = di->dc.c[0].dict_len * (di->dc.n > 1 ? di->dc.c[1].dict_len : 1) <= FUSED_DENSE_TABLE;
      for (int c = 0; c < di->dc.n; c++) ok = ok && (((uintptr_t)di->dc.c[c].keys) & 15) == 0;
      if (!ok) { materialize_ids(ctx, gids); di.reset(); }
    }
    if (di) { fd.nk = di->dc.n; fd.kt = di->key_type; fd.len[0] = di->dc.c[0].dict_len; fd.len[1] = di->dc.n > 1 ? di->dc.c[1].dict_len : 1; fd.stride[0] = di->dc.c[0].stride; fd.stride[1] = di->dc.n > 1 ? di->dc.c[1].stride : 0; fd.has_mask = di->mask != nullptr; }
    std::string src = fused_source(dec, (int)total, R, filt != nullptr, nodes, n_nodes, info, cols, accs, acc_nodes, n_accs, &vn, di ? &fd : nullptr);
    hipFunction_t fn = (hipFunction_t)jit_kernel(ctx, src, "dfgpu_fused_agg");
    FusedArgs fa{};
    for (int c = 0; c < n_cols; c++) fa.col[c] = cols[c] && cols[c]->values ? cols[c]->values->ptr : nullptr;
    if (di) { for (int c = 0; c < di->dc.n; c++) { fa.kcode[c] = di->dc.c[c].keys; fa.canon[c] = di->dc.c[c].canon; } fa.dense_map = (const uint32_t*)di->dense_map->ptr; fa.gmask = di->mask ? (const uint64_t*)di->mask->ptr : nullptr; }
    fa.gids = (const uint32_t*)gids->values->ptr; fa.fbits = filt ? (const uint64_t*)filt->values->ptr : nullptr; fa.n = n; fa.flags = ctx->d_flags;
    for (int i = 0; i < n_accs; i++) { fa.vals[i] = accs[i]->vals ? accs[i]->vals->ptr : nullptr; fa.counts[i] = (uint64_t*)accs[i]->counts->ptr; fa.seen[i] = (uint8_t*)accs[i]->seen->ptr; }
    void* params[] = { &fa };
    int blocks = grid_for(n, BLOCK * R * 4, ctx->num_cus * 8);
    { KernelTimer kt_(ctx, "k_acc_fused");
      HIP_CHECK(hipModuleLaunchKernel(fn, (unsigned)blocks, 1, 1, BLOCK, 1, 1, 0, ctx->stream, params, nullptr)); }
    check_flags(ctx, "acc_update_batch_fused");
  });
}

/* Build check without a device: compile the generator's output for one representative shape (Decimal128 and Float64). */
dfgpu_status dfgpu_jit_selftest(const char* arch, char* log, int64_t log_cap) {
  using namespace dfgpu;
  try {
    for (int dec = 0; dec < 2; dec++) {
      dfgpu_array one{}; one.type = dec ? DFGPU_DECIMAL128 : DFGPU_FLOAT64; one.length = 1; one.has_host_scalar = true; one.host_scalar_valid = true; one.precision = 20;
      if (dec) { i128 v = 1; memcpy(one.host_scalar, &v, 16); } else { double d = 1.0; memcpy(one.host_scalar, &d, 8); }
      dfgpu_array col{}; col.type = one.type; col.precision = 15; col.scale = 2;
      const dfgpu_array* cols[3] = { &col, &col, &one };
      dfgpu_expr_node nodes[5] = { {DFGPU_NODE_COLUMN, 0, 0}, {DFGPU_NODE_COLUMN, 1, 0}, {DFGPU_NODE_SCALAR, 2, 0}, {DFGPU_OP_SUB, 2, 1}, {DFGPU_OP_MUL, 0, 3} };
      std::vector<FusedNodeInfo> info(5);
      if (dec) { info[0] = info[1] = FusedNodeInfo{15, 2, 1, 1}; info[2] = FusedNodeInfo{20, 0, 1, 1};
        decimal_arith_plan(DFGPU_OP_SUB, 20, 0, 15, 2, &info[3].p, &info[3].s, &info[3].lmul, &info[3].rmul);
        decimal_arith_plan(DFGPU_OP_MUL, 15, 2, info[3].p, info[3].s, &info[4].p, &info[4].s, &info[4].lmul, &info[4].rmul); }
      dfgpu_acc a0{}, a1{}, a2{}; a0.kind = DFGPU_AGG_SUM; a1.kind = DFGPU_AGG_AVG; a2.kind = DFGPU_AGG_COUNT;
      dfgpu_acc* accs[3] = { &a0, &a1, &a2 }; int32_t acc_nodes[3] = { 4, 0, -1 }; std::vector<int> vn;
      FusedDense fd; fd.nk = 2; fd.kt = DFGPU_INT8; fd.stride[0] = 3; fd.stride[1] = 1; fd.len[0] = 3; fd.len[1] = 2; fd.has_mask = true;
      for (int dense = 0; dense < 2; dense++) {       // group ids from a stored column, and from two dictionary code columns
        std::string src = fused_source(dec != 0, 6, dec ? 2 : 4, true, nodes, 5, info, cols, accs, acc_nodes, 3, &vn, dense ? &fd : nullptr);
        std::string l;
        if (!jit_compile_only(src, arch ? arch : "gfx950", &l)) { if (log && log_cap > 0) snprintf(log, (size_t)log_cap, "%s", l.c_str()); return DFGPU_INTERNAL; }
      }
    }
  } catch (...) { if (log && log_cap > 0) snprintf(log, (size_t)log_cap, "kernel text generation failed"); return DFGPU_INTERNAL; }
  return DFGPU_OK;
}

// merge_batch of states that carry group i in row i (ids 0 .. total-1, flagged as the identity) into an accumulator that holds nothing yet -- the partial rows of a first,
// fully pre-aggregated batch (AggregateExec::merge_partial): the states ARE the accumulator's new contents, so they are copied in instead of being added row by row
static bool acc_adopt_identity(dfgpu_ctx* ctx, dfgpu_acc* a, const dfgpu_array* const* st, int32_t nst, const dfgpu_array* gids, const dfgpu_array* filt, int64_t total) {
  if (a->n != 0 || filt || !gids || !gids->identity || gids->length != total || total <= 0) return false;
  if (a->kind == DFGPU_AGG_AVG) return false;            // its `seen` is count > 0 per group: left to the row-by-row merge
  const int want = 1; if (nst != want) return false;
  for (int i = 0; i < nst; i++) if (!st[i] || st[i]->validity || st[i]->length != total || st[i]->type == DFGPU_DICTIONARY || !st[i]->values) return false;
  const dfgpu_array* vals = a->kind == DFGPU_AGG_COUNT ? nullptr : st[nst - 1];
  if (a->kind == DFGPU_AGG_COUNT && st[0]->type != DFGPU_INT64) return false;
  if (a->kind == DFGPU_AGG_AVG && st[0]->type != DFGPU_UINT64) return false;
  if (vals && (vals->type != a->state_type || type_width(vals->type) != a->width)) return false;
  // nothing is copied here: 20 M groups x (SUM, COUNT) were 0.35 ms of copies and fills per step, written only to be copied out again by the evaluation that follows a
  // single-batch aggregation (profiles/r04_n_timeline_gb20.txt)
  a->lazy = true; a->lazy_vals = vals ? vals->values : BufferPtr(); a->lazy_counts = a->kind == DFGPU_AGG_COUNT ? st[0]->values : BufferPtr();
  a->vals.reset(); a->counts.reset(); a->seen.reset(); a->cap = 0;
  a->n = total;
  return true;
}
dfgpu_status dfgpu_acc_merge_batch(dfgpu_ctx* ctx, dfgpu_acc* a, const dfgpu_array* const* st, int32_t nst, const dfgpu_array* gids, const dfgpu_array* filt, int64_t total) {
  if (!a || !st) return DFGPU_INVALID_ARGUMENT;
  { bool adopted = false; dfgpu_status rc = guard(ctx, [&] { adopted = acc_adopt_identity(ctx, a, st, nst, gids, filt, total); }); if (rc != DFGPU_OK || adopted) return rc; }
  if (a->kind == DFGPU_AGG_COUNT) {         // count.rs:135-170: add the partial counts (never null)
    return guard(ctx, [&] {
      if (nst != 1 || st[0]->type != DFGPU_INT64) fail(DFGPU_INVALID_ARGUMENT, "COUNT merge expects one Int64 state");
      acc_resize(a, total);
      launch_update(a, DFGPU_AGG_SUM, CLS_I64, st[0], gids, filt, total, a->counts->ptr);     // sum into counts; `seen` is unused by COUNT
      check_flags(ctx, "acc_merge_batch");
    });
  }
  if (a->kind == DFGPU_AGG_AVG) {           // average.rs:472-509
    return guard(ctx, [&] {
      if (nst != 2 || st[0]->type != DFGPU_UINT64) fail(DFGPU_INVALID_ARGUMENT, "AVG merge expects (UInt64 counts, sums)");
      acc_resize(a, total);
      // the counts are summed without touching the null state (accumulate_indices, average.rs:489-497); only a non-NULL partial sum marks its group as seen (:499-507) -- a
      // partial state of count 0 and a NULL sum (a group whose argument was NULL in every row) must leave the group NULL
      { BufferPtr seen = a->seen; a->seen = alloc_buffer(ctx, (size_t)a->cap + 64);
        struct Restore { dfgpu_acc* a; BufferPtr s; ~Restore() { a->seen = s; } } restore{a, seen};
        launch_update(a, DFGPU_AGG_SUM, CLS_U64, st[0], gids, filt, total, a->counts->ptr); }
      launch_update(a, DFGPU_AGG_SUM, a->cls, st[1], gids, filt, total, a->vals->ptr);
      check_flags(ctx, "acc_merge_batch");
    });
  }
  if (nst != 1) { if (ctx) ctx->err = "merge expects one state column"; return DFGPU_INVALID_ARGUMENT; }
  return dfgpu_acc_update_batch(ctx, a, st[0], gids, filt, total);     // prim_op.rs:119-127: update / merge are the same
}

static dfgpu_array* emit_values(dfgpu_ctx* ctx, dfgpu_acc* a, int32_t type, int32_t p, int32_t s, const void* src, bool with_seen) {
  if (a->lazy) {            // adopted states, every group seen: the adopted buffer is the result (or its narrowed copy)
    const BufferPtr& buf = a->kind == DFGPU_AGG_COUNT ? a->lazy_counts : a->lazy_vals;
    if (type_width(type) == a->width || a->cls == CLS_I128 || a->kind == DFGPU_AGG_COUNT) { ArrayHolder h(new_array(ctx, type, a->n, p, s)); h.get()->values = buf; h.get()->null_count = 0; return h.release(); }
    ArrayHolder h(new_fixed(ctx, type, a->n, p, s, false));
    hipLaunchKernelGGL(k_narrow, dim3(grid_for(a->n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)buf->ptr, a->n, type, h.get()->values->ptr); KERNEL_CHECK();
    return h.release();
  }
  ArrayHolder h(new_fixed(ctx, type, a->n, p, s, with_seen));
  if (a->n) {
    if (type_width(type) == a->width || a->cls == CLS_I128) HIP_CHECK(hipMemcpyAsync(h.get()->values->ptr, src, (size_t)a->n * type_width(type), hipMemcpyDeviceToDevice, ctx->stream));
    else hipLaunchKernelGGL(k_narrow, dim3(grid_for(a->n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)src, a->n, type, h.get()->values->ptr);
    if (with_seen) hipLaunchKernelGGL(k_seen_to_bits, dim3(grid_for(a->n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint8_t*)a->seen->ptr, a->n, (uint64_t*)h.get()->validity->ptr);
    KERNEL_CHECK();
  }
  if (with_seen) h.get()->null_count = -1;
  return h.release();
}

dfgpu_status dfgpu_acc_evaluate(dfgpu_ctx* ctx, dfgpu_acc* a, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!a || !out) fail(DFGPU_INVALID_ARGUMENT, "acc_evaluate: null argument");
    if (a->kind == DFGPU_AGG_COUNT) { *out = emit_values(ctx, a, DFGPU_INT64, 0, 0, a->counts ? a->counts->ptr : nullptr, false); return; }
    if (a->kind != DFGPU_AGG_AVG) { *out = emit_values(ctx, a, a->out_type, a->out_precision, a->out_scale, a->vals ? a->vals->ptr : nullptr, true); return; }
    ArrayHolder h(new_fixed(ctx, a->out_type, a->n, a->out_precision, a->out_scale, true));
    if (a->n) {
      dim3 grid(grid_for(a->n, BLOCK)), block(BLOCK);
      if (a->out_type == DFGPU_FLOAT64) hipLaunchKernelGGL(k_avg_f64, grid, block, 0, ctx->stream, (const double*)a->vals->ptr, (const uint64_t*)a->counts->ptr, (const uint8_t*)a->seen->ptr, a->n, (double*)h.get()->values->ptr);
      else { i128 factor = pow10_i128(a->out_scale - a->state_scale);
             hipLaunchKernelGGL(k_avg_dec, grid, block, 0, ctx->stream, (const uint64_t*)a->vals->ptr, (const uint64_t*)a->counts->ptr, (const uint8_t*)a->seen->ptr, a->n, factor, a->out_precision, (uint64_t*)h.get()->values->ptr, ctx->d_flags); }
      hipLaunchKernelGGL(k_seen_to_bits, grid, block, 0, ctx->stream, (const uint8_t*)a->seen->ptr, a->n, (uint64_t*)h.get()->validity->ptr);
      KERNEL_CHECK();
      uint32_t f = 0; fetch_to_pinned(ctx, 63, ctx->d_flags, 4); f = *(uint32_t*)(ctx->h_pinned + 63);
      if (f) { HIP_CHECK(hipMemsetAsync(ctx->d_flags, 0, 4, ctx->stream)); fail(DFGPU_EXECUTION, "Arithmetic Overflow in AvgAccumulator"); }
    }
    h.get()->null_count = -1;
    *out = h.release();
  });
}

dfgpu_status dfgpu_acc_state(dfgpu_ctx* ctx, dfgpu_acc* a, dfgpu_array** out_states, int32_t* n_states) {
  if (!a || !out_states || !n_states) return DFGPU_INVALID_ARGUMENT;
  if (a->kind != DFGPU_AGG_AVG) { *n_states = 1; return dfgpu_acc_evaluate(ctx, a, &out_states[0]); }
  return guard(ctx, [&] {
    ArrayHolder c(emit_values(ctx, a, DFGPU_UINT64, 0, 0, a->counts ? a->counts->ptr : nullptr, true));
    ArrayHolder s(emit_values(ctx, a, a->state_type, a->state_precision, a->state_scale, a->vals ? a->vals->ptr : nullptr, true));
    out_states[0] = c.release(); out_states[1] = s.release(); *n_states = 2;
  });
}

/* see include/dfgpu.h */
dfgpu_status dfgpu_acc_emit_first(dfgpu_ctx* ctx, dfgpu_acc* a, int64_t n, int32_t as_state, dfgpu_array** out, int32_t* n_out) {
  return guard(ctx, [&] {
    if (!a || !out || n < 0) fail(DFGPU_INVALID_ARGUMENT, "acc_emit_first: bad argument");
    const int64_t total = a->n, k = n < total ? n : total;
    dfgpu_array* st[2] = {nullptr, nullptr}; int32_t ns = 0;
    dfgpu_status rc = dfgpu_acc_state(ctx, a, st, &ns); if (rc != DFGPU_OK) fail(rc, "%s", ctx->err.c_str());
    ArrayHolder s0(st[0]), s1(st[1]);
    ArrayHolder ev; if (!as_state) { dfgpu_array* e = nullptr; rc = dfgpu_acc_evaluate(ctx, a, &e); if (rc != DFGPU_OK) fail(rc, "%s", ctx->err.c_str()); ev.a = e; }
    auto slice = [&](const dfgpu_array* x, int64_t off, int64_t len) { dfgpu_array* o = nullptr; dfgpu_status r2 = dfgpu_array_slice(ctx, x, off, len, &o); if (r2 != DFGPU_OK) fail(r2, "%s", ctx->err.c_str()); return o; };
    ArrayHolder o0, o1;
    if (as_state) { o0.a = slice(s0.get(), 0, k); if (ns > 1) o1.a = slice(s1.get(), 0, k); } else o0.a = slice(ev.get(), 0, k);
    // the groups that stay are renumbered from 0 (EmitTo::take_needed): their states are merged into the emptied accumulator under ids 0 .. total - k - 1
    const int kind = a->kind; const int32_t it = a->in_type, ip = a->in_precision, is = a->in_scale;
    a->n = 0; a->cap = 0; a->vals.reset(); a->counts.reset(); a->seen.reset(); a->lazy = false; a->lazy_vals.reset(); a->lazy_counts.reset();
    (void)kind; (void)it; (void)ip; (void)is;
    if (k < total) {
      ArrayHolder r0(slice(s0.get(), k, total - k)), r1; if (ns > 1) r1.a = slice(s1.get(), k, total - k);
      dfgpu_array* ids = nullptr; rc = dfgpu_array_iota(ctx, total - k, &ids); if (rc != DFGPU_OK) fail(rc, "%s", ctx->err.c_str()); ArrayHolder idh(ids);
      const dfgpu_array* sp[2] = { r0.get(), r1.get() };
      rc = dfgpu_acc_merge_batch(ctx, a, sp, ns, idh.get(), nullptr, total - k); if (rc != DFGPU_OK) fail(rc, "%s", ctx->err.c_str());
    }
    out[0] = o0.release(); if (as_state && ns > 1) out[1] = o1.release();
    if (n_out) *n_out = as_state ? ns : 1;
  });
}

}  // extern "C"
