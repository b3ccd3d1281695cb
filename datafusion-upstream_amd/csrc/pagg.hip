// pagg.hip -- partitioned pre-aggregation for high-cardinality, unclustered group keys: every partition's groups live in LDS.
//
// Reference semantics: the Partial stage of a two-phase aggregation (AggregateMode::Partial, physical-plan/src/aggregates/mod.rs:64-100;
// GroupedHashAggregateStream::group_aggregate_batch, row_hash.rs:524-613) applied to ONE batch inside the operator: the batch is reduced
// to (group key, partial state...) rows, which the caller interns (GroupValues::intern, group_values/primitive.rs:112-149) and merges
// (GroupsAccumulator::merge_batch, prim_op.rs:119-127; average.rs:472-509; count.rs:135-170) exactly as a Final stage would.
//
// Why: with more groups than an LDS cache holds, dfgpu_acc_update_batch pays one memory-side atomic per row and state array
// (~24 G/s: 8.5 ms per 100 M rows for SUM + COUNT) and groups.hip one random 64-byte sector per row for the key table.  Here the rows
// are split by key hash (radix_partition.h, one pass) into P partitions so that a partition's distinct keys fit an LDS table
// (4096 slots of key + first row + row count + one 8-byte state per aggregate); one workgroup per partition accumulates its rows with
// LDS atomics and writes one row per group.  If a partition holds more keys than the table (the cardinality estimate was low, or
// the keys are skewed towards it) the workgroup flushes the table and goes on: a key may then leave in several partial rows, which
// the merge downstream adds up -- never wrong, only less reduction.
//
// Order: partial rows come back sorted by the first input row of their group, so interning them numbers the groups in first-seen order
// of the ORIGINAL batch (group_values/primitive.rs:137-141).  Integer / count states are exact; Float64 sums are added in a different
// order than row order (as every multi-partition plan of the reference does): within 1e-9 relative of the sequential sum.
#include "device_utils.h"
#include "radix_partition.h"

namespace dfgpu {

constexpr uint64_t PA_EMPTY = ~0ull;
constexpr int PA_NT = 1024;
constexpr int PA_MAX_AGGS = 6;
enum { PA_SUM_I64 = 0, PA_SUM_F64 = 1, PA_MIN_I64 = 2, PA_MAX_I64 = 3, PA_MIN_U64 = 4, PA_MAX_U64 = 5, PA_MIN_F64 = 6, PA_MAX_F64 = 7, PA_NONE = 8,
       PA_SUM_I128_LO = 9, PA_SUM_I128_HI = 10, PA_SUM_I128_SX = 11,
       PA_COUNT_FLAG = 12 };      // COUNT_FLAG: the cell counts the rows whose flag bit is set (the non-NULL values of a nullable argument): a wrapping Int64 sum of the bit      // SX: the high word is the sign extension of the low one (Decimal128 of precision <= 18 moves 8 bytes per row through the partition)      // Decimal128 SUM / AVG: two neighbouring cells, the low add's returned value gives the carry (exact mod 2^128, as acc.hip does in HBM)

// Nullable arguments: the validity bits of the (at most 8) nullable value columns travel through the partition as ONE byte per row (vflag; bit b = value column b is valid in
// that row).  flag_bit[a] >= 0: cell a takes its row's value only where that bit is set (a NULL adds the operation's identity, i.e. nothing); a COUNT_FLAG cell sums the bit.
struct PaPlan { int32_t n_acc; int32_t op[PA_MAX_AGGS]; const uint64_t* val[PA_MAX_AGGS]; int32_t vstride[PA_MAX_AGGS]; uint64_t* out[PA_MAX_AGGS]; int32_t has_i128; const uint8_t* vflag; int32_t flag_bit[PA_MAX_AGGS]; };      // val[a][i * vstride[a]]: a Decimal128 column is two cells of stride 2

__device__ inline uint64_t pa_identity(int op) {
  switch (op) {
    case PA_MIN_I64: return (uint64_t)INT64_MAX; case PA_MAX_I64: return (uint64_t)INT64_MIN; case PA_MIN_U64: return ~0ull; case PA_MAX_U64: return 0ull;
    case PA_MIN_F64: return 0x7FEFFFFFFFFFFFFFull /* f64::MAX */; case PA_MAX_F64: return 0xFFEFFFFFFFFFFFFFull /* f64::MIN */;      // the reference's starting values (min_max.rs:102-139): `cur < new` never lets +-inf beyond them or a NaN in
    default: return 0ull;
  }
}
__device__ inline void pa_apply(int op, unsigned long long* cell, uint64_t v) {
  switch (op) {
    case PA_SUM_I64: case PA_COUNT_FLAG: atomicAdd(cell, (unsigned long long)v); break;                                   // wrapping, sum.rs:137
    case PA_SUM_F64: atomicAdd((double*)cell, __longlong_as_double((long long)v)); break;
    // MIN / MAX: a cell only ever moves one way, so a value that does not improve on a plain (possibly stale) read of it cannot improve on the cell: no atomic.  After a
    // group's first few rows that is nearly every row (the expected number of new maxima among n values is ln n), and on a hot slot it takes the read-modify-write queue away
    case PA_MIN_I64: if ((long long)v < *(volatile long long*)cell) atomicMin((long long*)cell, (long long)v); break;
    case PA_MAX_I64: if ((long long)v > *(volatile long long*)cell) atomicMax((long long*)cell, (long long)v); break;
    case PA_MIN_U64: if (v < *(volatile unsigned long long*)cell) atomicMin(cell, (unsigned long long)v); break;
    case PA_MAX_U64: if (v > *(volatile unsigned long long*)cell) atomicMax(cell, (unsigned long long)v); break;
    case PA_MIN_F64: { double d = __longlong_as_double((long long)v); if (d == d && d < *(volatile double*)cell) atomicMin((double*)cell, d); break; }     // `if *cur > new` (min_max.rs:124-139) is false for a NaN: NaN inputs never enter, whatever the order
    case PA_MAX_F64: { double d = __longlong_as_double((long long)v); if (d == d && d > *(volatile double*)cell) atomicMax((double*)cell, d); break; }
    default: break;
  }
}
__device__ inline uint64_t pa_combine(int op, uint64_t a, uint64_t b) {          // what pa_apply's atomic does, on two values
  switch (op) {
    case PA_SUM_I64: case PA_COUNT_FLAG: return a + b;
    case PA_SUM_F64: return (uint64_t)__double_as_longlong(__longlong_as_double((long long)a) + __longlong_as_double((long long)b));
    case PA_MIN_I64: return (long long)a < (long long)b ? a : b; case PA_MAX_I64: return (long long)a > (long long)b ? a : b;
    case PA_MIN_U64: return a < b ? a : b; case PA_MAX_U64: return a > b ? a : b;
    case PA_MIN_F64: { const double x = __longlong_as_double((long long)a), y = __longlong_as_double((long long)b); return y == y && (x != x || y < x) ? b : a; }      // a NaN never wins
    case PA_MAX_F64: { const double x = __longlong_as_double((long long)a), y = __longlong_as_double((long long)b); return y == y && (x != x || y > x) ? b : a; }
    default: return a;
  }
}
__device__ inline uint32_t pa_slot(uint64_t k, int cbits) {
  uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
  uint32_t a = lo ^ (hi * 0x9E3779B1u), x = a * 0x85EBCA6Bu; x ^= x >> 13;
  return (x * 0xC2B2AE35u) >> (32 - cbits);
}

__global__ void k_pa_max_len(const uint32_t* pstart, uint32_t P, unsigned long long* out) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; uint32_t len = p < P ? pstart[p + 1] - pstart[p] : 0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { uint32_t o = __shfl_xor(len, d, 64); len = o > len ? o : len; }
  if (lane_id() == 0 && len) atomicMax(out, (unsigned long long)len);
}
// one workgroup per partition (and slice).  LDS: keys u64[C + 1] | acc[n_acc] u64[C + 1] | first u32[C + 1] | cnt u32[C + 1]; slot C belongs to the key
// that equals the EMPTY marker.  Rows are taken PA_NT at a time with a barrier in between; before a chunk the table is flushed if the
// chunk could fill it beyond 7/8.
// (partition, slice) work items of the skewed case: item_start[p] = first item of partition p, item_start[P] = number of items.  A 2-D grid over
// (partition, max slices) would be mostly empty workgroups, and an empty workgroup still queues for a CU's LDS behind the ones doing work: the slices of the one
// long partition then start milliseconds late.
__global__ void __launch_bounds__(PA_NT) k_pa_items(const uint32_t* pstart, uint32_t P, uint32_t slice, uint32_t* item_start) {
  __shared__ uint32_t wsum[PA_NT / WAVE]; __shared__ uint32_t carry_sh;
  if (threadIdx.x == 0) carry_sh = 0;
  __syncthreads();
  for (uint32_t p0 = 0; p0 < P; p0 += PA_NT) {
    const uint32_t p = p0 + threadIdx.x; uint32_t c = 0;
    if (p < P) { const uint32_t len = pstart[p + 1] - pstart[p]; c = (uint32_t)(((uint64_t)len + slice - 1) / slice); }
    const uint32_t inc = wave_inclusive_sum(c);
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t run = carry_sh + inc - c, tot = 0;
    for (int w = 0; w < PA_NT / WAVE; w++) { if (w < (int)(threadIdx.x >> 6)) run += wsum[w]; tot += wsum[w]; }
    if (p < P) item_start[p] = run;
    __syncthreads();
    if (threadIdx.x == 0) carry_sh += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) item_start[P] = carry_sh;
}
// two-level partition: fine partition = (low hash bits -> P2) * P1 + (high hash bits -> P1)
__device__ inline uint32_t pa_fine_pid(uint64_t k, uint32_t P1, uint32_t P2) { uint64_t h = mix64(k); return (uint32_t)(((h & 0xFFFFFFFFull) * (uint64_t)P2) >> 32) * P1 + rp_pid(h, P1); }
template <bool I128, bool FLAGS>          // I128: the plan holds Decimal128 cell pairs / strided value columns; FLAGS: some argument is nullable (flag byte per row).  The plain instantiation is the round-2 kernel (the I128 branches cost it 6 %, the flag tests 15 %)
__global__ void __launch_bounds__(PA_NT) k_pa_aggregate(const uint64_t* pkey, const uint32_t* prow, PaPlan plan_arg, const uint32_t* pstart, const uint32_t* item_start, uint32_t P, int cbits, uint32_t slice,
                                                      uint64_t* orec /* records of `rs` words: key, count, accumulator cells */, int rs, uint32_t* ofirst, unsigned long long* cursor /*[0] rows written, [2] tables flushed before their partition ended*/,
                                                      uint32_t P1, uint32_t P2, uint32_t* misplaced /* two-level partition (P1 != 0): set when a row sits in a partition its key does not hash to */) {
  extern __shared__ unsigned long long pa_lds[];
  const uint32_t C = 1u << cbits, M = C - 1, C1 = C + 1;
  unsigned long long* keys = pa_lds; unsigned long long* acc = pa_lds + C1;
  uint32_t* first = (uint32_t*)(pa_lds + (size_t)C1 * (1 + plan_arg.n_acc)); uint32_t* cnt = first + C1;
  __shared__ uint32_t nfilled, wsum[PA_NT / WAVE], out_base; __shared__ PaPlan splan;
  if (threadIdx.x == 0) splan = plan_arg;              // the per-aggregate loops index the plan at run time: from LDS, not from the by-value argument
  __syncthreads();
  const PaPlan& plan = splan;
  // a partition far above the average size (skewed keys) is cut into slices of `slice` rows, each with a table of its own: workgroup = one item of item_start
  uint32_t p = blockIdx.x, sl = 0; const int lane = lane_id();
  if (item_start) {
    if (blockIdx.x >= item_start[P]) return;
    uint32_t lo = 0, hi = P - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (item_start[mid] <= blockIdx.x) lo = mid; else hi = mid - 1; }
    p = lo; sl = blockIdx.x - item_start[p];
  }
  const uint32_t p0 = pstart[p], p1 = pstart[p + 1];
  if ((uint64_t)sl * slice >= (uint64_t)(p1 - p0)) return;
  const uint32_t q0 = p0 + sl * slice, q1 = (uint64_t)q0 + slice < (uint64_t)p1 ? q0 + slice : p1;
  auto reset = [&]() {
    for (uint32_t s = threadIdx.x; s < C1; s += PA_NT) { keys[s] = PA_EMPTY; first[s] = 0xFFFFFFFFu; cnt[s] = 0;
      for (int a = 0; a < plan.n_acc; a++) acc[(size_t)a * C1 + s] = pa_identity(plan.op[a]); }
    if (threadIdx.x == 0) nfilled = 0;
  };
  auto flush = [&]() {      // occupied slots -> one output row each; slot ranks by a block scan over PER consecutive slots per thread
    const uint32_t per = (C1 + PA_NT - 1) / PA_NT, s0 = threadIdx.x * per; uint32_t c = 0;
    for (uint32_t j = 0; j < per; j++) { uint32_t s = s0 + j; if (s < C1 && cnt[s]) c++; }
    uint32_t inc = wave_inclusive_sum(c);
    if (lane == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t run = inc - c, tot = 0;
    for (int w = 0; w < PA_NT / WAVE; w++) { if (w < (int)(threadIdx.x >> 6)) run += wsum[w]; tot += wsum[w]; }
    if (threadIdx.x == 0) out_base = tot ? (uint32_t)atomicAdd(cursor, (unsigned long long)tot) : 0u;
    __syncthreads();
    uint32_t o = out_base + run;
    for (uint32_t j = 0; j < per; j++) { uint32_t s = s0 + j; if (s < C1 && cnt[s]) {
      uint64_t* rec = orec + (size_t)o * (size_t)rs; rec[0] = s == C ? PA_EMPTY : keys[s]; rec[1] = cnt[s]; ofirst[o] = first[s];
      for (int a = 0; a < plan.n_acc; a++) rec[2 + a] = acc[(size_t)a * C1 + s];
      o++; } }
    __syncthreads();
  };
  reset();
  __syncthreads();
  // the next chunk's row (key, row number, value cells) is loaded while the current one goes through the table
  uint64_t kn = 0, vn[PA_MAX_AGGS]; uint32_t rn = 0;
  const int na = plan_arg.n_acc;
  const uint8_t* vflag = plan_arg.vflag;
  // a row's cells as the table takes them: the value, or -- for a cell bound to a flag bit -- the operation's identity where the bit is clear (a NULL argument), the bit itself for a COUNT_FLAG cell
  auto load_cells = [&](uint32_t ic, uint64_t* v) {
#pragma unroll
    for (int a = 0; a < PA_MAX_AGGS; a++) v[a] = a < na && (!FLAGS || plan_arg.val[a]) ? plan_arg.val[a][I128 ? (size_t)ic * plan_arg.vstride[a] : (size_t)ic] : 0;
    if constexpr (FLAGS) {
      const uint32_t f = vflag[ic];
#pragma unroll
      for (int a = 0; a < PA_MAX_AGGS; a++) if (a < na && plan_arg.flag_bit[a] >= 0) {
        const bool set = (f >> plan_arg.flag_bit[a]) & 1u; const int op = plan_arg.op[a];
        v[a] = op == PA_COUNT_FLAG ? (uint64_t)set : set ? v[a] : (op == PA_SUM_I128_LO || op == PA_SUM_I128_HI || op == PA_SUM_I128_SX) ? 0ull : pa_identity(op);
      }
    }
  };
  { const uint32_t i = q0 + threadIdx.x, ic = i < q1 ? i : q1 - 1; kn = pkey[ic]; rn = prow ? prow[ic] : 0u; load_cells(ic, vn); }
  for (uint32_t i0 = q0; i0 < q1; i0 += PA_NT) {
    if (nfilled + PA_NT > C - C / 8) { __syncthreads(); flush(); reset(); if (threadIdx.x == 0) atomicAdd(cursor + 2, 1ull); __syncthreads(); }      // nfilled is only written between barriers: uniform
    const uint32_t i = i0 + threadIdx.x; const bool on = i < q1;
    const uint64_t k = kn; const uint32_t row = rn; uint64_t v[PA_MAX_AGGS];
    if (P1 && on && (i & 3u) == 0 && pa_fine_pid(k, P1, P2) != p) *misplaced = 1u;     // the partition bounds came from a binary search that relies on the order the two passes leave: every 4th row re-hashed as an assertion
#pragma unroll
    for (int a = 0; a < PA_MAX_AGGS; a++) v[a] = vn[a];
    { const uint32_t i2 = i + PA_NT, ic = i2 < q1 ? i2 : q1 - 1; kn = pkey[ic]; rn = prow ? prow[ic] : 0u; load_cells(ic, vn); }
    // Skewed keys: when at least 16 lanes of a wave carry the key of its first active lane, those lanes are combined in registers (shuffles) and the leader
    // alone touches the table: one LDS atomic per state instead of one per row on a slot every wave of the workgroup is hammering.  (Trying the last combined key
    // first -- a key with a fifth to a half of its partition's rows sits in the first lane only that often -- was measured on the Zipf ClickBench shape and changed
    // nothing: 2.34 -> 2.49 ms, round 4 call r; the partitions there hold several warm keys each, not one.  Keeping the hot key's rows in accumulators of the lanes' own
    // and combining them once, when another key takes over or the slice ends, was tried as well (call aa): Zipf 2.36 -> 2.16 ms, but every other shape lost 10-30 % to the
    // longer loop body -- uniform ClickBench 1.07 -> 1.27 ms, 20 M groups 1.64 -> 1.75, three keys + Decimal128 1.04 -> 1.35 -- and it was taken out again.  Nor did trying up to three candidate keys per round with DPP reductions in place of the shuffles (call ai): Zipf 2.36 -> 3.04 ms at the same threshold
    // of 16 lanes, 3.5 / 3.8 ms at 8 / 4; uniform 1.07 -> 1.2 ms.)
    uint32_t cntv = 1; uint32_t rowv = row; bool mine = on;
    {
      const uint64_t act = ballot64(on);
      const int lead = act ? __ffsll((long long)act) - 1 : 0;
      const uint32_t k0lo = (uint32_t)__shfl((int)(uint32_t)k, lead, 64), k0hi = (uint32_t)__shfl((int)(uint32_t)(k >> 32), lead, 64);
      const bool member = on && (uint32_t)k == k0lo && (uint32_t)(k >> 32) == k0hi;
      const uint64_t mem = ballot64(member);
      if (__popcll(mem) >= 16 && !I128) {
        uint32_t r = member ? row : 0xFFFFFFFFu;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { uint32_t o = (uint32_t)__shfl_xor((int)r, d, 64); r = o < r ? o : r; }
#pragma unroll
        for (int a = 0; a < PA_MAX_AGGS; a++) if (a < na) {
          const int op = plan_arg.op[a]; uint64_t x = member ? v[a] : pa_identity(op);
#pragma unroll
          for (int d = 32; d > 0; d >>= 1) { uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)x, d, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), d, 64); x = pa_combine(op, x, ((uint64_t)hi << 32) | lo); }
          if (lane == lead) v[a] = x;
        }
        if (lane == lead) { cntv = (uint32_t)__popcll(mem); rowv = r; } else if (member) mine = false;
      }
    }
    uint32_t s = C; bool fresh = false;
    if (mine) {
      if (k != PA_EMPTY) {
        s = pa_slot(k, cbits);
        for (;;) {
          unsigned long long old = keys[s];
          if (old == PA_EMPTY) { old = atomicCAS(&keys[s], (unsigned long long)PA_EMPTY, (unsigned long long)k); if (old == PA_EMPTY) { fresh = true; break; } }
          if (old == k) break;
          s = (s + 1) & M;
        }
      }
      if (rowv < first[s]) atomicMin(&first[s], rowv);           // a stale read only costs a redundant atomic
      atomicAdd(&cnt[s], cntv);
#pragma unroll
      for (int a = 0; a < PA_MAX_AGGS; a++) if (a < na) {
        const int op = plan_arg.op[a];
        if (!I128) pa_apply(op, &acc[(size_t)a * C1 + s], v[a]);
        else if (op == PA_SUM_I128_LO && plan_arg.op[a + 1 < PA_MAX_AGGS ? a + 1 : a] == PA_SUM_I128_SX) {            // value = sign-extended 64 bits
          const int ah = a + 1 < PA_MAX_AGGS ? a + 1 : a;
          const unsigned long long lo = v[a], old = atomicAdd(&acc[(size_t)a * C1 + s], lo);
          const unsigned long long hi = (unsigned long long)((long long)lo >> 63) + (unsigned long long)(old + lo < old);
          if (hi) atomicAdd(&acc[(size_t)ah * C1 + s], hi);
        } else if (op == PA_SUM_I128_LO) {            // a + 1 is the high word's cell
          const int ah = a + 1 < PA_MAX_AGGS ? a + 1 : a;      // a LO cell is never the last one (the host lays pairs out); the clamp keeps the unrolled index in range
          const unsigned long long lo = v[a], old = atomicAdd(&acc[(size_t)a * C1 + s], lo);
          const unsigned long long hi = v[ah] + (unsigned long long)(old + lo < old);
          if (hi) atomicAdd(&acc[(size_t)ah * C1 + s], hi);
        } else if (op != PA_SUM_I128_HI && op != PA_SUM_I128_SX) pa_apply(op, &acc[(size_t)a * C1 + s], v[a]);
      }
    }
    uint64_t fb = ballot64(fresh);
    if (lane == 0 && fb) atomicAdd(&nfilled, (uint32_t)__popcll(fb));
    __syncthreads();
  }
  flush();
}

// sample of the batch: distinct keys among `s` evenly spaced rows (open-addressing u64 table of `cap` slots) and how many sampled
// rows are not smaller than their predecessor row (clustered input keeps the run-numbering path of groups.hip)
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_pa_sample(const T* keys, const uint64_t* mask, int64_t n, int64_t s, int64_t stride, unsigned long long* table, uint64_t cap_mask, unsigned long long* out /*[0] distinct, [1] non-decreasing pairs, [2] pairs*/) {
  uint32_t nf = 0, nn = 0, np = 0;
  for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < s; j += (int64_t)gridDim.x * BLOCK) {
    int64_t i = j * stride;
    if (i >= n || (mask && !bit_get(mask, i))) continue;
    T kt = keys[i]; if (i > 0) { np++; nn += keys[i - 1] <= kt; }
    uint64_t k = (uint64_t)(int64_t)kt;
    uint64_t kk = k == PA_EMPTY ? 0x5555555555555555ull : k;                 // the marker value itself: folded onto another key (estimate only)
    uint64_t h = mix64(kk) & cap_mask;
    for (;;) {
      unsigned long long old = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // frequent keys: no atomic once they are in
      if (old == PA_EMPTY) { old = atomicCAS(&table[h], (unsigned long long)PA_EMPTY, (unsigned long long)kk); if (old == PA_EMPTY) { nf++; break; } }
      if (old == kk) break;
      h = (h + 1) & cap_mask;
    }
  }
  nf = wave_sum(nf); nn = wave_sum(nn); np = wave_sum(np);                    // one atomic per workgroup and counter: same-address atomics serialise
  __shared__ uint32_t sh[3][BLOCK / WAVE];
  if (lane_id() == 0) { sh[0][threadIdx.x >> 6] = nf; sh[1][threadIdx.x >> 6] = nn; sh[2][threadIdx.x >> 6] = np; }
  __syncthreads();
  if (threadIdx.x < 3) { uint32_t t = 0; for (int w = 0; w < BLOCK / WAVE; w++) t += sh[threadIdx.x][w]; if (t) atomicAdd(&out[threadIdx.x], (unsigned long long)t); }
}

// ---- two-level partition (more groups than 2048 partitions bring into LDS): pass 1 splits on the high hash bits into P1 partitions, pass 2 is a STABLE
// split of pass 1's output on the low hash bits into P2 <= 256 -- an LSD radix sort on two digits, so the rows end up ordered by (p2, p1): P1 * P2
// contiguous partitions.  Their boundaries are read off the partitioned keys.
struct RpHashU64Low {
  const uint64_t* keys;
  __device__ inline bool operator()(int64_t i, uint32_t P, uint32_t* pid, uint64_t* key) const { *key = keys[i]; *pid = (uint32_t)(((mix64(*key) & 0xFFFFFFFFull) * (uint64_t)P) >> 32); return true; }
};
// starts[f] = first row of fine partition f (starts[Pt] = m): after the two passes the rows are ordered by fine partition, so every bound is a binary search over
// the keys (27 probes for 100 M rows, ~9 000 partitions) instead of a pass over them; k_pa_aggregate verifies the order it relies on row by row
__global__ void __launch_bounds__(BLOCK) k_pa_bounds(const uint64_t* __restrict__ pkey, uint32_t m, uint32_t P1, uint32_t P2, uint32_t Pt, uint32_t* __restrict__ starts /*[Pt + 1]*/) {
  const uint32_t f = blockIdx.x * BLOCK + threadIdx.x; if (f > Pt) return;
  uint32_t lo = 0, hi = m;                                      // first row whose fine partition is >= f
  while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (pa_fine_pid(pkey[mid], P1, P2) < f) lo = mid + 1; else hi = mid; }
  starts[f] = f == Pt ? m : lo;
}

// first-seen order without a sort: every partial row's first input row is a distinct row number, so its rank among them is the number of marked rows in front of it --
// mark the rows in a bitmap over the input, prefix-count the words, read the rank back
__global__ void __launch_bounds__(BLOCK) k_pa_mark(const uint32_t* __restrict__ first, int64_t m, unsigned long long* bits) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i < m) { const uint32_t f = first[i]; atomicOr(&bits[f >> 6], 1ull << (f & 63)); }
}
__global__ void __launch_bounds__(BLOCK) k_pa_popc(const uint64_t* __restrict__ bits, int64_t nw, uint32_t* __restrict__ pc) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i < nw) pc[i] = (uint32_t)__popcll(bits[i]);
}
__global__ void __launch_bounds__(BLOCK) k_pa_rank_perm(const uint32_t* __restrict__ first, int64_t m, const uint64_t* __restrict__ bits, const uint32_t* __restrict__ pref, uint32_t* __restrict__ perm) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i >= m) return;
  const uint32_t f = first[i]; const uint32_t r = pref[f >> 6] + (uint32_t)__popcll(bits[f >> 6] & ((1ull << (f & 63)) - 1ull));
  perm[r] = (uint32_t)i;
}

// Many partial rows (millions): the inverse map.  inv[first row of partial row j] = j is one random 4-byte store per partial row into an n-entry array; the permutation in
// first-seen order is then the non-empty entries of inv in array order -- two streaming passes (count, write) instead of four passes of a stable sort over (first row, j) words
// (20 M partial rows of 100 M input rows: 1.3 ms of sort passes).  First rows are distinct (a row belongs to one partial row), so the stores never collide.
constexpr int INV_PER = 16;             // consecutive entries per lane
__global__ void __launch_bounds__(BLOCK) k_pa_inv(const uint32_t* __restrict__ first, int64_t m, uint32_t* __restrict__ inv) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i < m) inv[first[i]] = (uint32_t)i;
}
template <bool WRITE>
__global__ void __launch_bounds__(BLOCK) k_pa_inv_compact(const uint32_t* __restrict__ inv, int64_t n, uint32_t* __restrict__ counts, uint32_t* __restrict__ perm) {
  const int64_t base = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * INV_PER;
  uint32_t v[INV_PER]; uint32_t c = 0;
  if (base + INV_PER <= n) {
#pragma unroll
    for (int q = 0; q < INV_PER / 4; q++) { const uint4 x = ((const uint4*)(inv + base))[q]; v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w; }
  } else {
#pragma unroll
    for (int q = 0; q < INV_PER; q++) v[q] = base + q < n ? inv[base + q] : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int q = 0; q < INV_PER; q++) c += v[q] != 0xFFFFFFFFu;
  __shared__ uint32_t lds[BLOCK / WAVE];
  uint32_t tot; uint32_t ex = block_exclusive_sum<uint32_t>(c, lds, &tot);
  if (!WRITE) { if (threadIdx.x == 0) counts[blockIdx.x] = tot; return; }
  uint32_t o = counts[blockIdx.x] + ex;
#pragma unroll
  for (int q = 0; q < INV_PER; q++) if (v[q] != 0xFFFFFFFFu) perm[o++] = v[q];
}
// many partial rows, general form: (first row << jb | partial row) words through LSD passes of the stable partition, 8 bytes moving per row and pass
__global__ void __launch_bounds__(BLOCK) k_pa_order_words(const uint32_t* __restrict__ first, int64_t m, int jb, uint64_t* __restrict__ words) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i < m) words[i] = ((uint64_t)first[i] << jb) | (uint64_t)i;
}
__global__ void __launch_bounds__(BLOCK) k_pa_perm_of_words(const uint64_t* __restrict__ words, int64_t m, int jb, uint32_t* __restrict__ perm) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i < m) perm[i] = (uint32_t)(words[i] & ((1ull << jb) - 1ull));
}
// every output column of the partial rows in one pass over the permutation: a row's record (key, count, cells) is one or two 32-byte reads
struct PaEmit { int32_t n; int32_t word[2 + 2 * PA_MAX_AGGS]; void* dst[2 + 2 * PA_MAX_AGGS]; int32_t narrow[2 + 2 * PA_MAX_AGGS]; int32_t dstride[2 + 2 * PA_MAX_AGGS], doff[2 + 2 * PA_MAX_AGGS]; };      // narrow: store the low 32 bits; dst[i * dstride + doff]
template <int RS>
__global__ void __launch_bounds__(BLOCK) k_pa_emit(PaEmit e, const uint64_t* __restrict__ recs, const uint32_t* __restrict__ perm, int64_t m) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i >= m) return;
  const uint32_t j = perm ? perm[i] : (uint32_t)i; uint64_t v[RS];
  const ulonglong2* r = (const ulonglong2*)(recs + (size_t)j * RS);
#pragma unroll
  for (int q = 0; q < RS / 2; q++) { const ulonglong2 t = r[q]; v[2 * q] = t.x; v[2 * q + 1] = t.y; }
#pragma unroll
  for (int c = 0; c < 2 + 2 * PA_MAX_AGGS; c++) if (c < e.n) {
    uint64_t x = 0;
#pragma unroll
    for (int w = 0; w < RS; w++) if (e.word[c] == w) x = v[w];
    if (e.narrow[c]) ((uint32_t*)e.dst[c])[i] = (uint32_t)x; else ((uint64_t*)e.dst[c])[(size_t)i * e.dstride[c] + e.doff[c]] = x;
  }
}

// ---- 2..4 integer key columns -> one u64: key = sum over columns of (value - min + (nullable ? 1 : 0)) * stride; digit 0 of a nullable column = NULL (its own group,
// group_values/row.rs:94-146 treats NULL as a value).  The ranges multiply to < 2^62, so the packed key never equals PA_EMPTY.
struct PaPackCols { int32_t n; const void* v[4]; const uint64_t* valid[4]; int32_t type[4]; long long mn[4]; unsigned long long stride[4], range[4]; int32_t nullable[4]; };
// One row per thread and step.  Measured per 100 M rows x 3 key columns (both kernels together): this form 1.67 ms; four rows per thread, consecutive rows 1.92 ms,
// lane-contiguous rows 1.94 ms -- neither unrolled form helped, so the simple one stays (the per-element type switch of key_at is the suspect, not measured apart).
// step > 1: every step-th row only (the optimistic ranges of a large batch come from a sample; k_pa_pack then checks every row against them)
__global__ void __launch_bounds__(BLOCK) k_pa_cols_minmax(PaPackCols pc, const uint64_t* mask, int64_t n, int64_t step, long long* mm /*[2 * n] min, max*/) {
  long long lo[4] = { INT64_MAX, INT64_MAX, INT64_MAX, INT64_MAX }, hi[4] = { INT64_MIN, INT64_MIN, INT64_MIN, INT64_MIN };
  for (int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * step; i < n; i += (int64_t)gridDim.x * BLOCK * step) {
    if (mask && !bit_get(mask, i)) continue;
#pragma unroll
    for (int c = 0; c < 4; c++) if (c < pc.n && valid_at(pc.valid[c], i)) { const long long x = key_at(pc.v[c], pc.type[c], i); lo[c] = x < lo[c] ? x : lo[c]; hi[c] = x > hi[c] ? x : hi[c]; }
  }
#pragma unroll
  for (int c = 0; c < 4; c++) if (c < pc.n) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const long long a = __shfl_xor(lo[c], d, 64), b = __shfl_xor(hi[c], d, 64); lo[c] = a < lo[c] ? a : lo[c]; hi[c] = b > hi[c] ? b : hi[c]; }
    if (lane_id() == 0 && lo[c] <= hi[c]) { atomicMin(&mm[2 * c], lo[c]); atomicMax(&mm[2 * c + 1], hi[c]); }       // one pair per wave
  }
}
// *outside (optional) is raised when a value lies outside [mn, mn + range): the ranges were an estimate and the caller packs again with exact ones
__global__ void __launch_bounds__(BLOCK) k_pa_pack(PaPackCols pc, int64_t n, uint64_t* out, unsigned long long* outside) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i >= n) return;
  unsigned long long k = 0; bool bad = false;
#pragma unroll
  for (int c = 0; c < 4; c++) if (c < pc.n) {
    const bool ok = valid_at(pc.valid[c], i);
    const unsigned long long off = ok ? (unsigned long long)(key_at(pc.v[c], pc.type[c], i) - pc.mn[c]) : 0ull;         // wraps to a huge value below mn
    bad |= ok && off >= pc.range[c] - (unsigned long long)pc.nullable[c];
    k += (ok ? off + (unsigned long long)pc.nullable[c] : 0ull) * pc.stride[c];
  }
  out[i] = bad ? 0ull : k;
  if (outside && __ballot(bad) && lane_id() == 0) *outside = 1ull;
}
// packed keys of the partial rows -> column c of the group keys (values as 64-bit patterns narrowed by the caller's type width; validity word by word)
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_pa_unpack(const uint64_t* packed, int64_t m, unsigned long long stride, unsigned long long range, long long mn, int nullable, T* out, uint64_t* valid) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; bool ok = true; 
  if (i < m) { const unsigned long long dgt = (packed[i] / stride) % range; ok = !nullable || dgt != 0; out[i] = ok ? (T)((long long)(dgt - (unsigned long long)nullable) + mn) : (T)0; }
  if (valid) { const uint64_t b = ballot64(i < m && ok); if (lane_id() == 0 && i < ((m + 63) / 64) * 64) valid[i >> 6] = b; }
}

struct PaFlagCols { int32_t n; const uint64_t* valid[8]; };
__global__ void __launch_bounds__(BLOCK) k_pa_flags(PaFlagCols fc, int64_t n, uint8_t* out) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i >= n) return;
  uint32_t f = 0;
#pragma unroll
  for (int b = 0; b < 8; b++) if (b < fc.n) f |= (uint32_t)bit_get(fc.valid[b], i) << b;
  out[i] = (uint8_t)f;
}
// validity bitmap of a state column: a group whose nullable argument held no value has a NULL state
__global__ void __launch_bounds__(BLOCK) k_pa_state_valid(const uint64_t* __restrict__ nvalid, int64_t m, uint64_t* __restrict__ valid) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const uint64_t b = ballot64(i < m && nvalid[i] != 0);
  if (lane_id() == 0 && i < ((m + 63) / 64) * 64) valid[i >> 6] = b;
}

static bool pa_key_type_ok(int32_t t) { return t == DFGPU_INT64 || t == DFGPU_UINT64 || t == DFGPU_INT32 || t == DFGPU_UINT32 || t == DFGPU_DATE32; }

}  // namespace dfgpu

using namespace dfgpu;
extern "C" dfgpu_status dfgpu_agg_preaggregate(dfgpu_ctx* ctx, const dfgpu_array* const* keys, int32_t nkeys, const int32_t* kinds, const dfgpu_array* const* values, int32_t n_aggs,
                                               const dfgpu_array* opt_mask, dfgpu_array** out_keys, dfgpu_array** out_states) {
  return dfgpu_agg_preaggregate_flags(ctx, keys, nkeys, kinds, values, nullptr, n_aggs, opt_mask, 0, out_keys, out_states);
}
extern "C" dfgpu_status dfgpu_agg_preaggregate_flags(dfgpu_ctx* ctx, const dfgpu_array* const* keys, int32_t nkeys, const int32_t* kinds, const dfgpu_array* const* values, const int32_t* value_casts, int32_t n_aggs,
                                                     const dfgpu_array* opt_mask, int32_t flags, dfgpu_array** out_keys, dfgpu_array** out_states) {
  return guard(ctx, [&] {
    const bool first_seen = ctx->first_seen_group_order && !(flags & DFGPU_PREAGG_ANY_ORDER);
    const bool verdict_only = out_keys == nullptr;         // would this batch be taken?  (key column + selection only; see include/dfgpu.h)
    if (!keys || (!verdict_only && n_aggs && (!kinds || !values || !out_states))) fail(DFGPU_INVALID_ARGUMENT, "agg_preaggregate: null argument");
    if (verdict_only) n_aggs = 0;
    auto skip = [&](const char* why) { ctx->pa_sample_key = nullptr; ctx->pa_pack.reset(); fail(DFGPU_NOT_IMPLEMENTED, "agg_preaggregate: %s", why); };      // a skipped batch's sample must not answer for the next batch at a recycled address
    if (!ctx->agg_partitioned) skip("switched off (option agg_partitioned)");
    if (nkeys < 1 || nkeys > 4) skip("one to four key columns");
    const dfgpu_array* key = keys[0]; const int64_t n = key->length;
    // A dictionary key column is pre-aggregated by its CODES: codes with equal dictionary values leave as separate partial rows, which
    // interning the emitted dictionary array merges (dictionary keys intern by value).
    const bool is_dict = nkeys == 1 && key->type == DFGPU_DICTIONARY;
    // Several key columns, or one with NULLs, travel as ONE packed u64 (value ranges multiplied out, NULL = digit 0 of its column): the partial rows' keys are unpacked again
    const bool packed_keys = nkeys > 1 || (!is_dict && key->validity);
    auto plain_int = [](int32_t t) { switch (t) { case DFGPU_INT8: case DFGPU_INT16: case DFGPU_INT32: case DFGPU_INT64: case DFGPU_DATE32: case DFGPU_UINT8: case DFGPU_UINT16: case DFGPU_UINT32: case DFGPU_UINT64: return true; default: return false; } };
    if (packed_keys) { for (int c = 0; c < nkeys; c++) if (!keys[c] || !plain_int(keys[c]->type) || keys[c]->length != n) skip("integer / Date32 key columns (not dictionary-encoded) when there are several or one carries NULLs"); }
    int32_t ktype = is_dict ? key->key_type : key->type;
    if (!packed_keys && (!pa_key_type_ok(ktype) || key->validity)) skip("a 4- or 8-byte integer key column (or dictionary codes of that width)");
    if (n < ctx->agg_partitioned_min_rows || n > 0xFFFF0000ll) skip("batch below agg_partitioned_min_rows");
    if (n_aggs > 16) skip("at most 16 aggregates");
    // accumulator plan: one 8-byte LDS cell per SUM / MIN / MAX; COUNT and AVG counts come from the row count (value columns carry no NULLs)
    // A nullable argument (accumulate.rs:126-233: NULL values take no part; a group that saw none has a NULL state): its validity travels as one bit of a per-row flag byte,
    // its cells skip the NULL rows, and one COUNT_FLAG cell per nullable column counts the values seen -- COUNT(x), AVG's count and the validity of SUM / MIN / MAX states.
    PaPlan plan{}; int cell_of[16], nvalid_of[16]; const dfgpu_array* cell_src[PA_MAX_AGGS]; int cell_word[PA_MAX_AGGS]; bool cell_cast[PA_MAX_AGGS] = {};      // cell c reads word cell_word[c] of its source's rows (cell_cast: of their doubles)
    // value_casts[i] == DFGPU_FLOAT64 over an integer column: the argument is CAST(column AS DOUBLE) (what AVG / SUM over an integer column are planned as); the column is
    // converted while the partition moves it -- the cast's own pass (400 MB read, 800 MB written per 100 M Int32 rows: 0.40 ms) does not run
    auto cast_f64 = [&](int i) { return value_casts && value_casts[i] == DFGPU_FLOAT64 && values[i] && (values[i]->type == DFGPU_INT32 || values[i]->type == DFGPU_INT64); };
    for (int i = 0; i < n_aggs; i++) if (value_casts && value_casts[i] != 0 && !cast_f64(i)) skip("an argument cast other than Int32 / Int64 -> Float64");
    for (int c = 0; c < PA_MAX_AGGS; c++) { plan.flag_bit[c] = -1; cell_src[c] = nullptr; }
    const dfgpu_array* flag_src[8]; int n_flags = 0;
    auto flag_of = [&](const dfgpu_array* v) { for (int b = 0; b < n_flags; b++) if (flag_src[b] == v) return b; if (n_flags == 8) skip("at most 8 nullable value columns"); flag_src[n_flags] = v; return n_flags++; };
    auto count_cell = [&](const dfgpu_array* v) {           // the COUNT_FLAG cell of a nullable column (one per column)
      const int b = flag_of(v);
      for (int j = 0; j < plan.n_acc; j++) if (plan.op[j] == PA_COUNT_FLAG && plan.flag_bit[j] == b) return j;
      if (plan.n_acc + 1 > PA_MAX_AGGS) skip("at most 6 accumulator cells (a nullable argument takes one more for its count)");
      const int c = plan.n_acc++; plan.op[c] = PA_COUNT_FLAG; plan.flag_bit[c] = b; cell_src[c] = nullptr; cell_word[c] = 0; return c;
    };
    for (int i = 0; i < n_aggs; i++) {
      const dfgpu_array* v = values[i]; cell_of[i] = -1; nvalid_of[i] = -1;
      if (v && (v->length != n || v->type == DFGPU_DICTIONARY)) skip("value columns of the batch's length, not dictionary-encoded");
      if (v && v->validity) nvalid_of[i] = count_cell(v);
      if (kinds[i] == DFGPU_AGG_COUNT) continue;
      if (!v) skip("aggregate without an argument");
      int op = PA_NONE;
      const bool asf = cast_f64(i);
      const bool i64 = !asf && v->type == DFGPU_INT64, u64 = !asf && v->type == DFGPU_UINT64, f64 = asf || v->type == DFGPU_FLOAT64, d128 = v->type == DFGPU_DECIMAL128;
      if (!i64 && !u64 && !f64 && !d128) skip("Int64 / UInt64 / Float64 / Decimal128 aggregate arguments");
      switch (kinds[i]) {
        case DFGPU_AGG_SUM: op = d128 ? PA_SUM_I128_LO : f64 ? PA_SUM_F64 : PA_SUM_I64; break;
        case DFGPU_AGG_AVG: if (!f64 && !d128) skip("AVG over Float64 / Decimal128"); op = d128 ? PA_SUM_I128_LO : PA_SUM_F64; break;
        case DFGPU_AGG_MIN: if (d128) skip("MIN over Decimal128"); op = f64 ? PA_MIN_F64 : i64 ? PA_MIN_I64 : PA_MIN_U64; break;
        case DFGPU_AGG_MAX: if (d128) skip("MAX over Decimal128"); op = f64 ? PA_MAX_F64 : i64 ? PA_MAX_I64 : PA_MAX_U64; break;
        default: skip("SUM / AVG / COUNT / MIN / MAX");
      }
      int c = -1; for (int j = 0; j < plan.n_acc; j++) if (plan.op[j] == op && cell_src[j] == v && cell_cast[j] == asf) c = j;          // SUM(x) and AVG(x) share a cell
      if (c < 0) {
        const int need = d128 ? 2 : 1;
        if (plan.n_acc + need > PA_MAX_AGGS) skip("at most 6 accumulator cells (a Decimal128 sum takes two)");
        c = plan.n_acc; plan.n_acc += need; plan.op[c] = op; cell_src[c] = v; cell_word[c] = 0; cell_cast[c] = asf;
        const int fbit = v->validity ? flag_of(v) : -1; plan.flag_bit[c] = fbit;
        if (d128) { const bool fits64 = v->precision > 0 && v->precision <= 18;      // |unscaled value| < 10^18 < 2^63: the high word carries no information
          plan.op[c + 1] = fits64 ? PA_SUM_I128_SX : PA_SUM_I128_HI; cell_src[c + 1] = v; cell_word[c + 1] = fits64 ? 0 : 1; plan.flag_bit[c + 1] = fbit; plan.has_i128 = 1; }
      }
      cell_of[i] = c;
    }
    // the distinct value columns the partition moves (a Decimal128 column once, 16 bytes wide)
    const dfgpu_array* srcs[PA_MAX_AGGS]; bool src_cast[PA_MAX_AGGS] = {}; int n_src = 0, src_of[PA_MAX_AGGS];
    for (int c = 0; c < plan.n_acc; c++) { src_of[c] = -1; if (!cell_src[c]) continue; int j = -1; for (int q = 0; q < n_src; q++) if (srcs[q] == cell_src[c] && src_cast[q] == cell_cast[c]) j = q; if (j < 0) { j = n_src; src_cast[n_src] = cell_cast[c]; srcs[n_src++] = cell_src[c]; } src_of[c] = j; }
    const uint64_t* mk = nullptr; BufferPtr mask = effective_mask(ctx, opt_mask, n); if (mask) mk = (const uint64_t*)mask->ptr;
    // ---- sample: clustered? how many groups?  (a verdict-only call leaves its sample for the call that follows on the same column)
    const int64_t s = n < (1 << 19) ? n : (1 << 19), stride = n / s; const uint64_t cap = 1ull << 21;
    const void* ident = keys[0]->values->ptr;           // what the verdict-only call and the call that follows share
    const bool cached = ctx->pa_sample_key == ident && ctx->pa_sample_n == n && ctx->pa_sample_mask == (const void*)mk && (!packed_keys || (ctx->pa_pack && ctx->pa_pack_n == nkeys));
    BufferPtr packed; PaPackCols pc{};
    if (packed_keys) {
      pc.n = nkeys; for (int c = 0; c < nkeys; c++) { pc.v[c] = keys[c]->values->ptr; pc.valid[c] = keys[c]->validity ? (const uint64_t*)keys[c]->validity->ptr : nullptr; pc.type[c] = keys[c]->type == DFGPU_DATE32 ? DFGPU_INT32 : keys[c]->type; pc.nullable[c] = keys[c]->validity ? 1 : 0;
        const int w = type_width(keys[c]->type); if (w != 1 && w != 2 && w != 4 && w != 8) fail(DFGPU_INTERNAL, "agg_preaggregate: key column %d of type %d has no integer width", c, keys[c]->type); }      // the kernels read w bytes per row: checked here, not assumed there
      if (cached) { packed = ctx->pa_pack; for (int c = 0; c < nkeys; c++) { pc.mn[c] = ctx->pa_pack_min[c]; pc.stride[c] = ctx->pa_pack_stride[c]; pc.range[c] = ctx->pa_pack_range[c]; } }
      else {
        // Ranges: exact (one pass over the key columns) for small batches; for large ones the min / max of every step-th row, widened by their own width on either side -- the pack
        // pass checks every row against them and the exact pass only runs when a row falls outside (saves a pass of 16 B per row: 0.8 ms per 100 M rows x 3 columns)
        zero_scratch(ctx);
        const int64_t step0 = n >= ctx->agg_pack_estimate_min_rows ? std::max<int64_t>(2, n >> 19) : 1;
        bool done = false;
        for (int attempt = 0; attempt < 2 && !done; attempt++) {
          const int64_t step = attempt == 0 ? step0 : 1; const bool estimate = step > 1;
          long long init[8]; for (int c = 0; c < 4; c++) { init[2 * c] = INT64_MAX; init[2 * c + 1] = INT64_MIN; }
          HIP_CHECK(hipMemcpyAsync(ctx->d_scratch64 + 16, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
          HIP_CHECK(hipMemsetAsync(ctx->d_scratch64 + 24, 0, 8, ctx->stream));
          { KernelTimer kt_(ctx, "pa_pack");
            hipLaunchKernelGGL(k_pa_cols_minmax, dim3(grid_for((n + step - 1) / step, BLOCK * 8, ctx->num_cus * 8)), dim3(BLOCK), 0, ctx->stream, pc, mk, n, step, (long long*)(ctx->d_scratch64 + 16));
            KERNEL_CHECK(); }
          const uint64_t* mm = read_scratch_range(ctx, 16, 8);
          ctx->count_sync("sync:pa_pack_ranges");
          unsigned __int128 prod = 1;
          for (int c = nkeys - 1; c >= 0; c--) {
            __int128 lo = (long long)mm[2 * c], hi = (long long)mm[2 * c + 1]; if (lo > hi) lo = hi = 0;              // a column of NULLs only (or none sampled)
            if (estimate) { const __int128 w = hi - lo + 1; lo -= w; hi += w; if (lo < (__int128)INT64_MIN) lo = INT64_MIN; if (hi > (__int128)INT64_MAX) hi = INT64_MAX; }
            const unsigned __int128 range = (unsigned __int128)(hi - lo) + 1 + (unsigned)pc.nullable[c];
            if (range > ((unsigned __int128)1 << 62)) { prod = (unsigned __int128)1 << 100; break; }
            pc.mn[c] = (long long)lo; pc.range[c] = (unsigned long long)range; pc.stride[c] = (unsigned long long)prod; prod *= range;
            if (prod > ((unsigned __int128)1 << 62)) break;
          }
          if (prod > ((unsigned __int128)1 << 62)) { if (estimate) continue; skip("key value ranges multiply beyond 2^62 (no packed key)"); }      // the widened estimate does not fit: try the exact ranges
          if (!packed) packed = alloc_buffer(ctx, (size_t)n * 8);
          { KernelTimer kt_(ctx, "pa_pack");
            hipLaunchKernelGGL(k_pa_pack, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, pc, n, (uint64_t*)packed->ptr, estimate ? (unsigned long long*)(ctx->d_scratch64 + 24) : nullptr);
            KERNEL_CHECK(); }
          if (!estimate) { done = true; break; }
          if (read_scratch(ctx, 24) == 0) done = true;          // every row inside the estimated ranges
          else ctx->count_sync("sync:pa_pack_outside");
        }
      }
      ktype = DFGPU_UINT64;
    }
    const void* kptr = packed_keys ? packed->ptr : key->values->ptr;
    if (cached) { for (int q = 0; q < 3; q++) ctx->h_pinned[q] = ctx->pa_sample[q]; ctx->pa_sample_key = nullptr; ctx->pa_pack.reset(); }
    else {
    BufferPtr table = alloc_buffer(ctx, cap * 8); HIP_CHECK(hipMemsetAsync(table->ptr, 0xFF, cap * 8, ctx->stream));
    HIP_CHECK(hipMemsetAsync(ctx->d_scratch64, 0, 24, ctx->stream));          // the sample's three counters
    { KernelTimer kt_(ctx, "pa_sample");
#define PA_SAMPLE(T) hipLaunchKernelGGL((k_pa_sample<T>), dim3(grid_for(s, BLOCK * 8, 256)), dim3(BLOCK), 0, ctx->stream, (const T*)kptr, mk, n, s, stride, (unsigned long long*)table->ptr, cap - 1, (unsigned long long*)ctx->d_scratch64)
      switch (ktype) { case DFGPU_INT64: PA_SAMPLE(int64_t); break; case DFGPU_UINT64: PA_SAMPLE(uint64_t); break; case DFGPU_UINT32: PA_SAMPLE(uint32_t); break; default: PA_SAMPLE(int32_t); break; }
#undef PA_SAMPLE
      KERNEL_CHECK(); }
    ctx->count_sync("sync:pa_sample"); fetch_to_pinned(ctx, 0, ctx->d_scratch64, 24);
    if (verdict_only) {
      ctx->pa_sample_key = ident; ctx->pa_sample_n = n; ctx->pa_sample_mask = (const void*)mk; for (int q = 0; q < 3; q++) ctx->pa_sample[q] = ctx->h_pinned[q];
      ctx->pa_pack = packed; ctx->pa_pack_n = packed_keys ? nkeys : 0;
      for (int c = 0; c < nkeys && packed_keys; c++) { ctx->pa_pack_min[c] = pc.mn[c]; ctx->pa_pack_stride[c] = pc.stride[c]; ctx->pa_pack_range[c] = pc.range[c]; ctx->pa_pack_nullable[c] = pc.nullable[c] != 0; }
    }
    }
    const double d = (double)ctx->h_pinned[0], nondec = (double)ctx->h_pinned[1], pairs = (double)ctx->h_pinned[2], ss = pairs + 1;
    if (pairs > 0 && nondec >= 0.98 * pairs && !ctx->agg_partitioned_force) skip("keys arrive clustered (run numbering is cheaper)");
    // distinct keys D of the batch from d distinct among ss sampled rows: d = D (1 - exp(-ss / D))
    double D = d;
    if (d >= 0.999 * ss) D = 1e12; else { double lo = d, hi = 1e12; for (int it = 0; it < 200; it++) { double mid = 0.5 * (lo + hi); double e = mid * (1.0 - exp(-ss / mid)); if (e < d) lo = mid; else hi = mid; } D = 0.5 * (lo + hi); }
    if (D > (double)n) D = (double)n;
    if (!ctx->agg_partitioned_force) {
      if (D < 3000) skip("few groups (the LDS cache of the accumulators holds them)");
      if (D > (double)n / 3.0) skip("fewer than three rows per group (pre-aggregation would not reduce the batch)");
    }
    if (verdict_only) return;
    const int cell_bytes = 16 + 8 * plan.n_acc; int cbits = 12; while (cbits > 6 && (((size_t)1 << cbits) + 1) * cell_bytes > 150 * 1024) cbits--;
    // groups a partition may hold: 0.55 of the table, and not more than the table takes before the kernel flushes it early (k_pa_aggregate: nfilled + PA_NT > C - C / 8 --
    // with five or six cells the table has 2048 slots and that bound, 768, is the smaller one: at 0.55 x 2048 every partition was cut and its keys left in two partial rows)
    const double table = (double)(1u << cbits), room = table - table / 8 - (double)PA_NT;
    const double per_part = room > 0 && 0.85 * room < 0.55 * table ? 0.85 * room : 0.55 * table;
    const double Dp = D > 4.0 * d ? D : 4.0 * d;          // skewed keys: the uniform model underestimates the tail, stay on the many-partitions side
    int64_t P = (int64_t)(Dp / per_part) + 1; if (P < 64) P = 64;
    // beyond 2048 partitions one pass writes bursts too short to pay (measured: P = 8192 slower than the atomics it replaces): two passes, P1 x P2
    const bool two_level = P > 2048;
    int64_t P1 = P, P2 = 1;
    if (two_level) { P2 = P > 2048 * 128 ? 256 : 128; P1 = (P + P2 - 1) / P2; if (P1 < 16) P1 = 16; if (P1 > 2048) P1 = 2048; P = P1 * P2; }
    else { if (P > n / 2048 + 1) P = n / 2048 + 1; if (P > ctx->num_cus) P = std::min<int64_t>(2048, (P + ctx->num_cus - 1) / ctx->num_cus * ctx->num_cus); P1 = P; }
    // ---- partition (key, row, value cells)
    // the row number travels with a row only to find every group's first row (first-seen order): with DFGPU_PREAGG_ANY_ORDER it stays behind -- 4 of the 20..28 bytes a row
    // costs each partition level and the aggregation's read
    BufferPtr pkey = alloc_buffer(ctx, (size_t)n * 8), prow = first_seen ? alloc_buffer(ctx, (size_t)n * 4) : BufferPtr(); std::vector<BufferPtr> pval((size_t)n_src);
    // the flag byte of every row: bit b = nullable value column b holds a value there
    BufferPtr flags_in, pflag;
    if (n_flags) {
      flags_in = alloc_buffer(ctx, (size_t)n + 64); pflag = alloc_buffer(ctx, (size_t)n + 64);
      PaFlagCols fc{}; fc.n = n_flags; for (int b = 0; b < n_flags; b++) fc.valid[b] = (const uint64_t*)flag_src[b]->validity->ptr;
      hipLaunchKernelGGL(k_pa_flags, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, fc, n, (uint8_t*)flags_in->ptr);
      KERNEL_CHECK();
    }
    RpCols cols{}; cols.n = 1 + n_src + (n_flags ? 1 : 0); cols.rowid_dst = prow ? (uint32_t*)prow->ptr : nullptr;
    if (n_flags) cols.c[1 + n_src] = RpCol{ flags_in->ptr, pflag->ptr, 1, RP_RAW, 0 };
    cols.c[0] = RpCol{ kptr, pkey->ptr, 8, RP_HASHKEY, ktype };
    auto lo16 = [&](int j) { return srcs[j]->type == DFGPU_DECIMAL128 && srcs[j]->precision > 0 && srcs[j]->precision <= 18; };
    auto src_width = [&](int j) { return srcs[j]->type == DFGPU_DECIMAL128 && !lo16(j) ? 16 : 8; };
    for (int j = 0; j < n_src; j++) { pval[(size_t)j] = alloc_buffer(ctx, (size_t)n * (size_t)src_width(j)); cols.c[1 + j] = src_cast[j] ? RpCol{ srcs[j]->values->ptr, pval[(size_t)j]->ptr, 8, RP_CASTF64, srcs[j]->type } : RpCol{ srcs[j]->values->ptr, pval[(size_t)j]->ptr, 8 * (src_width(j) / 8), lo16(j) ? RP_LO16 : RP_RAW, 0 }; }
    auto bind_cells = [&]() { plan.vflag = n_flags ? (const uint8_t*)pflag->ptr : nullptr;
      for (int c = 0; c < plan.n_acc; c++) { const int j = src_of[c]; if (j < 0) { plan.vstride[c] = 1; plan.val[c] = nullptr; continue; } plan.vstride[c] = src_width(j) / 8; plan.val[c] = (const uint64_t*)pval[(size_t)j]->ptr + cell_word[c]; } };
    bind_cells();
    RpResult r;
#define PA_PART(T) r = rp_partition(ctx, RpHashInt<T>{ (const T*)kptr, nullptr, mk }, n, (uint32_t)P1, cols, false, ctx->d_scratch64 + 9, "pa_hist", "pa_scan", "pa_scatter")
    switch (ktype) { case DFGPU_INT64: case DFGPU_UINT64: PA_PART(int64_t); break; case DFGPU_UINT32: PA_PART(uint32_t); break; default: PA_PART(int32_t); break; }
#undef PA_PART
    packed.reset();
    if (two_level) {
      const int64_t m1 = mk ? (int64_t)read_scratch(ctx, 9) : n;          // rows the selection kept
      BufferPtr pkey2 = alloc_buffer(ctx, (size_t)n * 8), prow2 = prow ? alloc_buffer(ctx, (size_t)n * 4) : BufferPtr(); std::vector<BufferPtr> pval2((size_t)n_src);
      BufferPtr pflag2; if (n_flags) pflag2 = alloc_buffer(ctx, (size_t)n + 64);
      const int c0 = prow ? 2 : 1;          // columns in front of the value columns: key (, row number)
      RpCols c2{}; c2.n = c0 + n_src + (n_flags ? 1 : 0);
      if (n_flags) c2.c[c0 + n_src] = RpCol{ pflag->ptr, pflag2->ptr, 1, RP_RAW, 0 };
      c2.c[0] = RpCol{ pkey->ptr, pkey2->ptr, 8, RP_RAW, 0 }; if (prow) c2.c[1] = RpCol{ prow->ptr, prow2->ptr, 4, RP_RAW, 0 };
      for (int j = 0; j < n_src; j++) { pval2[(size_t)j] = alloc_buffer(ctx, (size_t)n * (size_t)src_width(j)); c2.c[c0 + j] = RpCol{ pval[(size_t)j]->ptr, pval2[(size_t)j]->ptr, src_width(j), RP_RAW, 0 }; }
      (void)rp_partition(ctx, RpHashU64Low{ (const uint64_t*)pkey->ptr }, m1, (uint32_t)P2, c2, true, ctx->d_scratch64 + 10, "pa_hist2", "pa_scan2", "pa_scatter2", true, true);
      pkey = pkey2; prow = prow2; for (int j = 0; j < n_src; j++) pval[(size_t)j] = pval2[(size_t)j];
      if (n_flags) pflag = pflag2;
      bind_cells();
      HIP_CHECK(hipMemsetAsync(ctx->d_scratch64 + 11, 0, 8, ctx->stream));
      r.starts = alloc_buffer(ctx, (size_t)(P + 1) * 4); r.P = (uint32_t)P;
      { KernelTimer kt_(ctx, "pa_bounds");
        hipLaunchKernelGGL(k_pa_bounds, dim3(grid_for(P + 1, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)pkey->ptr, (uint32_t)m1, (uint32_t)P1, (uint32_t)P2, (uint32_t)P, (uint32_t*)r.starts->ptr);
        KERNEL_CHECK(); }
    }
    // ---- aggregate every partition out of LDS
    // a partial row leaves the table as ONE record (key, count, cells; 4 or 8 words): the emit gathers a row with one or two sector reads instead of one per column
    const int rs = 2 + plan.n_acc <= 4 ? 4 : 8;
    BufferPtr orec = alloc_buffer(ctx, (size_t)n * (size_t)rs * 8), ofirst = alloc_buffer(ctx, (size_t)n * 4);
    HIP_CHECK(hipMemsetAsync(ctx->d_scratch64 + 12, 0, 24, ctx->stream));
    hipLaunchKernelGGL(k_pa_max_len, dim3((unsigned)((P + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)r.starts->ptr, (uint32_t)P, (unsigned long long*)(ctx->d_scratch64 + 13));
    KERNEL_CHECK();
    int64_t slice = (n / P + 1) * 3 / 2; if (slice < 65536) slice = 65536;          // uniform keys never split (a partition is within a percent of the average) if (slice > 0x7FFFFFFF) slice = 0x7FFFFFFF;
    // whether any partition is longer than a slice (skewed keys) is the device's to know: the (partition, slice) work list is always built, the longest partition comes back
    // with the aggregate's own read-back below (one host round trip less per call)
    { KernelTimer kt_(ctx, "pa_aggregate");
      BufferPtr items = alloc_buffer(ctx, (size_t)(P + 1) * 4);            // sum over partitions of ceil(len / slice) <= P + n / slice
      hipLaunchKernelGGL(k_pa_items, dim3(1), dim3(PA_NT), 0, ctx->stream, (const uint32_t*)r.starts->ptr, (uint32_t)P, (uint32_t)slice, (uint32_t*)items->ptr);
      const unsigned grid = (unsigned)(P + n / slice + 1);
      const size_t lds = (((size_t)1 << cbits) + 1) * cell_bytes;
#define PA_AGG(I, F) do { HIP_CHECK(hipFuncSetAttribute((const void*)k_pa_aggregate<I, F>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));      /* per device: set on every call, no process-wide flag */ \
        hipLaunchKernelGGL((k_pa_aggregate<I, F>), dim3(grid), dim3(PA_NT), lds, ctx->stream, (const uint64_t*)pkey->ptr, prow ? (const uint32_t*)prow->ptr : (const uint32_t*)nullptr, plan, (const uint32_t*)r.starts->ptr, \
                           items ? (const uint32_t*)items->ptr : nullptr, (uint32_t)P, cbits, (uint32_t)slice, (uint64_t*)orec->ptr, rs, (uint32_t*)ofirst->ptr, (unsigned long long*)(ctx->d_scratch64 + 12), \
                           two_level ? (uint32_t)P1 : 0u, (uint32_t)P2, (uint32_t*)(ctx->d_scratch64 + 11)); } while (0)
      if (n_flags) { if (plan.has_i128) PA_AGG(true, true); else PA_AGG(false, true); }
      else { if (plan.has_i128) PA_AGG(true, false); else PA_AGG(false, false); }
#undef PA_AGG
      KERNEL_CHECK(); }
    const uint64_t* back = read_scratch_range(ctx, 11, 4);          // [0] rows out of order, [1] partial rows written, [2] longest partition, [3] tables flushed early: one read-back
    const int64_t m = (int64_t)back[1]; const uint64_t early = back[3]; const int64_t n_slices = back[2] ? ((int64_t)back[2] + slice - 1) / slice : 1;
    if (two_level && (uint32_t)back[0] != 0) fail(DFGPU_INTERNAL, "agg_preaggregate: the two-level partition left rows out of partition order");
    // every key left in exactly one partial row unless a hot partition was cut into slices or a table overflowed mid-partition: the plan layer then
    // needs no hash table to number the groups of a first batch (option "agg_preaggregate_distinct", read only)
    ctx->pa_last_distinct = n_slices == 1 && !is_dict && early == 0;          // dictionary codes: two codes may carry one value
    pkey.reset(); prow.reset(); pval.clear(); pflag.reset(); flags_in.reset();
    // ---- partial rows in first-seen order of their groups
    BufferPtr perm = alloc_buffer(ctx, (size_t)(m + 1) * 4);
    if (first_seen) { KernelTimer kt_(ctx, "pa_order");
      if (first_seen && m >= 32768 && m <= (4 << 20)) {       // beyond a few million rows the random atomics and rank look-ups lose to the sort (20 M: 1.44 ms against 1.28)
        const int64_t nw = (n + 63) / 64;
        BufferPtr bits = alloc_buffer(ctx, (size_t)nw * 8, true), pref = alloc_buffer(ctx, (size_t)(nw + 1) * 4);
        hipLaunchKernelGGL(k_pa_mark, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)ofirst->ptr, m, (unsigned long long*)bits->ptr);
        hipLaunchKernelGGL(k_pa_popc, dim3(grid_for(nw, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)bits->ptr, nw, (uint32_t*)pref->ptr);
        exclusive_scan_u32_inplace32(ctx, (uint32_t*)pref->ptr, nw, nullptr);
        hipLaunchKernelGGL(k_pa_rank_perm, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)ofirst->ptr, m, (const uint64_t*)bits->ptr, (const uint32_t*)pref->ptr, (uint32_t*)perm->ptr);
        KERNEL_CHECK();
      } else if (first_seen && m > (4 << 20) && ctx->agg_order_inverse_map) {
        BufferPtr inv = alloc_buffer(ctx, (size_t)n * 4 + 64);
        HIP_CHECK(hipMemsetAsync(inv->ptr, 0xFF, (size_t)n * 4, ctx->stream));
        hipLaunchKernelGGL(k_pa_inv, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)ofirst->ptr, m, (uint32_t*)inv->ptr);
        const int64_t nblk = grid_for(n, BLOCK * INV_PER);
        BufferPtr cnts = alloc_buffer(ctx, (size_t)(nblk + 1) * 4);
        hipLaunchKernelGGL((k_pa_inv_compact<false>), dim3((unsigned)nblk), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)inv->ptr, n, (uint32_t*)cnts->ptr, (uint32_t*)nullptr);
        exclusive_scan_u32_inplace32(ctx, (uint32_t*)cnts->ptr, nblk, nullptr);
        hipLaunchKernelGGL((k_pa_inv_compact<true>), dim3((unsigned)nblk), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)inv->ptr, n, (uint32_t*)cnts->ptr, (uint32_t*)perm->ptr);
        KERNEL_CHECK();
      } else if (first_seen && m > (4 << 20)) {
        int jb = 1; while (((uint64_t)(m - 1) >> jb) != 0) jb++;
        int fb = 1; while (((uint64_t)(n - 1) >> fb) != 0) fb++;
        BufferPtr w0 = alloc_buffer(ctx, (size_t)m * 8), w1 = alloc_buffer(ctx, (size_t)m * 8);
        hipLaunchKernelGGL(k_pa_order_words, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)ofirst->ptr, m, jb, (uint64_t*)w0->ptr);
        uint64_t* wa = (uint64_t*)w0->ptr; uint64_t* wb = (uint64_t*)w1->ptr;
        const int npass = (fb + 7) / 8, dbits = (fb + npass - 1) / npass;
        for (int shift = 0; shift < fb; shift += dbits) {
          const int bits = fb - shift < dbits ? fb - shift : dbits;
          RpCols rc{}; rc.n = 1; rc.c[0] = RpCol{ nullptr, wb, 8, RP_HASHKEY, 0 };
          (void)rp_partition(ctx, RpHashDigit{ wa, shift + jb, (1u << bits) - 1u }, m, 1u << bits, rc, true, ctx->d_scratch64 + 9, "pa_order_hist", "pa_order_scan", "pa_order_scatter", false);
          std::swap(wa, wb);
        }
        hipLaunchKernelGGL(k_pa_perm_of_words, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)wa, m, jb, (uint32_t*)perm->ptr);
        KERNEL_CHECK();
      } else if (first_seen) {
        launch_iota_u32(ctx, (uint32_t*)perm->ptr, m, 0);
        radix_sort_pairs_u32(ctx, (uint32_t*)ofirst->ptr, (uint32_t*)perm->ptr, m, 32);
      } }
    KernelTimer kt_(ctx, "pa_emit");
    const uint32_t* pp = first_seen ? (const uint32_t*)perm->ptr : nullptr; dim3 grid(grid_for(m, BLOCK));          // any order: the records leave as the partitions wrote them
    ArrayHolder ok(new_fixed(ctx, ktype, m));
    PaEmit em{}; auto add = [&](int word, void* dst, int narrow, int dstride = 1, int doff = 0) { if (em.n >= 2 + 2 * PA_MAX_AGGS) fail(DFGPU_NOT_IMPLEMENTED, "agg_preaggregate: more output columns than one emit pass writes"); em.word[em.n] = word; em.dst[em.n] = dst; em.narrow[em.n] = narrow; em.dstride[em.n] = dstride; em.doff[em.n] = doff; em.n++; };
    add(0, ok.get()->values->ptr, (ktype == DFGPU_INT64 || ktype == DFGPU_UINT64) ? 0 : 1);
    std::vector<ArrayHolder> st((size_t)n_aggs * 2);
    // the values-seen count of every nullable column, once (UInt64 words): COUNT(x) / AVG counts copy it, state validities derive from it
    std::vector<BufferPtr> nval_buf((size_t)PA_MAX_AGGS);
    for (int c = 0; c < plan.n_acc; c++) if (plan.op[c] == PA_COUNT_FLAG) { nval_buf[(size_t)c] = alloc_buffer(ctx, (size_t)(m + 1) * 8); add(2 + c, nval_buf[(size_t)c]->ptr, 0); }
    std::vector<std::pair<dfgpu_array*, int>> need_valid;          // (state array, its COUNT_FLAG cell)
    for (int i = 0; i < n_aggs; i++) {
      auto counts_as = [&](int32_t type) { dfgpu_array* a = new_fixed(ctx, type, m); add(nvalid_of[i] >= 0 ? 2 + nvalid_of[i] : 1, a->values->ptr, 0); return a; };
      auto cell_as = [&](int32_t type) { dfgpu_array* a = new_fixed(ctx, type, m); add(2 + cell_of[i], a->values->ptr, 0); return a; };
      // Decimal128 sums: state type Decimal128(min(38, p + 10), s) (sum.rs:75-86, average.rs:96-110); the two cells interleave into 16-byte values
      auto dec_as = [&]() { const dfgpu_array* v = values[i]; dfgpu_array* a = new_fixed(ctx, DFGPU_DECIMAL128, m, std::min(38, v->precision + 10), v->scale); add(2 + cell_of[i], a->values->ptr, 0, 2, 0); add(3 + cell_of[i], a->values->ptr, 0, 2, 1); return a; };
      const bool d128 = values[i] && values[i]->type == DFGPU_DECIMAL128;
      if (kinds[i] == DFGPU_AGG_COUNT) st[(size_t)2 * i].a = counts_as(DFGPU_INT64);                                                       // count.rs: Int64 state
      else if (kinds[i] == DFGPU_AGG_AVG) { st[(size_t)2 * i].a = counts_as(DFGPU_UINT64); st[(size_t)2 * i + 1].a = d128 ? dec_as() : cell_as(DFGPU_FLOAT64); }   // average.rs:392-430: (counts, sums)
      else if (d128) st[(size_t)2 * i].a = dec_as();
      else st[(size_t)2 * i].a = cell_as(cast_f64(i) ? DFGPU_FLOAT64 : values[i]->type);          // SUM / MIN / MAX of an 8-byte type: state type == input type (sum.rs:75-86, min_max.rs:102-139)
      if (nvalid_of[i] >= 0 && kinds[i] != DFGPU_AGG_COUNT) need_valid.emplace_back(st[(size_t)2 * i + (kinds[i] == DFGPU_AGG_AVG ? 1 : 0)].get(), nvalid_of[i]);
      if (em.n > 2 + 2 * PA_MAX_AGGS) fail(DFGPU_NOT_IMPLEMENTED, "agg_preaggregate: more output columns than one emit pass writes");
    }
    if (m) { if (rs == 4) hipLaunchKernelGGL(k_pa_emit<4>, grid, dim3(BLOCK), 0, ctx->stream, em, (const uint64_t*)orec->ptr, pp, m); else hipLaunchKernelGGL(k_pa_emit<8>, grid, dim3(BLOCK), 0, ctx->stream, em, (const uint64_t*)orec->ptr, pp, m); }
    KERNEL_CHECK();
    for (auto& nv : need_valid) {          // NullState::build (accumulate.rs:328-356): the state of a group that saw no value is NULL
      nv.first->validity = alloc_buffer(ctx, bitmap_bytes(m) + 8); nv.first->null_count = -1;
      if (m) hipLaunchKernelGGL(k_pa_state_valid, dim3(grid_for(((m + 63) / 64) * 64, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)nval_buf[(size_t)nv.second]->ptr, m, (uint64_t*)nv.first->validity->ptr);
      KERNEL_CHECK();
    }
    if (is_dict) {           // DictionaryArray::try_new(codes, the input's dictionary)
      dfgpu_array* d = nullptr; dfgpu_status st2 = dfgpu_array_make_dictionary(ctx, ok.get(), key->dictionary, &d);
      if (st2 != DFGPU_OK) fail(st2, "%s", ctx->err.c_str());
      out_keys[0] = d;
    } else if (packed_keys) {  // the packed keys of the partial rows back into their columns (NULL digits -> validity)
      std::vector<ArrayHolder> kc((size_t)nkeys);
      for (int c = 0; c < nkeys; c++) {
        kc[(size_t)c].a = new_fixed(ctx, keys[c]->type, m, 0, 0, pc.nullable[c] != 0);
        uint64_t* vb = pc.nullable[c] ? (uint64_t*)kc[(size_t)c].get()->validity->ptr : nullptr; void* dst = kc[(size_t)c].get()->values->ptr;
#define PA_UNPACK(T) hipLaunchKernelGGL((k_pa_unpack<T>), dim3(grid_for(((m + 63) / 64) * 64, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)ok.get()->values->ptr, m, pc.stride[c], pc.range[c], pc.mn[c], pc.nullable[c], (T*)dst, vb)
        if (m) switch (type_width(keys[c]->type)) { case 1: PA_UNPACK(uint8_t); break; case 2: PA_UNPACK(uint16_t); break; case 4: PA_UNPACK(uint32_t); break; default: PA_UNPACK(uint64_t); break; }
#undef PA_UNPACK
        kc[(size_t)c].get()->null_count = pc.nullable[c] ? -1 : 0;
      }
      KERNEL_CHECK();
      for (int c = 0; c < nkeys; c++) out_keys[c] = kc[(size_t)c].release();
    } else out_keys[0] = ok.release();
    for (int i = 0; i < n_aggs; i++) { out_states[2 * i] = st[(size_t)2 * i].release(); out_states[2 * i + 1] = st[(size_t)2 * i + 1].release(); }
  });
}
