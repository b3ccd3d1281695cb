// groups.hip -- a8: GroupValues (key interning) for AggregateExec on gfx950.
//
// Reference: physical-plan/src/aggregates/group_values/{mod,primitive,row,bytes}.rs.  One device
// implementation serves all three reference variants: keys are compared column-wise in place (no
// row-format copy as GroupValuesRows does, row.rs:97), strings by (length, bytes).
//
// Persistent table in HBM: open addressing, 8-byte slots (hash tag:32 | payload:32).  payload is a
// group id, or NEW|batch-row while a key first seen in the current batch is still unnumbered.
// Ids must come out in FIRST-SEEN order (primitive.rs:137-141): every new slot keeps
// atomicMin(first row); rows that are "first of their group" are flagged in a bitmap, compacted in
// row order (ballot + scan), and their rank is the new id -- O(rows) streaming work, no sort.
#include <new>
#include "device_utils.h"
#include <algorithm>

namespace dfgpu {
constexpr uint64_t G_EMPTY = ~0ull;
constexpr uint32_t G_NEW = 0x80000000u;
constexpr uint32_t G_NONE = 0xFFFFFFFFu;
constexpr uint64_t GROUP_SEED = 0x5851f42d4c957f2dULL;
}
using namespace dfgpu;

struct dfgpu_groups {
  dfgpu_ctx* ctx = nullptr; int32_t nkeys = 0;
  int64_t n_groups = 0;
  std::vector<dfgpu_array*> keys;     // stored key columns, one row per group (null until first batch)
  uint64_t capacity = 0;
  int64_t size_hint = 0;               // upper bound / expectation of the number of groups when the caller knows one (0 = none)
  BufferPtr slots, first_row;          // u64[capacity], u32[capacity]
  BufferPtr ghash; int64_t ghash_cap = 0;   // u64 per group
  // run mode: every batch so far arrived with its keys clustered (first key column non-decreasing, the other columns constant within
  // equal first keys), so group ids are run numbers and no hash table exists yet; ghash is filled only if a later batch breaks the order
  bool run_mode = false;
  // canon mode: dictionary key columns are interned through a de-duplicated dictionary -- canon[code] = id of the distinct dictionary
  // VALUE (so codes with equal values stay one group) -- and rows are hashed / compared as u32 tuples instead of strings.  Valid while
  // every batch brings the same dictionary arrays; another dictionary drops back to value keys (stored groups are re-hashed).
  bool canon_mode = false;
  struct Canon { dfgpu_array* dict = nullptr; BufferPtr ids; int64_t n_ids = 0; };
  std::vector<Canon> canon;               // per key column (dict == null: not a dictionary column)
  std::vector<dfgpu_array*> canon_keys;   // per key column: u32 canon id per group (dictionary columns) or null (use keys[c])
  // dense canon mode: every key column is a dictionary column and the product of the canonical domains is <= 4096 (TPC-H Q1: 4 x 3):
  // the composite canonical id indexes dense_map (-> group id or none) directly -- no hashing, no table, two streaming passes
  BufferPtr dense_map; int64_t dense_size = 0; std::vector<uint32_t> dense_host;
  // direct map: ONE dictionary key column with a larger canonical domain: dmap[canonical id] = group id (or none) replaces the hash table
  // altogether -- the ids are dense in [0, n_ids], so "find or insert" is an array access
  BufferPtr dmap; int64_t dmap_size = 0;
  // primitive-key table (≙ GroupValuesPrimitive, group_values/primitive.rs): ONE 8-byte integer key column without NULLs.  A slot is
  // 16 bytes {key, group id, first row}: find-or-insert touches one sector per row (the general table needs the slot, its first-row
  // word and the representative key behind a matching tag: three), and no stored-key column is consulted.
  BufferPtr pslots; uint64_t pcap = 0; bool prim_mode = false, prim_banned = false;
  // run mode, first batch: the stored keys are take(lazy_src[c], lazy_rows) -- the batch's key columns at the first row of every run -- and are not gathered until somebody
  // needs them here (a second batch, emit).  dfgpu_groups_emit_deferred hands (columns, rows) to a caller that gathers lazily itself: TPC-H Q18's HAVING keeps a few thousand
  // of 150 M groups, and only their keys are ever read.
  std::vector<dfgpu_array*> lazy_src; dfgpu_array* lazy_rows = nullptr;
  ~dfgpu_groups() { for (auto* a : lazy_src) if (a) dfgpu_array_release(a); if (lazy_rows) dfgpu_array_release(lazy_rows); for (auto* a : keys) if (a) dfgpu_array_release(a); for (auto* a : canon_keys) if (a) dfgpu_array_release(a); for (auto& c : canon) if (c.dict) dfgpu_array_release(c.dict); }
};

namespace dfgpu {

__global__ void __launch_bounds__(BLOCK) k_groups_find(KeySet bk, KeySet stored, int has_stored, int64_t n, const uint64_t* mask, int force_zero,
                                                       uint64_t* slots, uint64_t cap_mask, uint32_t* first_row, uint32_t* tmp,
                                                       unsigned long long* counters /* [1] overflow */, uint64_t max_steps) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t res = G_NONE;
  if (mask == nullptr || bit_get(mask, i)) {
    bool an; uint64_t h = keyset_hash(bk, i, GROUP_SEED, &an);
    if (force_zero) h = 0;
    uint64_t tag = h >> 32, s = h & cap_mask, mine = (tag << 32) | (uint64_t)(G_NEW | (uint32_t)i);
    bool done = false;
    // No single-address group counter (1e6 atomics on one word cost 10+ ms): the number of new groups is the
    // popcount of the first-row bitmap computed afterwards.  A table that fills up shows as a probe sequence longer
    // than max_steps; the flag is polled so that the remaining workgroups drain quickly before the host retries.
    for (uint64_t step = 0; step <= max_steps && !done; step++) {
      if ((step & 15) == 0 && __hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;       // also before the first step: rows that start after an overflow leave at once
      uint64_t cur = slots[s];
      if (cur == G_EMPTY) {
        cur = atomicCAS((unsigned long long*)&slots[s], (unsigned long long)G_EMPTY, (unsigned long long)mine);
        if (cur == G_EMPTY) cur = mine;
      }
      if ((cur >> 32) == tag) {
        uint32_t pl = (uint32_t)cur;
        if (pl & G_NEW) {
          int64_t rep = pl & ~G_NEW;
          if (rep == i || keyset_equal(bk, i, bk, rep, true)) {
            // first_row only ever decreases, so a (possibly stale) read that is already <= i proves the atomic a no-op: with few or
            // skewed groups nearly every row skips it (60 M rows into 6 groups: 110 ms of serialised atomics -> 3 ms)
            if (__hip_atomic_load(&first_row[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (uint32_t)i) atomicMin(&first_row[s], (uint32_t)i);
            res = G_NEW | (uint32_t)s; done = true;
          }
        } else if (has_stored && keyset_equal(bk, i, stored, (int64_t)pl, true)) { res = pl; done = true; }
      }
      s = (s + 1) & cap_mask;
    }
    if (!done) counters[1] = 1ull;      // table (nearly) full: host grows it and redoes the batch
  }
  tmp[i] = res;
}
// ---------------------------------------------------------------- primitive-key table
struct alignas(16) PSlot { unsigned long long key; uint32_t gid; uint32_t first; };     // all-ones = empty key / no group id yet / no first row
constexpr unsigned long long P_EMPTY = ~0ull;
constexpr int PF_ROWS = 4;
// tmp[i] = slot of row i's key (G_NONE for masked rows).  New keys are claimed by a CAS on the key word; every row of a key without a
// group id lowers the slot's first-row word (read first: it only moves down).  counters[1] = table too full, counters[3] = a key equals
// the empty marker (the caller leaves this table for the general one).
// KT: unsigned long long (8-byte keys) or uint32_t (4-byte keys, zero-extended: the bit pattern is the key, so they can never equal the marker)
template <typename KT, bool HAS_MASK>
__global__ void __launch_bounds__(BLOCK) k_prim_find(const KT* keys, int64_t n, const uint64_t* mask, PSlot* slots, uint64_t cap_mask, uint32_t* tmp,
                                                     unsigned long long* counters, uint64_t max_steps) {
  const int64_t base = (int64_t)blockIdx.x * BLOCK * PF_ROWS + threadIdx.x;
  unsigned long long k[PF_ROWS]; uint64_t sl[PF_ROWS]; PSlot cur[PF_ROWS]; bool on[PF_ROWS];
#pragma unroll
  for (int q = 0; q < PF_ROWS; q++) { int64_t i = base + (int64_t)q * BLOCK; on[q] = i < n && (!HAS_MASK || bit_get(mask, i)); k[q] = (unsigned long long)keys[i < n ? i : n - 1]; }
#pragma unroll
  for (int q = 0; q < PF_ROWS; q++) sl[q] = mix64(k[q] ^ GROUP_SEED) & cap_mask;
#pragma unroll
  for (int q = 0; q < PF_ROWS; q++) cur[q] = slots[sl[q]];                  // one 16-byte load per row, all rows in flight
#pragma unroll
  for (int q = 0; q < PF_ROWS; q++) {
    const int64_t i = base + (int64_t)q * BLOCK;
    if (i >= n) continue;
    uint32_t res = G_NONE;
    if (on[q] && k[q] == P_EMPTY) counters[3] = 1ull;
    else if (on[q]) {
      uint64_t s = sl[q]; unsigned long long c = cur[q].key; uint32_t gid = cur[q].gid, first = cur[q].first; bool done = false;
      for (uint64_t step = 0; step <= max_steps; step++) {
        if ((step & 15) == 0 && __hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;       // the table overflowed somewhere: drain (also before the first step)
        if (step) { PSlot t = slots[s]; c = t.key; gid = t.gid; first = t.first; }
        if (c == P_EMPTY) { c = atomicCAS(&slots[s].key, P_EMPTY, k[q]); if (c == P_EMPTY) c = k[q]; gid = 0xFFFFFFFFu; first = 0xFFFFFFFFu; }
        if (c == k[q]) { done = true; break; }
        s = (s + 1) & cap_mask;
      }
      if (!done) counters[1] = 1ull;
      else { res = (uint32_t)s; if (gid == 0xFFFFFFFFu && first > (uint32_t)i) atomicMin(&slots[s].first, (uint32_t)i); }
    }
    tmp[i] = res;
  }
}
// slots claimed in this batch: key set, no group id yet
__global__ void __launch_bounds__(BLOCK) k_prim_new_bits(const PSlot* slots, int64_t cap, uint64_t* bits) {
  int64_t s = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  bool f = false;
  if (s < cap) { PSlot t = slots[s]; f = t.key != P_EMPTY && t.gid == 0xFFFFFFFFu && t.first != 0xFFFFFFFFu; }
  uint64_t m = ballot64(f);
  if (lane_id() == 0 && (s >> 6) < ((cap + 63) >> 6)) bits[s >> 6] = m;
}
__global__ void __launch_bounds__(BLOCK) k_prim_first_of(const uint32_t* new_slots, int64_t n_new, const PSlot* slots, uint32_t* first) {
  int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (r < n_new) first[r] = slots[new_slots[r]].first;
}
__global__ void __launch_bounds__(BLOCK) k_prim_assign(const uint32_t* new_slots, int64_t n_new, uint32_t base, PSlot* slots) {
  int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (r < n_new) slots[new_slots[r]].gid = base + (uint32_t)r;
}
__global__ void __launch_bounds__(BLOCK) k_prim_ids(const uint32_t* tmp, int64_t n, const PSlot* slots, uint32_t* out) {
  const int64_t base = (int64_t)blockIdx.x * BLOCK * PF_ROWS + threadIdx.x;
  uint32_t t[PF_ROWS], g[PF_ROWS];
#pragma unroll
  for (int q = 0; q < PF_ROWS; q++) { int64_t i = base + (int64_t)q * BLOCK; t[q] = tmp[i < n ? i : n - 1]; }
#pragma unroll
  for (int q = 0; q < PF_ROWS; q++) g[q] = t[q] == G_NONE ? G_NONE : slots[t[q]].gid;
#pragma unroll
  for (int q = 0; q < PF_ROWS; q++) { int64_t i = base + (int64_t)q * BLOCK; if (i < n) out[i] = g[q]; }
}
// (re)build: the numbered groups' keys into an empty table (keys are distinct)
template <typename KT>
__global__ void __launch_bounds__(BLOCK) k_prim_insert(const KT* gkeys, int64_t n_groups, PSlot* slots, uint64_t cap_mask) {
  int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (g >= n_groups) return;
  unsigned long long k = (unsigned long long)gkeys[g]; uint64_t s = mix64(k ^ GROUP_SEED) & cap_mask;
  for (;;) { unsigned long long c = atomicCAS(&slots[s].key, P_EMPTY, k); if (c == P_EMPTY) break; s = (s + 1) & cap_mask; }
  slots[s].gid = (uint32_t)g; slots[s].first = 0u;
}

__global__ void __launch_bounds__(BLOCK) k_groups_mark_first(const uint32_t* tmp, const uint32_t* first_row, int64_t n, uint64_t* bits) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  bool f = false;
  if (i < n) { uint32_t t = tmp[i]; f = t != G_NONE && (t & G_NEW) && first_row[t & ~G_NEW] == (uint32_t)i; }
  uint64_t m = ballot64(f);
  if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) bits[i >> 6] = m;
}
__global__ void __launch_bounds__(BLOCK) k_groups_assign(KeySet bk, const uint32_t* first_rows, int64_t n_new, const uint32_t* tmp, uint64_t* slots,
                                                         uint32_t* first_row, uint32_t base, int force_zero, uint64_t* ghash) {
  int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (k >= n_new) return;
  uint32_t i = first_rows[k], s = tmp[i] & ~G_NEW;
  slots[s] = (slots[s] & 0xFFFFFFFF00000000ull) | (uint64_t)(base + (uint32_t)k);
  first_row[s] = G_NONE;
  bool an; uint64_t h = keyset_hash(bk, i, GROUP_SEED, &an);
  ghash[base + k] = force_zero ? 0 : h;
}
__global__ void __launch_bounds__(BLOCK) k_groups_finalize(const uint32_t* tmp, const uint64_t* slots, int64_t n, uint32_t* out) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t t = tmp[i];
  out[i] = (t != G_NONE && (t & G_NEW)) ? (uint32_t)slots[t & ~G_NEW] : t;
}
// re-insert numbered groups into a fresh table (all keys distinct: no comparisons needed)
__global__ void __launch_bounds__(BLOCK) k_groups_rehash(const uint64_t* ghash, int64_t n_groups, uint64_t* slots, uint64_t cap_mask) {
  int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (g >= n_groups) return;
  uint64_t h = ghash[g], s = h & cap_mask, mine = ((h >> 32) << 32) | (uint64_t)g;
  for (uint64_t step = 0; step <= cap_mask; step++) {
    if (slots[s] == G_EMPTY && atomicCAS((unsigned long long*)&slots[s], (unsigned long long)G_EMPTY, (unsigned long long)mine) == G_EMPTY) return;
    s = (s + 1) & cap_mask;
  }
}

// ---- clustered keys (≙ GroupOrdering::Full, aggregates/order/full.rs: input sorted on the group keys): groups are runs.
// bad = the batch is not clustered; heads bit i = row i starts a new run.  `prev` (device, 1 value, may be null) is the
// first-column key of the last group of earlier batches: the batch must start strictly above it.
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_run_heads(KeySet bk, int64_t n, const T* prev, uint64_t* heads, unsigned long long* bad) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  bool head = false, wrong = false;
  if (i < n) {
    const T* k0 = (const T*)bk.c[0].values;
    if (i == 0) { head = true; wrong = prev != nullptr && !(prev[0] < k0[0]); }
    else {
      T a = k0[i - 1], b = k0[i];
      if (b < a) wrong = true;
      else if (a < b) head = true;
      else for (int c = 1; c < bk.n && !wrong; c++) {          // equal first key: every other key column must repeat too (NULL == NULL)
        int64_t ri, rj; bool va = cell_resolve(bk.c[c], i, &ri), vb = cell_resolve(bk.c[c], i - 1, &rj);
        if (va != vb || (va && !cell_equal(bk.c[c], ri, bk.c[c], rj))) wrong = true;
      }
    }
  }
  uint64_t m = ballot64(head);
  if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) heads[i >> 6] = m;
  if (ballot64(wrong) && lane_id() == 0) *bad = 1ull;
}
// single key column: 4 slabs of 64 rows per wave, every key loaded once and unconditionally (the neighbour comes from the lane below)
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_run_heads1(const T* k0, int64_t n, const T* prev, uint64_t* heads, unsigned long long* bad) {
  constexpr int SLABS = 4;
  const int lane = lane_id();
  const int64_t base = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * (WAVE * SLABS);
  if (base >= n) return;                                             // wave-uniform
  T k[SLABS];
#pragma unroll
  for (int r = 0; r < SLABS; r++) { int64_t i = base + r * WAVE + lane; k[r] = k0[i < n ? i : n - 1]; }
  T edge = base > 0 ? k0[base - 1] : (T)0;                           // the key before this wave's rows
  bool wrong = false;
#pragma unroll
  for (int r = 0; r < SLABS; r++) {
    int64_t i = base + r * WAVE + lane;
    T p = (T)__shfl_up((long long)k[r], 1, 64);
    if (lane == 0) p = edge;
    edge = (T)__shfl((long long)k[r], 63, 64);
    bool in = i < n;
    bool head = in && (i == 0 || p < k[r]);
    wrong |= in && (i == 0 ? (prev != nullptr && !(prev[0] < k[r])) : (k[r] < p));
    uint64_t m = ballot64(head);
    if (lane == 0 && base + r * WAVE < n) heads[(base >> 6) + r] = m;
  }
  if (ballot64(wrong) && lane == 0) *bad = 1ull;
}
__global__ void __launch_bounds__(BLOCK) k_popc_words_g(const uint64_t* words, int64_t nw, uint32_t* out) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < nw) out[i] = (uint32_t)__popcll(words[i]);
}
__global__ void __launch_bounds__(BLOCK) k_run_ids(const uint64_t* heads, const uint32_t* word_prefix, int64_t n, uint32_t base, uint32_t* out) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint64_t w = heads[i >> 6]; int b = (int)(i & 63);
  out[i] = base + word_prefix[i >> 6] + (uint32_t)__popcll(w & ((b == 63) ? ~0ull : ((2ull << b) - 1ull))) - 1u;      // heads at or before i, minus one
}
__global__ void __launch_bounds__(BLOCK) k_groups_hash_stored(KeySet stored, int64_t n_groups, int force_zero, uint64_t* ghash) {
  int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (g >= n_groups) return;
  bool an; uint64_t h = keyset_hash(stored, g, GROUP_SEED, &an);
  ghash[g] = force_zero ? 0 : h;
}

// rows of a dictionary key column -> canonical id of the value (NULL -> n_ids, one extra id)
__global__ void __launch_bounds__(BLOCK) k_canon_lookup(const void* keys, int key_type, const uint64_t* key_valid, int64_t n, const uint32_t* canon, const uint64_t* dict_valid, int64_t dict_len, uint32_t n_ids, uint32_t* out) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  int64_t c = valid_at(key_valid, i) ? key_at(keys, key_type, i) : -1;
  out[i] = (c >= 0 && c < dict_len && valid_at(dict_valid, c)) ? canon[c] : n_ids;          // a NULL code and a NULL dictionary value are the same NULL key
}

constexpr int DENSE_MAX = 4096;
constexpr int DENSE_ROWS = 8;
__device__ inline uint32_t dense_composite(const DenseCols& dc, int64_t i) {
  uint32_t comp = 0;
#pragma unroll
  for (int c = 0; c < MAX_KEYS; c++) {              // constant indices: a runtime-indexed kernel-argument array would live in scratch
    if (c >= dc.n) break;
    const DenseCol& d = dc.c[c];
    int64_t code = valid_at(d.key_valid, i) ? key_at(d.keys, d.key_type, i) : -1;
    uint32_t id = (code >= 0 && code < d.dict_len && valid_at(d.dict_valid, code)) ? d.canon[code] : d.n_ids;
    comp += id * d.stride;
  }
  return comp;
}
// straight-line variant for the common shape (all code columns of one integer type K, no NULL codes, no NULL dictionary values):
// the generic dense_composite branches per column (validity pointers, code type), which puts every load behind its own
// s_waitcnt; here the loads of a lane's rows are unconditional and overlap
template <typename K, int NC>
__device__ inline uint32_t dense_composite_fast(const DenseCols& dc, int64_t i) {
  uint32_t comp = 0;
#pragma unroll
  for (int c = 0; c < NC; c++) comp += dc.c[c].canon[((const K*)dc.c[c].keys)[i]] * dc.c[c].stride;
  return comp;
}
template <typename K, int NC, bool HAS_MASK>
__global__ void __launch_bounds__(BLOCK) k_dense_first_fast(DenseCols dc, int64_t n, const uint64_t* mask, const uint32_t* dense_map, uint32_t* first, int dsize) {
  __shared__ uint32_t lfirst[DENSE_MAX];
  for (int t = threadIdx.x; t < dsize; t += BLOCK) lfirst[t] = dense_map[t] == G_NONE ? G_NONE : 0u;
  __syncthreads();
  for (int64_t base = (int64_t)blockIdx.x * BLOCK * DENSE_ROWS; base < n; base += (int64_t)gridDim.x * BLOCK * DENSE_ROWS) {
    if (base + (int64_t)BLOCK * DENSE_ROWS <= n) {
      uint32_t comp[DENSE_ROWS];
#pragma unroll
      for (int r = 0; r < DENSE_ROWS; r++) comp[r] = dense_composite_fast<K, NC>(dc, base + (int64_t)r * BLOCK + threadIdx.x);
#pragma unroll
      for (int r = 0; r < DENSE_ROWS; r++) { int64_t i64 = base + (int64_t)r * BLOCK + threadIdx.x; uint32_t i = (uint32_t)i64; if ((!HAS_MASK || bit_get(mask, i64)) && lfirst[comp[r]] > i) atomicMin(&lfirst[comp[r]], i); }
    } else {
      for (int r = 0; r < DENSE_ROWS; r++) { int64_t i = base + (int64_t)r * BLOCK + threadIdx.x; if (i < n && (!HAS_MASK || bit_get(mask, i))) { uint32_t c = dense_composite_fast<K, NC>(dc, i); if (lfirst[c] > (uint32_t)i) atomicMin(&lfirst[c], (uint32_t)i); } }
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < dsize; t += BLOCK) if (dense_map[t] == G_NONE && lfirst[t] != G_NONE) atomicMin(&first[t], lfirst[t]);
}
// Same pass with the code tuple -> composite table composed in LDS per workgroup (the two dependent global lookups of
// dense_composite_fast become one LDS read) and 4 consecutive rows per lane (one load per code column and 4 rows).  Host checks:
// len0 * len1 <= DENSE_MAX and 16-byte aligned code columns.
template <typename K, int NC, bool HAS_MASK>
__global__ void __launch_bounds__(BLOCK) k_dense_first_tab(DenseCols dc, int64_t n, const uint64_t* mask, const uint32_t* dense_map, uint32_t* first, int dsize, int len1) {
  __shared__ uint32_t lfirst[DENSE_MAX]; __shared__ uint16_t ctab[DENSE_MAX];
  const int tsize = (int)dc.c[0].dict_len * len1;
  for (int t = threadIdx.x; t < dsize; t += BLOCK) lfirst[t] = dense_map[t] == G_NONE ? G_NONE : 0u;
  for (int t = threadIdx.x; t < tsize; t += BLOCK) ctab[t] = (uint16_t)(dc.c[0].canon[t / len1] * dc.c[0].stride + (NC == 2 ? dc.c[1].canon[t % len1] * dc.c[1].stride : 0u));
  __syncthreads();
  struct alignas(sizeof(K) * 4) K4 { K v[4]; };
  const int64_t nvec = n / 4;
  for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * BLOCK) {
    const int64_t i0 = v * 4;
    K4 a = *(const K4*)((const K*)dc.c[0].keys + i0); K4 b{}; if (NC == 2) b = *(const K4*)((const K*)dc.c[1].keys + i0);
    uint64_t m = HAS_MASK ? mask[i0 >> 6] >> (i0 & 63) : 0xFull;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      uint32_t comp = ctab[(uint32_t)a.v[r] * (uint32_t)len1 + (NC == 2 ? (uint32_t)b.v[r] : 0u)], i = (uint32_t)(i0 + r);
      if (((m >> r) & 1ull) && lfirst[comp] > i) atomicMin(&lfirst[comp], i);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {           // ragged tail
    int64_t i = nvec * 4 + threadIdx.x;
    if (!HAS_MASK || bit_get(mask, i)) { uint32_t c = dense_composite_fast<K, NC>(dc, i); if (lfirst[c] > (uint32_t)i) atomicMin(&lfirst[c], (uint32_t)i); }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < dsize; t += BLOCK) if (dense_map[t] == G_NONE && lfirst[t] != G_NONE) atomicMin(&first[t], lfirst[t]);
}
template <typename K, int NC, bool HAS_MASK>
__global__ void __launch_bounds__(BLOCK) k_dense_ids_fast(DenseCols dc, int64_t n, const uint64_t* mask, const uint32_t* dense_map, uint32_t* out) {
  int64_t base = (int64_t)blockIdx.x * BLOCK * DENSE_ROWS;
  if (base + (int64_t)BLOCK * DENSE_ROWS <= n) {
    uint32_t g[DENSE_ROWS];
#pragma unroll
    for (int r = 0; r < DENSE_ROWS; r++) g[r] = dense_map[dense_composite_fast<K, NC>(dc, base + (int64_t)r * BLOCK + threadIdx.x)];
#pragma unroll
    for (int r = 0; r < DENSE_ROWS; r++) { int64_t i = base + (int64_t)r * BLOCK + threadIdx.x; out[i] = (!HAS_MASK || bit_get(mask, i)) ? g[r] : G_NONE; }
  } else {
    for (int r = 0; r < DENSE_ROWS; r++) { int64_t i = base + (int64_t)r * BLOCK + threadIdx.x; if (i < n) out[i] = (!HAS_MASK || bit_get(mask, i)) ? dense_map[dense_composite_fast<K, NC>(dc, i)] : G_NONE; }
  }
}
// pass A: first row of every composite that has no group id yet.  Workgroups stride over the rows keeping their minima in LDS (a
// single global line hammered by every wave serialises in its L2 channel), one global atomicMin per workgroup and composite at the
// end.  8 rows per lane and iteration: code -> canonical id -> map is a chain of dependent loads, so the rows' chains must overlap.
__global__ void __launch_bounds__(BLOCK) k_dense_first(DenseCols dc, int64_t n, const uint64_t* mask, const uint32_t* dense_map, uint32_t* first, int dsize) {
  __shared__ uint32_t lfirst[DENSE_MAX];
  for (int t = threadIdx.x; t < dsize; t += BLOCK) lfirst[t] = dense_map[t] == G_NONE ? G_NONE : 0u;       // 0: already numbered, nothing can lower it
  __syncthreads();
  for (int64_t base = (int64_t)blockIdx.x * BLOCK * DENSE_ROWS; base < n; base += (int64_t)gridDim.x * BLOCK * DENSE_ROWS) {
    uint32_t comp[DENSE_ROWS]; bool on[DENSE_ROWS];
#pragma unroll
    for (int r = 0; r < DENSE_ROWS; r++) { int64_t i = base + (int64_t)r * BLOCK + threadIdx.x; on[r] = i < n && (mask == nullptr || bit_get(mask, i)); comp[r] = on[r] ? dense_composite(dc, i) : 0; }
#pragma unroll
    for (int r = 0; r < DENSE_ROWS; r++) { uint32_t i = (uint32_t)(base + (int64_t)r * BLOCK + threadIdx.x); if (on[r] && lfirst[comp[r]] > i) atomicMin(&lfirst[comp[r]], i); }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < dsize; t += BLOCK) if (dense_map[t] == G_NONE && lfirst[t] != G_NONE) atomicMin(&first[t], lfirst[t]);
}
// pass B: group id of every row
__global__ void __launch_bounds__(BLOCK) k_dense_ids(DenseCols dc, int64_t n, const uint64_t* mask, const uint32_t* dense_map, uint32_t* out) {
  int64_t base = (int64_t)blockIdx.x * BLOCK * DENSE_ROWS;
  uint32_t g[DENSE_ROWS];
#pragma unroll
  for (int r = 0; r < DENSE_ROWS; r++) { int64_t i = base + (int64_t)r * BLOCK + threadIdx.x; g[r] = (i < n && (mask == nullptr || bit_get(mask, i))) ? dense_map[dense_composite(dc, i)] : G_NONE; }
#pragma unroll
  for (int r = 0; r < DENSE_ROWS; r++) { int64_t i = base + (int64_t)r * BLOCK + threadIdx.x; if (i < n) out[i] = g[r]; }
}

// ---- direct map over canonical ids (one dictionary key column).  4 rows per lane, loads clamped and unconditional.
constexpr int DM_ROWS = 4;
// pass 1: first row of every canonical id that has no group yet (read before the atomicMin: it only ever moves down)
__global__ void __launch_bounds__(BLOCK) k_dm_first(const uint32_t* cid, int64_t n, const uint64_t* mask, const uint32_t* dmap, uint32_t* first) {
  const int64_t base = (int64_t)blockIdx.x * BLOCK * DM_ROWS + threadIdx.x;
  uint32_t id[DM_ROWS], gm[DM_ROWS], fr[DM_ROWS]; bool on[DM_ROWS];
#pragma unroll
  for (int q = 0; q < DM_ROWS; q++) { int64_t i = base + (int64_t)q * BLOCK; on[q] = i < n && (mask == nullptr || bit_get(mask, i)); id[q] = cid[i < n ? i : n - 1]; }
#pragma unroll
  for (int q = 0; q < DM_ROWS; q++) { gm[q] = dmap[id[q]]; fr[q] = first[id[q]]; }
#pragma unroll
  for (int q = 0; q < DM_ROWS; q++) { uint32_t i = (uint32_t)(base + (int64_t)q * BLOCK); if (on[q] && gm[q] == G_NONE && fr[q] > i) atomicMin(&first[id[q]], i); }
}
// pass 2: head bit of row i = it is the first row of a canonical id without a group (one word per wave slab)
__global__ void __launch_bounds__(BLOCK) k_dm_heads(const uint32_t* cid, int64_t n, const uint64_t* mask, const uint32_t* dmap, const uint32_t* first, uint64_t* heads) {
  const int64_t base = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * (WAVE * DM_ROWS);
  if (base >= n) return;
  const int lane = lane_id();
  uint32_t id[DM_ROWS], gm[DM_ROWS], fr[DM_ROWS]; bool on[DM_ROWS];
#pragma unroll
  for (int q = 0; q < DM_ROWS; q++) { int64_t i = base + q * WAVE + lane; on[q] = i < n && (mask == nullptr || bit_get(mask, i)); id[q] = cid[i < n ? i : n - 1]; }
#pragma unroll
  for (int q = 0; q < DM_ROWS; q++) { gm[q] = dmap[id[q]]; fr[q] = first[id[q]]; }
#pragma unroll
  for (int q = 0; q < DM_ROWS; q++) {
    int64_t i = base + q * WAVE + lane;
    uint64_t m = ballot64(on[q] && gm[q] == G_NONE && fr[q] == (uint32_t)i);
    if (lane == 0 && base + q * WAVE < n) heads[(base >> 6) + q] = m;
  }
}
// the r-th new group (first rows ascending = first-seen order) takes id base + r
__global__ void __launch_bounds__(BLOCK) k_dm_assign(const uint32_t* firsts, int64_t n_new, const uint32_t* cid, uint32_t base, uint32_t* dmap, uint32_t* new_cid) {
  int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (r >= n_new) return;
  uint32_t id = cid[firsts[r]]; dmap[id] = base + (uint32_t)r; new_cid[r] = id;
}
__global__ void __launch_bounds__(BLOCK) k_dm_ids(const uint32_t* cid, int64_t n, const uint64_t* mask, const uint32_t* dmap, uint32_t* out) {
  const int64_t base = (int64_t)blockIdx.x * BLOCK * DM_ROWS + threadIdx.x;
  uint32_t id[DM_ROWS], g[DM_ROWS];
#pragma unroll
  for (int q = 0; q < DM_ROWS; q++) { int64_t i = base + (int64_t)q * BLOCK; id[q] = cid[i < n ? i : n - 1]; }
#pragma unroll
  for (int q = 0; q < DM_ROWS; q++) g[q] = dmap[id[q]];
#pragma unroll
  for (int q = 0; q < DM_ROWS; q++) { int64_t i = base + (int64_t)q * BLOCK; if (i < n) out[i] = (mask == nullptr || bit_get(mask, i)) ? g[q] : G_NONE; }
}
__global__ void __launch_bounds__(BLOCK) k_dm_seed(const uint32_t* group_cid, int64_t n_groups, uint32_t* dmap) {       // groups numbered by earlier (hash-path) batches
  int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (g < n_groups) dmap[group_cid[g]] = (uint32_t)g;
}

void materialize_ids(dfgpu_ctx* ctx, const dfgpu_array* ids_c) {
  if (!ids_c || !ids_c->deferred_ids) return;
  dfgpu_array* ids = const_cast<dfgpu_array*>(ids_c); std::shared_ptr<DeferredIds> d = ids->deferred_ids;
  int64_t n = ids->length; const uint64_t* mk = d->mask ? (const uint64_t*)d->mask->ptr : nullptr;
  if (d->kind == 1) {
    KernelTimer kt_(ctx, "k_groups_runs");
    hipLaunchKernelGGL(k_run_ids, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)d->heads->ptr, (const uint32_t*)d->prefix->ptr, n, d->base, (uint32_t*)ids->values->ptr);
    KERNEL_CHECK(); ids->deferred_ids.reset(); return;
  }
  dim3 grid(grid_for(n, BLOCK * DENSE_ROWS)), block(BLOCK);
  KernelTimer kt_(ctx, "k_groups_dense");
#define IDS(K, NC) do { if (mk) hipLaunchKernelGGL((k_dense_ids_fast<K, NC, true>), grid, block, 0, ctx->stream, d->dc, n, mk, (const uint32_t*)d->dense_map->ptr, (uint32_t*)ids->values->ptr); \
                        else hipLaunchKernelGGL((k_dense_ids_fast<K, NC, false>), grid, block, 0, ctx->stream, d->dc, n, mk, (const uint32_t*)d->dense_map->ptr, (uint32_t*)ids->values->ptr); } while (0)
  if (d->key_type == DFGPU_INT8) { if (d->dc.n == 1) IDS(int8_t, 1); else IDS(int8_t, 2); }
  else if (d->key_type == DFGPU_INT16) { if (d->dc.n == 1) IDS(int16_t, 1); else IDS(int16_t, 2); }
  else { if (d->dc.n == 1) IDS(int32_t, 1); else IDS(int32_t, 2); }
#undef IDS
  KERNEL_CHECK();
  ids->deferred_ids.reset();
}

static void groups_alloc_table(dfgpu_groups* g, uint64_t cap) {
  dfgpu_ctx* ctx = g->ctx;
  g->capacity = cap;
  g->slots = alloc_buffer(ctx, cap * 8); HIP_CHECK(hipMemsetAsync(g->slots->ptr, 0xFF, cap * 8, ctx->stream));
  g->first_row = alloc_buffer(ctx, cap * 4); HIP_CHECK(hipMemsetAsync(g->first_row->ptr, 0xFF, cap * 4, ctx->stream));
  if (g->n_groups) hipLaunchKernelGGL(k_groups_rehash, dim3(grid_for(g->n_groups, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)g->ghash->ptr, g->n_groups, (uint64_t*)g->slots->ptr, cap - 1);
  KERNEL_CHECK();
}

}  // namespace dfgpu

extern "C" {

dfgpu_status dfgpu_groups_new(dfgpu_ctx* ctx, int32_t nkeys, dfgpu_groups** out) {
  return guard(ctx, [&] {
    if (nkeys < 1 || nkeys > MAX_KEYS || !out) fail(DFGPU_INVALID_ARGUMENT, "groups_new: 1..%d key columns", MAX_KEYS);
    auto* g = new dfgpu_groups(); g->ctx = ctx; g->nkeys = nkeys; g->keys.assign(nkeys, nullptr);
    *out = g;
  });
}
void dfgpu_groups_free(dfgpu_groups* g) { delete g; }
int64_t dfgpu_groups_len(const dfgpu_groups* g) { return g ? g->n_groups : 0; }
int64_t dfgpu_groups_size(const dfgpu_groups* g) {
  if (!g) return 0;
  int64_t b = (int64_t)g->capacity * 12 + g->ghash_cap * 8 + (int64_t)g->pcap * 16 + g->dmap_size * 4;
  for (auto* a : g->keys) if (a) b += (a->values ? (int64_t)a->values->bytes : 0) + (a->validity ? (int64_t)a->validity->bytes : 0) + (a->offsets ? (int64_t)a->offsets->bytes : 0);
  return b;
}

// append the key values of the new groups (first-seen rows, in id order) to the stored key columns
static void groups_append_keys(dfgpu_ctx* ctx, dfgpu_groups* g, const dfgpu_array* const* cols, int32_t nkeys, const dfgpu_array* firsts, int64_t n_new) {
  for (int c = 0; c < nkeys; c++) {
    ArrayHolder nk(take_impl(ctx, cols[c], firsts->values->ptr, 4, nullptr, n_new));
    if (nk.get()->type == DFGPU_DICTIONARY) {       // store plain values (GroupValues emits the value type)
      dfgpu_array* k = nk.get(); int kw = type_width(k->key_type);
      if (kw != 4 && kw != 8) { dfgpu_array* wide = nullptr; ArrayHolder keys_only(new_array(ctx, k->key_type, k->length)); keys_only.get()->values = k->values; keys_only.get()->validity = k->validity; keys_only.get()->null_count = k->null_count;
        dfgpu_status st = dfgpu_cast(ctx, keys_only.get(), DFGPU_INT64, 0, 0, &wide); if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str());
        ArrayHolder w(wide); ArrayHolder dec(take_impl(ctx, k->dictionary, w.get()->values->ptr, 8, w.get()->validity ? (const uint64_t*)w.get()->validity->ptr : nullptr, k->length)); dfgpu_array_release(nk.release()); nk.a = dec.release(); }
      else { ArrayHolder dec(take_impl(ctx, k->dictionary, k->values->ptr, kw, k->validity ? (const uint64_t*)k->validity->ptr : nullptr, k->length)); dfgpu_array_release(nk.release()); nk.a = dec.release(); }
    }
    if (!g->keys[c]) g->keys[c] = nk.release();
    else { const dfgpu_array* parts[2] = { g->keys[c], nk.get() }; dfgpu_array* cat = nullptr; dfgpu_status st = dfgpu_concat(ctx, parts, 2, &cat); if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str());
           dfgpu_array_release(g->keys[c]); g->keys[c] = cat; }
  }
}
static void groups_resolve_keys(dfgpu_ctx* ctx, dfgpu_groups* g) {        // the pending gather of a first run-mode batch's keys, now
  if (!g->lazy_rows) return;
  std::vector<dfgpu_array*> src; src.swap(g->lazy_src); dfgpu_array* rows = g->lazy_rows; g->lazy_rows = nullptr;
  struct Drop { std::vector<dfgpu_array*>& s; dfgpu_array* r; ~Drop() { for (auto* a : s) if (a) dfgpu_array_release(a); dfgpu_array_release(r); } } drop{src, rows};
  std::vector<const dfgpu_array*> cs(src.begin(), src.end());
  groups_append_keys(ctx, g, cs.data(), g->nkeys, rows, rows->length);
}
static void groups_reserve_ghash(dfgpu_ctx* ctx, dfgpu_groups* g, int64_t need, int64_t keep) {
  if (need <= g->ghash_cap) return;
  int64_t nc = g->ghash_cap ? g->ghash_cap : 1024; while (nc < need) nc *= 2;
  BufferPtr nh = alloc_buffer(ctx, (size_t)nc * 8);
  if (keep && g->ghash) HIP_CHECK(hipMemcpyAsync(nh->ptr, g->ghash->ptr, (size_t)keep * 8, hipMemcpyDeviceToDevice, ctx->stream));
  g->ghash = nh; g->ghash_cap = nc;
}
static bool run_key_type(const dfgpu_array* a) {
  if (a->type == DFGPU_DICTIONARY || a->validity) return false;
  switch (a->type) { case DFGPU_INT8: case DFGPU_INT16: case DFGPU_INT32: case DFGPU_INT64: case DFGPU_DATE32: case DFGPU_UINT8: case DFGPU_UINT16: case DFGPU_UINT32: case DFGPU_UINT64: return true; default: return false; }
}
// Clustered batch -> ids by run number.  Returns false (nothing changed) when the batch is not clustered.
static bool groups_intern_runs(dfgpu_ctx* ctx, dfgpu_groups* g, const dfgpu_array* const* cols, int32_t nkeys, const KeySet& bk, int64_t n, dfgpu_array* ids, bool allow_deferred) {
  const dfgpu_array* k0 = cols[0];
  if (g->n_groups && logical_type(k0) != g->keys[0]->type) return false;
  KernelTimer kt_(ctx, "k_groups_runs");
  BufferPtr heads = alloc_buffer(ctx, bitmap_bytes(n));
  zero_scratch(ctx);
  int w = type_width(k0->type);
  const void* prev = g->n_groups ? (const void*)((const char*)g->keys[0]->values->ptr + (size_t)(g->n_groups - 1) * w) : nullptr;
  dim3 grid(grid_for(n, BLOCK)), block(BLOCK);
#define RUNS(T) do { if (nkeys == 1) hipLaunchKernelGGL((k_run_heads1<T>), dim3(grid_for(n, BLOCK * 4)), block, 0, ctx->stream, (const T*)k0->values->ptr, n, (const T*)prev, (uint64_t*)heads->ptr, (unsigned long long*)(ctx->d_scratch64 + 2)); \
                     else hipLaunchKernelGGL((k_run_heads<T>), grid, block, 0, ctx->stream, bk, n, (const T*)prev, (uint64_t*)heads->ptr, (unsigned long long*)(ctx->d_scratch64 + 2)); } while (0)
  switch (k0->type) {
    case DFGPU_INT8: RUNS(int8_t); break; case DFGPU_INT16: RUNS(int16_t); break; case DFGPU_INT32: case DFGPU_DATE32: RUNS(int32_t); break; case DFGPU_INT64: RUNS(int64_t); break;
    case DFGPU_UINT8: RUNS(uint8_t); break; case DFGPU_UINT16: RUNS(uint16_t); break; case DFGPU_UINT32: RUNS(uint32_t); break; default: RUNS(uint64_t); break; }
#undef RUNS
  KERNEL_CHECK();
  // "not clustered" and the number of runs in ONE read-back (counting the heads of an unclustered batch is the price of the lost bet: two small kernels)
  ArrayHolder firsts(mask_to_indices_checked(ctx, (const uint64_t*)heads->ptr, n, 2, nullptr));
  if (!firsts.get()) return false;
  int64_t n_new = firsts.get()->length;
  if (g->n_groups + n_new >= (int64_t)G_NEW) fail(DFGPU_RESOURCES_EXHAUSTED, "more than 2^31 groups");
  int64_t nw = (n + 63) / 64;
  BufferPtr prefix = alloc_buffer(ctx, (size_t)nw * 4);
  hipLaunchKernelGGL(k_popc_words_g, dim3(grid_for(nw, BLOCK)), block, 0, ctx->stream, (const uint64_t*)heads->ptr, nw, (uint32_t*)prefix->ptr);
  exclusive_scan_u32_inplace32(ctx, (uint32_t*)prefix->ptr, nw, nullptr);
  if (allow_deferred && n >= (1 << 20)) {      // the run number of a row is a popcount away from the head bits: the accumulate pass derives it
    auto d = std::make_shared<DeferredIds>(); d->kind = 1; d->heads = heads; d->prefix = prefix; d->base = (uint32_t)g->n_groups; ids->deferred_ids = d;
  } else hipLaunchKernelGGL(k_run_ids, grid, block, 0, ctx->stream, (const uint64_t*)heads->ptr, (const uint32_t*)prefix->ptr, n, (uint32_t)g->n_groups, (uint32_t*)ids->values->ptr);
  KERNEL_CHECK();
  if (n_new && g->n_groups == 0 && ctx->group_lazy_keys && n_new >= (1 << 16)) {        // first batch: the gather waits (groups_resolve_keys)
    for (int c = 0; c < nkeys; c++) { g->lazy_src.push_back(const_cast<dfgpu_array*>(cols[c])); dfgpu_array_retain(g->lazy_src.back()); }
    g->lazy_rows = firsts.release();
  } else if (n_new) groups_append_keys(ctx, g, cols, nkeys, firsts.get(), n_new);
  g->n_groups += n_new; g->run_mode = true;
  return true;
}

static dfgpu_status groups_intern_impl(dfgpu_ctx* ctx, dfgpu_groups* g, const dfgpu_array* const* cols, int32_t nkeys, const dfgpu_array* opt_mask, dfgpu_array** out_group_ids, bool allow_deferred);
dfgpu_status dfgpu_groups_intern(dfgpu_ctx* ctx, dfgpu_groups* g, const dfgpu_array* const* cols, int32_t nkeys, const dfgpu_array* opt_mask, dfgpu_array** out_group_ids) {
  return groups_intern_impl(ctx, g, cols, nkeys, opt_mask, out_group_ids, false);
}
dfgpu_status dfgpu_groups_intern_deferred(dfgpu_ctx* ctx, dfgpu_groups* g, const dfgpu_array* const* cols, int32_t nkeys, const dfgpu_array* opt_mask, dfgpu_array** out_group_ids) {
  return groups_intern_impl(ctx, g, cols, nkeys, opt_mask, out_group_ids, true);
}
static dfgpu_status groups_intern_impl(dfgpu_ctx* ctx, dfgpu_groups* g, const dfgpu_array* const* cols, int32_t nkeys, const dfgpu_array* opt_mask, dfgpu_array** out_group_ids, bool allow_deferred) {
  return guard(ctx, [&] {
    if (!g || !cols || !out_group_ids) fail(DFGPU_INVALID_ARGUMENT, "groups_intern: null argument");
    if (nkeys != g->nkeys) fail(DFGPU_INVALID_ARGUMENT, "groups_intern: %d key columns given, %d expected", nkeys, g->nkeys);
    groups_resolve_keys(ctx, g);
    KeySet bk = make_keyset(cols, nkeys);
    int64_t n = cols[0]->length;
    if (n >= (int64_t)G_NEW) fail(DFGPU_NOT_IMPLEMENTED, "intern batches above 2^31 rows; split the batch");
    for (int c = 0; c < nkeys; c++) if (g->keys[c] && logical_type(cols[c]) != g->keys[c]->type) fail(DFGPU_INVALID_ARGUMENT, "groups_intern: key %d changed type", c);
    BufferPtr mask = effective_mask(ctx, opt_mask, n);
    ArrayHolder ids(new_fixed(ctx, DFGPU_UINT32, n));
    if (n == 0) { *out_group_ids = ids.release(); return; }
    // clustered keys: group ids are run numbers, no hash table (the shape of GROUP BY over a fact table stored in key order)
    if (ctx->group_run_detection && !ctx->force_hash_collisions && !mask && g->capacity == 0 && (g->n_groups == 0 || g->run_mode) && run_key_type(cols[0]) &&
        groups_intern_runs(ctx, g, cols, nkeys, bk, n, ids.get(), allow_deferred)) { *out_group_ids = ids.release(); return; }
    // one 8-byte integer key column without NULLs: the primitive-key table
    auto plain8 = [](const dfgpu_array* a) { return a && (a->type == DFGPU_INT64 || a->type == DFGPU_UINT64 || a->type == DFGPU_INT32 || a->type == DFGPU_UINT32 || a->type == DFGPU_DATE32) && !a->validity; };
    const bool prim_ok = nkeys == 1 && !ctx->force_hash_collisions && !g->prim_banned && !g->canon_mode && g->capacity == 0 && plain8(cols[0]) && (g->n_groups == 0 || (plain8(g->keys[0]) && g->keys[0]->type == cols[0]->type));
    const bool key4 = prim_ok && type_width(cols[0]->type) == 4;
    if (g->prim_mode && !prim_ok) { g->prim_mode = false; g->pslots.reset(); g->pcap = 0; g->run_mode = g->n_groups > 0; }      // NULLs / another type arrived: re-hash the stored groups below
    if (prim_ok) {
      const uint64_t* mk = mask ? (const uint64_t*)mask->ptr : nullptr;
      const void* kp = cols[0]->values->ptr;
      auto rebuild = [&](uint64_t cap) {
        g->pslots = alloc_buffer(ctx, (size_t)cap * sizeof(PSlot)); g->pcap = cap;
        HIP_CHECK(hipMemsetAsync(g->pslots->ptr, 0xFF, (size_t)cap * sizeof(PSlot), ctx->stream));
        if (g->n_groups) {
          if (key4) hipLaunchKernelGGL((k_prim_insert<uint32_t>), dim3(grid_for(g->n_groups, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)g->keys[0]->values->ptr, g->n_groups, (PSlot*)g->pslots->ptr, cap - 1);
          else hipLaunchKernelGGL((k_prim_insert<unsigned long long>), dim3(grid_for(g->n_groups, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const unsigned long long*)g->keys[0]->values->ptr, g->n_groups, (PSlot*)g->pslots->ptr, cap - 1);
          KERNEL_CHECK();
        }
      };
      uint64_t expect = (uint64_t)(n < (1 << 22) ? n : (1 << 22));
      if (g->size_hint > 0) { uint64_t h = (uint64_t)g->size_hint < (uint64_t)n ? (uint64_t)g->size_hint : (uint64_t)n; if (h > expect) expect = h; }
      uint64_t want = 1ull << 16; while (want < (uint64_t)g->n_groups * 4 + 2 * expect) want <<= 1;
      if (g->pcap < want) rebuild(want);
      g->prim_mode = true; g->run_mode = false;
      BufferPtr tmp = alloc_buffer(ctx, (size_t)n * 4);
      bool banned = false;
      for (;;) {
        zero_scratch(ctx);
        { KernelTimer kt_(ctx, "k_groups_find");
          dim3 fg(grid_for(n, BLOCK * PF_ROWS));
#define PFIND(KT, HM) hipLaunchKernelGGL((k_prim_find<KT, HM>), fg, dim3(BLOCK), 0, ctx->stream, (const KT*)kp, n, mk, (PSlot*)g->pslots->ptr, g->pcap - 1, (uint32_t*)tmp->ptr, (unsigned long long*)ctx->d_scratch64, (uint64_t)256)
          if (key4) { if (mk) PFIND(uint32_t, true); else PFIND(uint32_t, false); } else { if (mk) PFIND(unsigned long long, true); else PFIND(unsigned long long, false); }
#undef PFIND
        }
        KERNEL_CHECK();
        ctx->count_sync("sync:group_table"); fetch_to_pinned(ctx, 0, ctx->d_scratch64, 32);
        if (ctx->h_pinned[3]) { banned = true; break; }
        if (ctx->h_pinned[1] == 0) break;
        if (g->pcap >= (1ull << 31)) fail(DFGPU_RESOURCES_EXHAUSTED, "group table would exceed 2^31 slots");
        uint64_t ncap = g->pcap << 3; if (ncap > (1ull << 31)) ncap = 1ull << 31;
        rebuild(ncap);                                            // the claims of the overfull pass go with the old table
      }
      if (banned) {                                               // a key equals the empty marker: this column takes the general table from now on
        g->prim_banned = true; g->prim_mode = false; g->pslots.reset(); g->pcap = 0; g->run_mode = g->n_groups > 0;
      } else {
        BufferPtr bits = alloc_buffer(ctx, bitmap_bytes((int64_t)g->pcap));
        hipLaunchKernelGGL(k_prim_new_bits, dim3(grid_for((int64_t)g->pcap, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const PSlot*)g->pslots->ptr, (int64_t)g->pcap, (uint64_t*)bits->ptr);
        KERNEL_CHECK();
        ArrayHolder new_slots(mask_to_indices_impl(ctx, (const uint64_t*)bits->ptr, (int64_t)g->pcap));
        int64_t n_new = new_slots.get()->length;
        if (g->n_groups + n_new >= (int64_t)G_NEW) fail(DFGPU_RESOURCES_EXHAUSTED, "more than 2^31 groups");
        if (n_new) {
          ArrayHolder firsts(new_fixed(ctx, DFGPU_UINT32, n_new));
          hipLaunchKernelGGL(k_prim_first_of, dim3(grid_for(n_new, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)new_slots.get()->values->ptr, n_new, (const PSlot*)g->pslots->ptr, (uint32_t*)firsts.get()->values->ptr);
          KERNEL_CHECK();
          int bitsn = 1; while ((1ll << bitsn) < n) bitsn++;
          radix_sort_pairs_u32(ctx, (uint32_t*)firsts.get()->values->ptr, (uint32_t*)new_slots.get()->values->ptr, n_new, bitsn);      // first-seen order
          hipLaunchKernelGGL(k_prim_assign, dim3(grid_for(n_new, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)new_slots.get()->values->ptr, n_new, (uint32_t)g->n_groups, (PSlot*)g->pslots->ptr);
          KERNEL_CHECK();
          groups_append_keys(ctx, g, cols, nkeys, firsts.get(), n_new);
          check_flags(ctx, "groups_intern");
        }
        hipLaunchKernelGGL(k_prim_ids, dim3(grid_for(n, BLOCK * PF_ROWS)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)tmp->ptr, n, (const PSlot*)g->pslots->ptr, (uint32_t*)ids.get()->values->ptr);
        KERNEL_CHECK();
        g->n_groups += n_new;
        if ((uint64_t)g->n_groups * 2 > g->pcap) { uint64_t ncap = g->pcap; while (ncap < (uint64_t)g->n_groups * 4) ncap <<= 1; if (ncap > (1ull << 31)) ncap = 1ull << 31; rebuild(ncap); }
        *out_group_ids = ids.release();
        return;
      }
    }
    if (g->run_mode) {          // a batch broke the order: hash the groups numbered so far, the table is built below
      std::vector<const dfgpu_array*> sk(g->keys.begin(), g->keys.end()); KeySet stored_ks = make_keyset(sk.data(), nkeys);
      groups_reserve_ghash(ctx, g, g->n_groups, 0);
      hipLaunchKernelGGL(k_groups_hash_stored, dim3(grid_for(g->n_groups, BLOCK)), dim3(BLOCK), 0, ctx->stream, stored_ks, g->n_groups, ctx->force_hash_collisions ? 1 : 0, (uint64_t*)g->ghash->ptr);
      KERNEL_CHECK();
      g->run_mode = false;
    }
    // dictionary key columns: intern u32 canonical ids of the dictionary VALUES instead of hashing / comparing the values per row
    std::vector<const dfgpu_array*> eff(cols, cols + nkeys); std::vector<ArrayHolder> sub((size_t)nkeys);
    bool any_dict = false; for (int c = 0; c < nkeys; c++) any_dict |= cols[c]->type == DFGPU_DICTIONARY && cols[c]->dictionary != nullptr;
    bool want_canon = ctx->group_dictionary_canon && !ctx->force_hash_collisions && any_dict && (g->n_groups == 0 || g->canon_mode);
    if (want_canon && g->canon_mode)
      for (int c = 0; c < nkeys; c++) { const dfgpu_array* d = cols[c]->type == DFGPU_DICTIONARY ? cols[c]->dictionary : nullptr; if (d != g->canon[(size_t)c].dict) want_canon = false; }
    if (g->canon_mode && !want_canon) {       // another dictionary (or none): back to value keys; the numbered groups are re-hashed by value
      for (auto*& a : g->canon_keys) { if (a) dfgpu_array_release(a); a = nullptr; }
      for (auto& cc : g->canon) { if (cc.dict) dfgpu_array_release(cc.dict); cc = dfgpu_groups::Canon{}; }
      g->canon_mode = false; g->dense_size = 0; g->dense_map.reset(); g->dense_host.clear(); g->dmap.reset(); g->dmap_size = 0;
      if (g->n_groups) {
        std::vector<const dfgpu_array*> sk(g->keys.begin(), g->keys.end()); KeySet stored_ks = make_keyset(sk.data(), nkeys);
        groups_reserve_ghash(ctx, g, g->n_groups, 0);
        hipLaunchKernelGGL(k_groups_hash_stored, dim3(grid_for(g->n_groups, BLOCK)), dim3(BLOCK), 0, ctx->stream, stored_ks, g->n_groups, 0, (uint64_t*)g->ghash->ptr);
        KERNEL_CHECK();
        if (g->capacity) { uint64_t cap = g->capacity; while (cap < (uint64_t)g->n_groups * 4) cap <<= 1; groups_alloc_table(g, cap); }     // groups numbered through the direct map never entered the table
      }
    }
    if (want_canon) {
      if (!g->canon_mode) {
        g->canon.assign((size_t)nkeys, dfgpu_groups::Canon{}); g->canon_keys.assign((size_t)nkeys, nullptr);
        for (int c = 0; c < nkeys; c++) {
          if (cols[c]->type != DFGPU_DICTIONARY) continue;
          const dfgpu_array* dict = cols[c]->dictionary;
          dfgpu_groups tmp; tmp.ctx = ctx; tmp.nkeys = 1; tmp.keys.assign(1, nullptr); tmp.size_hint = dict->length;
          dfgpu_array* dids = nullptr; dfgpu_status st = dfgpu_groups_intern(ctx, &tmp, &dict, 1, nullptr, &dids);
          if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str());
          ArrayHolder hold(dids);
          auto& cc = g->canon[(size_t)c]; cc.dict = const_cast<dfgpu_array*>(dict); dfgpu_array_retain(cc.dict); cc.ids = dids->values; cc.n_ids = tmp.n_groups;
        }
        g->canon_mode = true;
      }
      // dense composite domain: index a small map directly
      bool all_dict = true; int64_t dsize = 1;
      for (int c = 0; c < nkeys; c++) { if (!g->canon[(size_t)c].dict) { all_dict = false; break; } dsize *= g->canon[(size_t)c].n_ids + 1; if (dsize > DENSE_MAX) break; }
      if (all_dict && dsize <= DENSE_MAX && (g->dense_size == dsize || g->n_groups == 0)) {
        KernelTimer kt_(ctx, "k_groups_dense");
        if (g->dense_size != dsize) { g->dense_size = dsize; g->dense_host.assign((size_t)dsize, G_NONE); g->dense_map = alloc_buffer(ctx, (size_t)dsize * 4); HIP_CHECK(hipMemsetAsync(g->dense_map->ptr, 0xFF, (size_t)dsize * 4, ctx->stream)); }
        DenseCols dc{}; dc.n = nkeys; uint32_t stride = 1;
        for (int c = nkeys - 1; c >= 0; c--) {
          auto& cc = g->canon[(size_t)c]; DenseCol& d = dc.c[c];
          d.keys = cols[c]->values->ptr; d.key_valid = cols[c]->validity ? (const uint64_t*)cols[c]->validity->ptr : nullptr; d.key_type = cols[c]->key_type;
          d.canon = (const uint32_t*)cc.ids->ptr; d.dict_valid = cc.dict->validity ? (const uint64_t*)cc.dict->validity->ptr : nullptr; d.dict_len = cc.dict->length;
          d.n_ids = (uint32_t)cc.n_ids; d.stride = stride; stride *= (uint32_t)cc.n_ids + 1;
        }
        BufferPtr first = alloc_buffer(ctx, (size_t)dsize * 4);
        HIP_CHECK(hipMemsetAsync(first->ptr, 0xFF, (size_t)dsize * 4, ctx->stream));
        const uint64_t* mk = mask ? (const uint64_t*)mask->ptr : nullptr;
        dim3 grid(grid_for(n, BLOCK)), block(BLOCK);
        // straight-line kernels when every code column has the same integer type and nothing is nullable (codes must then be in range:
        // Arrow requires valid dictionary codes; the generic kernels also tolerate out-of-range codes as NULL)
        bool fast = nkeys <= 2; int kt0 = cols[0]->key_type;
        for (int c = 0; c < nkeys; c++) fast = fast && cols[c]->key_type == kt0 && !dc.c[c].key_valid && !dc.c[c].dict_valid;
        fast = fast && (kt0 == DFGPU_INT8 || kt0 == DFGPU_INT16 || kt0 == DFGPU_INT32);
        int fgrid = grid_for(n, BLOCK * DENSE_ROWS, ctx->num_cus * 8);
#define DENSE_FAST(K, NC, WHICH, GRID, ...) do { if (mk) hipLaunchKernelGGL((WHICH<K, NC, true>), dim3(GRID), block, 0, ctx->stream, __VA_ARGS__); else hipLaunchKernelGGL((WHICH<K, NC, false>), dim3(GRID), block, 0, ctx->stream, __VA_ARGS__); } while (0)
#define DENSE_DISPATCH(WHICH, GRID, ...) do { \
          if (kt0 == DFGPU_INT8) { if (nkeys == 1) DENSE_FAST(int8_t, 1, WHICH, GRID, __VA_ARGS__); else DENSE_FAST(int8_t, 2, WHICH, GRID, __VA_ARGS__); } \
          else if (kt0 == DFGPU_INT16) { if (nkeys == 1) DENSE_FAST(int16_t, 1, WHICH, GRID, __VA_ARGS__); else DENSE_FAST(int16_t, 2, WHICH, GRID, __VA_ARGS__); } \
          else { if (nkeys == 1) DENSE_FAST(int32_t, 1, WHICH, GRID, __VA_ARGS__); else DENSE_FAST(int32_t, 2, WHICH, GRID, __VA_ARGS__); } } while (0)
        const int64_t len1 = nkeys == 2 ? dc.c[1].dict_len : 1;
        bool tab = fast && dc.c[0].dict_len * len1 <= DENSE_MAX && dsize <= 65535;
        for (int c = 0; c < nkeys && tab; c++) tab = (((uintptr_t)dc.c[c].keys) & 15) == 0;
        if (tab) DENSE_DISPATCH(k_dense_first_tab, grid_for(n, BLOCK * 4, ctx->num_cus * 8), dc, n, mk, (const uint32_t*)g->dense_map->ptr, (uint32_t*)first->ptr, (int)dsize, (int)len1);
        else if (fast) DENSE_DISPATCH(k_dense_first_fast, fgrid, dc, n, mk, (const uint32_t*)g->dense_map->ptr, (uint32_t*)first->ptr, (int)dsize);
        else hipLaunchKernelGGL(k_dense_first, dim3(fgrid), block, 0, ctx->stream, dc, n, mk, (const uint32_t*)g->dense_map->ptr, (uint32_t*)first->ptr, (int)dsize);
        KERNEL_CHECK();
        std::vector<uint32_t> fh((size_t)dsize);
        ctx->count_sync("sync:dense_groups"); fetch_to_host(ctx, fh.data(), first->ptr, (size_t)dsize * 4);
        std::vector<std::pair<uint32_t, uint32_t>> fresh;             // (first row, composite) of the composites met for the first time
        for (int64_t comp = 0; comp < dsize; comp++) if (fh[(size_t)comp] != G_NONE) fresh.emplace_back(fh[(size_t)comp], (uint32_t)comp);
        std::sort(fresh.begin(), fresh.end());                        // first-seen order
        int64_t n_new = (int64_t)fresh.size();
        if (n_new) {
          std::vector<uint32_t> rows((size_t)n_new);
          for (int64_t k2 = 0; k2 < n_new; k2++) { g->dense_host[fresh[(size_t)k2].second] = (uint32_t)(g->n_groups + k2); rows[(size_t)k2] = fresh[(size_t)k2].first; }
          HIP_CHECK(hipMemcpyAsync(g->dense_map->ptr, g->dense_host.data(), (size_t)dsize * 4, hipMemcpyHostToDevice, ctx->stream));
          ArrayHolder firsts(new_fixed(ctx, DFGPU_UINT32, n_new));
          HIP_CHECK(hipMemcpyAsync(firsts.get()->values->ptr, rows.data(), (size_t)n_new * 4, hipMemcpyHostToDevice, ctx->stream));
          HIP_CHECK(hipStreamSynchronize(ctx->stream));               // rows / dense_host are host vectors
          groups_append_keys(ctx, g, cols, nkeys, firsts.get(), n_new);
        }
        if (fast && allow_deferred) {       // the ids are a pure function of the code columns and the map: the consumer computes them in its own pass
          auto d = std::make_shared<DeferredIds>(); d->dc = dc; d->key_type = kt0; d->mask = mask; d->dense_map = g->dense_map;
          for (int c = 0; c < nkeys; c++) { d->keep.push_back(cols[c]->values); d->keep.push_back(g->canon[(size_t)c].ids); }
          ids.get()->deferred_ids = d;
        }
        else if (fast) DENSE_DISPATCH(k_dense_ids_fast, grid_for(n, BLOCK * DENSE_ROWS), dc, n, mk, (const uint32_t*)g->dense_map->ptr, (uint32_t*)ids.get()->values->ptr);
        else hipLaunchKernelGGL(k_dense_ids, dim3(grid_for(n, BLOCK * DENSE_ROWS)), block, 0, ctx->stream, dc, n, mk, (const uint32_t*)g->dense_map->ptr, (uint32_t*)ids.get()->values->ptr);
#undef DENSE_DISPATCH
#undef DENSE_FAST
        KERNEL_CHECK();
        g->n_groups += n_new;
        *out_group_ids = ids.release();
        return;
      }
      if (g->dense_size) {           // the composite domain outgrew the dense map (cannot happen with unchanged dictionaries); safest: value keys
        fail(DFGPU_INTERNAL, "dense dictionary group map changed size");
      }
      for (int c = 0; c < nkeys; c++) {
        if (cols[c]->type != DFGPU_DICTIONARY) continue;
        auto& cc = g->canon[(size_t)c];
        sub[(size_t)c].a = new_fixed(ctx, DFGPU_UINT32, n);
        KernelTimer kt_(ctx, "k_canon_lookup");
        hipLaunchKernelGGL(k_canon_lookup, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, cols[c]->values->ptr, cols[c]->key_type, cols[c]->validity ? (const uint64_t*)cols[c]->validity->ptr : nullptr, n,
                           (const uint32_t*)cc.ids->ptr, cc.dict->validity ? (const uint64_t*)cc.dict->validity->ptr : nullptr, cc.dict->length, (uint32_t)cc.n_ids, (uint32_t*)sub[(size_t)c].get()->values->ptr);
        KERNEL_CHECK();
        eff[(size_t)c] = sub[(size_t)c].get();
      }
    }
    // one dictionary key column: the canonical ids are dense, so a direct map replaces the hash table (find, first-row marking, numbering)
    if (g->canon_mode && nkeys == 1 && sub[0].get() && (g->dmap || n >= (1 << 16)) && g->canon[0].n_ids + 1 <= (1ll << 28)) {
      KernelTimer kt_(ctx, "k_groups_dmap");
      const int64_t dom = g->canon[0].n_ids + 1;
      const uint32_t* cid = (const uint32_t*)sub[0].get()->values->ptr;
      const uint64_t* mk = mask ? (const uint64_t*)mask->ptr : nullptr;
      if (!g->dmap) {
        g->dmap = alloc_buffer(ctx, (size_t)dom * 4); g->dmap_size = dom;
        HIP_CHECK(hipMemsetAsync(g->dmap->ptr, 0xFF, (size_t)dom * 4, ctx->stream));
        if (g->n_groups) { hipLaunchKernelGGL(k_dm_seed, dim3(grid_for(g->n_groups, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)g->canon_keys[0]->values->ptr, g->n_groups, (uint32_t*)g->dmap->ptr); KERNEL_CHECK(); }
      }
      if (g->dmap_size != dom) fail(DFGPU_INTERNAL, "dictionary group map changed size");
      BufferPtr first = alloc_buffer(ctx, (size_t)dom * 4), heads = alloc_buffer(ctx, bitmap_bytes(n));
      HIP_CHECK(hipMemsetAsync(first->ptr, 0xFF, (size_t)dom * 4, ctx->stream));
      dim3 rgrid(grid_for(n, BLOCK * DM_ROWS)), block(BLOCK);
      hipLaunchKernelGGL(k_dm_first, rgrid, block, 0, ctx->stream, cid, n, mk, (const uint32_t*)g->dmap->ptr, (uint32_t*)first->ptr);
      hipLaunchKernelGGL(k_dm_heads, rgrid, block, 0, ctx->stream, cid, n, mk, (const uint32_t*)g->dmap->ptr, (const uint32_t*)first->ptr, (uint64_t*)heads->ptr);
      KERNEL_CHECK();
      ArrayHolder firsts(mask_to_indices_impl(ctx, (const uint64_t*)heads->ptr, n));
      int64_t n_new = firsts.get()->length;
      if (g->n_groups + n_new >= (int64_t)G_NEW) fail(DFGPU_RESOURCES_EXHAUSTED, "more than 2^31 groups");
      if (n_new) {
        ArrayHolder nc(new_fixed(ctx, DFGPU_UINT32, n_new));
        hipLaunchKernelGGL(k_dm_assign, dim3(grid_for(n_new, BLOCK)), block, 0, ctx->stream, (const uint32_t*)firsts.get()->values->ptr, n_new, cid, (uint32_t)g->n_groups, (uint32_t*)g->dmap->ptr, (uint32_t*)nc.get()->values->ptr);
        KERNEL_CHECK();
        groups_append_keys(ctx, g, cols, nkeys, firsts.get(), n_new);
        dfgpu_array*& dst = g->canon_keys[0];             // the groups' canonical ids, should a later batch fall back to the table
        if (!dst) dst = nc.release();
        else { const dfgpu_array* parts[2] = { dst, nc.get() }; dfgpu_array* cat = nullptr; dfgpu_status st = dfgpu_concat(ctx, parts, 2, &cat); if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str()); dfgpu_array_release(dst); dst = cat; }
        check_flags(ctx, "groups_intern");
      }
      hipLaunchKernelGGL(k_dm_ids, rgrid, block, 0, ctx->stream, cid, n, mk, (const uint32_t*)g->dmap->ptr, (uint32_t*)ids.get()->values->ptr);
      KERNEL_CHECK();
      g->n_groups += n_new;
      *out_group_ids = ids.release();
      return;
    }
    KeySet hk = make_keyset(eff.data(), nkeys);
    KeySet stored{}; int has_stored = 0;
    if (g->n_groups) {
      std::vector<const dfgpu_array*> sk(g->keys.begin(), g->keys.end());
      if (g->canon_mode) for (int c = 0; c < nkeys; c++) if (g->canon_keys[(size_t)c]) sk[(size_t)c] = g->canon_keys[(size_t)c];
      stored = make_keyset(sk.data(), nkeys); has_stored = 1;
    }
    BufferPtr tmp = alloc_buffer(ctx, (size_t)n * 4);
    // optimistic table size (at most 2^23 slots up front); a batch that overfills it is redone on a table 8x larger
    // Known bounds beat the optimism: a dictionary is interned whole (its entries are mostly distinct: that is what it is for), and
    // canonical-id tuples cannot form more groups than the product of their domains.
    uint64_t expect = (uint64_t)(n < (1 << 22) ? n : (1 << 22));
    if (g->canon_mode) { uint64_t dom = 1; for (int c = 0; c < nkeys && dom < (1ull << 31); c++) dom *= g->canon[(size_t)c].dict ? (uint64_t)g->canon[(size_t)c].n_ids + 1 : (1ull << 31); g->size_hint = (int64_t)(dom < (1ull << 31) ? dom : (1ull << 31)); }
    if (g->size_hint > 0) { uint64_t h = (uint64_t)g->size_hint < (uint64_t)n ? (uint64_t)g->size_hint : (uint64_t)n; if (h > expect) expect = h; }
    uint64_t guess = (uint64_t)g->n_groups * 4 + 2 * expect;
    uint64_t want = 1ull << 16; while (want < guess) want <<= 1;
    if (g->capacity < want) groups_alloc_table(g, want);
    int64_t n_new = 0;
    for (;;) {
      // a probe sequence this long means the table is ~95 % full (linear probing: ~1 / (2 (1 - load)^2) steps): stop and grow now rather
      // than crawl to 4096-step sequences first.  Forced collisions put every key in one sequence: its length says nothing there.
      uint64_t step_cap = ctx->force_hash_collisions ? 4096 : 256;
      uint64_t max_steps = g->capacity - 1 < step_cap ? g->capacity - 1 : step_cap;
      zero_scratch(ctx);
      { KernelTimer kt_(ctx, "k_groups_find");
      hipLaunchKernelGGL(k_groups_find, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, hk, stored, has_stored, n, mask ? (const uint64_t*)mask->ptr : nullptr,
                         ctx->force_hash_collisions ? 1 : 0, (uint64_t*)g->slots->ptr, g->capacity - 1, (uint32_t*)g->first_row->ptr, (uint32_t*)tmp->ptr,
                         (unsigned long long*)ctx->d_scratch64, max_steps); }
      KERNEL_CHECK();
      if (read_scratch(ctx, 1) == 0) break;
      if (g->capacity >= (1ull << 31)) fail(DFGPU_RESOURCES_EXHAUSTED, "group table would exceed 2^31 slots");
      uint64_t ncap = g->capacity << 3; if (ncap > (1ull << 31)) ncap = 1ull << 31;
      groups_alloc_table(g, ncap);
    }
    BufferPtr bits = alloc_buffer(ctx, bitmap_bytes(n));
    hipLaunchKernelGGL(k_groups_mark_first, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)tmp->ptr, (const uint32_t*)g->first_row->ptr, n, (uint64_t*)bits->ptr);
    KERNEL_CHECK();
    ArrayHolder firsts(mask_to_indices_impl(ctx, (const uint64_t*)bits->ptr, n));
    n_new = firsts.get()->length;
    if (g->n_groups + n_new >= (int64_t)G_NEW) fail(DFGPU_RESOURCES_EXHAUSTED, "more than 2^31 groups");
    if (n_new) {
      groups_reserve_ghash(ctx, g, g->n_groups + n_new, g->n_groups);
      hipLaunchKernelGGL(k_groups_assign, dim3(grid_for(n_new, BLOCK)), dim3(BLOCK), 0, ctx->stream, hk, (const uint32_t*)firsts.get()->values->ptr, n_new, (const uint32_t*)tmp->ptr,
                         (uint64_t*)g->slots->ptr, (uint32_t*)g->first_row->ptr, (uint32_t)g->n_groups, ctx->force_hash_collisions ? 1 : 0, (uint64_t*)g->ghash->ptr);
      KERNEL_CHECK();
      groups_append_keys(ctx, g, cols, nkeys, firsts.get(), n_new);
      if (g->canon_mode) for (int c = 0; c < nkeys; c++) {          // the new groups' canon tuples, for comparisons in later batches
        if (!sub[(size_t)c].get()) continue;
        ArrayHolder nk(take_impl(ctx, sub[(size_t)c].get(), firsts.get()->values->ptr, 4, nullptr, n_new));
        dfgpu_array*& dst = g->canon_keys[(size_t)c];
        if (!dst) dst = nk.release();
        else { const dfgpu_array* parts[2] = { dst, nk.get() }; dfgpu_array* cat = nullptr; dfgpu_status st = dfgpu_concat(ctx, parts, 2, &cat); if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str()); dfgpu_array_release(dst); dst = cat; }
      }
      check_flags(ctx, "groups_intern");
    }
    hipLaunchKernelGGL(k_groups_finalize, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)tmp->ptr, (const uint64_t*)g->slots->ptr, n, (uint32_t*)ids.get()->values->ptr);
    KERNEL_CHECK();
    g->n_groups += n_new;
    // keep the load factor <= 1/2 for the next batch (rehash of the numbered groups only: no key comparisons)
    if ((uint64_t)g->n_groups * 2 > g->capacity) { uint64_t ncap = g->capacity; while (ncap < (uint64_t)g->n_groups * 4) ncap <<= 1; if (ncap > (1ull << 31)) ncap = 1ull << 31; groups_alloc_table(g, ncap); }
    *out_group_ids = ids.release();
  });
}

dfgpu_status dfgpu_groups_emit(dfgpu_ctx* ctx, dfgpu_groups* g, dfgpu_array** out_cols) {
  return guard(ctx, [&] {
    if (!g || !out_cols) fail(DFGPU_INVALID_ARGUMENT, "groups_emit: null argument");
    if (g->n_groups == 0) fail(DFGPU_INVALID_ARGUMENT, "groups_emit: no groups interned yet (key types unknown)");
    groups_resolve_keys(ctx, g);
    for (int c = 0; c < g->nkeys; c++) { dfgpu_array_retain(g->keys[c]); out_cols[c] = g->keys[c]; }
  });
}
/* see include/dfgpu.h */
dfgpu_status dfgpu_groups_emit_deferred(dfgpu_ctx* ctx, dfgpu_groups* g, dfgpu_array** out_sources, dfgpu_array** out_rows) {
  return guard(ctx, [&] {
    if (!g || !out_sources || !out_rows) fail(DFGPU_INVALID_ARGUMENT, "groups_emit_deferred: null argument");
    if (!g->lazy_rows) fail(DFGPU_NOT_IMPLEMENTED, "groups_emit_deferred: the keys are stored (no gather is pending)");
    for (auto* a : g->lazy_src) if (a->type == DFGPU_DICTIONARY) fail(DFGPU_NOT_IMPLEMENTED, "groups_emit_deferred: dictionary key columns are emitted as values");
    for (int c = 0; c < g->nkeys; c++) { dfgpu_array_retain(g->lazy_src[(size_t)c]); out_sources[c] = g->lazy_src[(size_t)c]; }
    dfgpu_array_retain(g->lazy_rows); *out_rows = g->lazy_rows;
  });
}

/* see include/dfgpu.h */
dfgpu_status dfgpu_groups_emit_first(dfgpu_ctx* ctx, dfgpu_groups* g, int64_t n, dfgpu_array** out_cols) {
  return guard(ctx, [&] {
    if (!g || !out_cols || n < 0) fail(DFGPU_INVALID_ARGUMENT, "groups_emit_first: bad argument");
    if (g->n_groups == 0) fail(DFGPU_INVALID_ARGUMENT, "groups_emit_first: no groups interned yet (key types unknown)");
    groups_resolve_keys(ctx, g);
    const int64_t total = g->n_groups, k = n < total ? n : total; const int32_t nkeys = g->nkeys;
    std::vector<ArrayHolder> first((size_t)nkeys), rest((size_t)nkeys);
    for (int c = 0; c < nkeys; c++) {
      dfgpu_array* a = nullptr; dfgpu_status st = dfgpu_array_slice(ctx, g->keys[c], 0, k, &a); if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str()); first[(size_t)c].a = a;
      if (k < total) { dfgpu_array* b = nullptr; st = dfgpu_array_slice(ctx, g->keys[c], k, total - k, &b); if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str()); rest[(size_t)c].a = b; }
    }
    // the remaining groups are renumbered from 0 (EmitTo::take_needed, expr/src/groups_accumulator.rs:44-57; GroupValuesRows::emit First(n) rebuilds its map the same way,
    // group_values/row.rs:176-212): a fresh table over the remaining keys -- distinct, interned in order, so they get ids 0 .. total - k - 1
    g->~dfgpu_groups(); new (g) dfgpu_groups(); g->ctx = ctx; g->nkeys = nkeys; g->keys.assign((size_t)nkeys, nullptr);
    if (k < total) {
      std::vector<const dfgpu_array*> rp; for (auto& h : rest) rp.push_back(h.get());
      dfgpu_array* ids = nullptr; dfgpu_status st = dfgpu_groups_intern(ctx, g, rp.data(), nkeys, nullptr, &ids); if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str());
      dfgpu_array_release(ids);
      if (g->n_groups != total - k) fail(DFGPU_INTERNAL, "groups_emit_first: %lld groups left after re-interning %lld keys", (long long)g->n_groups, (long long)(total - k));
    }
    for (int c = 0; c < nkeys; c++) out_cols[c] = first[(size_t)c].release();
  });
}

}  // extern "C"
