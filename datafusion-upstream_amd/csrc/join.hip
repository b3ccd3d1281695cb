// join.hip -- a2-a6: HashJoinExec build / probe / join-type index algebra on gfx950.
//
// Reference: datafusion/physical-plan/src/joins/hash_join.rs + joins/utils.rs (see include/dfgpu.h).
//
// Table layout (HBM): open addressing, linear probing, one 8-byte slot = (hash tag : 32 | representative
// build row : 32), capacity = next_pow2(2 * rows) so a probe touches one 64-B line in the common case.
// Instead of the reference's serial `next[]` chains (joins/utils.rs:203-229) the build is a parallel
// find-or-insert (64-bit CAS) that groups equal KEYS per slot, followed -- only when some key repeats --
// by a stable radix sort of (slot, row) into a CSR row list.  That makes the probe deterministic and
// reproduces the reference's emission order (probe order, then build input order; hash_join.rs:161-197)
// without pointer chasing; unique-key builds (every TPC-H PK join) skip the CSR entirely.
// The probe is count -> exclusive scan -> fill, so output order never depends on scheduling.
#include "join_table.h"

namespace dfgpu {
constexpr uint64_t SLOT_EMPTY = ~0ull;
constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;
}
using namespace dfgpu;


namespace dfgpu {

__device__ inline bool row_selected(const uint64_t* mask, int64_t i) { return mask == nullptr || bit_get(mask, i); }

__global__ void __launch_bounds__(BLOCK) k_join_build(KeySet ks, int64_t n, const uint64_t* mask, int null_eq, int force_zero,
                                                      uint64_t* slots, uint32_t* slot_count, uint64_t cap_mask, uint32_t* row_slot,
                                                      unsigned long long* counters /* [0]=1 when some key repeats */) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t my_slot = NO_SLOT;
  if (row_selected(mask, i)) {
    bool any_null; uint64_t h = keyset_hash(ks, i, 0, &any_null);
    if (force_zero) h = 0;
    if (!any_null || null_eq) {     // a NULL key never matches (eq -> NULL, hash_join.rs:1067-1076) unless null_equals_null
      uint64_t tag = h >> 32, s = h & cap_mask, mine = (tag << 32) | (uint64_t)i;
      for (uint64_t step = 0; step <= cap_mask; step++) {
        uint64_t cur = slots[s];
        if (cur == SLOT_EMPTY) {
          cur = atomicCAS((unsigned long long*)&slots[s], (unsigned long long)SLOT_EMPTY, (unsigned long long)mine);
          if (cur == SLOT_EMPTY) cur = mine;
        }
        if ((cur >> 32) == tag) {
          int64_t rep = (int64_t)(cur & 0xFFFFFFFFull);
          if (rep == i || keyset_equal(ks, i, ks, rep, true)) { my_slot = (uint32_t)s; if (rep != i) counters[0] = 1ull; break; }   // joined another row's group: key repeats
        }
        s = (s + 1) & cap_mask;
      }
      // no single-address counters here: 1e7 atomics on one word serialise (measured 10 ms per build at SF100)
      // no per-row counters: group sizes are only counted (k_count_slots) when some key repeats (flag above)
    }
  }
  row_slot[i] = my_slot;
}

__global__ void __launch_bounds__(BLOCK) k_count_slots(const uint32_t* row_slot, int64_t n, uint32_t* slot_count) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n && row_slot[i] != NO_SLOT) atomicAdd(&slot_count[row_slot[i]], 1u);
}
__global__ void __launch_bounds__(BLOCK) k_fix_unslotted(uint32_t* row_slot, int64_t n, uint32_t sentinel) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n && row_slot[i] == NO_SLOT) row_slot[i] = sentinel;
}

// ---- probe pass 1: one match bit per probe row (order-preserving compaction of the bitmap gives the probe indices)
__global__ void __launch_bounds__(BLOCK) k_probe_match_hash(KeySet bks, KeySet pks, int64_t n, const uint64_t* mask, int null_eq, int force_zero,
                                                            const uint64_t* slots, uint64_t cap_mask, uint64_t* match_bits) {
  int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  bool hit = false;
  if (j < n && row_selected(mask, j)) {
    bool any_null; uint64_t h = keyset_hash(pks, j, 0, &any_null);
    if (force_zero) h = 0;
    if (!any_null || null_eq) {
      uint64_t tag = h >> 32, s = h & cap_mask;
      for (uint64_t step = 0; step <= cap_mask; step++) {
        uint64_t cur = slots[s];
        if (cur == SLOT_EMPTY) break;
        if ((cur >> 32) == tag && keyset_equal(bks, (int64_t)(cur & 0xFFFFFFFFull), pks, j, null_eq != 0)) { hit = true; break; }
        s = (s + 1) & cap_mask;
      }
    }
  }
  uint64_t m = ballot64(hit);
  if (lane_id() == 0 && (j >> 6) < ((n + 63) >> 6)) match_bits[j >> 6] = m;
}
// The same pass for the plain case -- one or two 8-byte integer key columns, no NULLs on either side: HP_ROWS rows per lane, each
// level of the chain (key -> first slot -> build key behind a matching tag) issued for all of them before the next; the rare longer
// probe sequences finish row by row.  The slot a row was found in is kept (4 B per probe row), so pass 2 is a gather through the
// match list instead of a second hash probe.
constexpr int HP_ROWS = 4;
template <int NK, bool HAS_MASK>
__global__ void __launch_bounds__(BLOCK) k_probe_hash_i64(const uint64_t* b0, const uint64_t* b1, const uint64_t* p0, const uint64_t* p1, int64_t n, const uint64_t* mask,
                                                          const uint64_t* slots, uint64_t cap_mask, uint64_t* match_bits, uint32_t* found_slot) {
  const int64_t base = (int64_t)blockIdx.x * BLOCK * HP_ROWS + threadIdx.x;
  uint64_t k0[HP_ROWS], k1[HP_ROWS], tag[HP_ROWS], sl[HP_ROWS], cur[HP_ROWS], v0[HP_ROWS], v1[HP_ROWS]; bool on[HP_ROWS];
#pragma unroll
  for (int q = 0; q < HP_ROWS; q++) {
    int64_t i = base + (int64_t)q * BLOCK, ic = i < n ? i : n - 1;
    on[q] = i < n && (!HAS_MASK || bit_get(mask, i));
    k0[q] = p0[ic]; k1[q] = NK == 2 ? p1[ic] : 0;
  }
#pragma unroll
  for (int q = 0; q < HP_ROWS; q++) {
    uint64_t h = mix64(k0[q]); if (NK == 2) h = combine_hashes(mix64(k1[q]), h);          // keyset_hash over the key columns, seed 0
    tag[q] = h >> 32; sl[q] = h & cap_mask;
  }
#pragma unroll
  for (int q = 0; q < HP_ROWS; q++) cur[q] = slots[sl[q]];
#pragma unroll
  for (int q = 0; q < HP_ROWS; q++) { uint64_t r = (cur[q] != SLOT_EMPTY && (cur[q] >> 32) == tag[q]) ? (cur[q] & 0xFFFFFFFFull) : 0; v0[q] = b0[r]; v1[q] = NK == 2 ? b1[r] : 0; }
#pragma unroll
  for (int q = 0; q < HP_ROWS; q++) {
    bool hit = false;
    if (on[q]) {
      uint64_t c = cur[q], s2 = sl[q];
      if (c != SLOT_EMPTY && (c >> 32) == tag[q] && v0[q] == k0[q] && (NK == 1 || v1[q] == k1[q])) hit = true;
      else if (c != SLOT_EMPTY) {
        for (uint64_t step = 0; step < cap_mask; step++) {
          s2 = (s2 + 1) & cap_mask; c = slots[s2];
          if (c == SLOT_EMPTY) break;
          if ((c >> 32) == tag[q]) { uint64_t r = c & 0xFFFFFFFFull; if (b0[r] == k0[q] && (NK == 1 || b1[r] == k1[q])) { hit = true; break; } }
        }
      }
      if (hit) found_slot[base + (int64_t)q * BLOCK] = (uint32_t)s2;
    }
    int64_t i = base + (int64_t)q * BLOCK;
    uint64_t m = ballot64(hit);
    if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) match_bits[i >> 6] = m;
  }
}
// pass 2 after k_probe_hash_i64: the build row (unique keys) or the slot and its group size (repeated keys) of every match
__global__ void __launch_bounds__(BLOCK) k_probe_found(const uint32_t* rows, int64_t m, const uint32_t* found_slot, const uint64_t* slots, const uint32_t* slot_count, int unique,
                                                       uint64_t* out_build, uint32_t* out_slot, uint32_t* out_cnt) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= m) return;
  uint32_t s2 = found_slot[rows[i]];
  if (unique) out_build[i] = slots[s2] & 0xFFFFFFFFull;
  else { out_slot[i] = s2; out_cnt[i] = slot_count[s2]; }
}
// dense integer key domain: stream the probe keys, test one bit each
constexpr int PM_ROWS = 8;          // rows per lane: 4 iterations x 2 consecutive keys (one 16-B load for Int64 keys)
__device__ inline uint64_t spread32(uint64_t x) {      // bit i -> bit 2i
  x &= 0xFFFFFFFFull;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFull; x = (x | (x << 8)) & 0x00FF00FF00FF00FFull; x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x << 2)) & 0x3333333333333333ull; x = (x | (x << 1)) & 0x5555555555555555ull; return x;
}
// One 512-row chunk per wave, launched 1:1 (measured faster than a persistent grid: consecutive workgroups keep the key
// stream and the bitmap window local).  Mask / validity presence and "the whole chunk is in range" are compile-time so
// that the loads of one chunk are not split across branches (measured 1.34 -> 1.18 ms per 600M keys).
// Bitmap window: clustered probe keys (a fact table stored in key order) put the wave's 512 keys inside one run of 64
// bitmap words, so the wave loads that run once, coalesced, from the first key of its chunk and looks bits up with
// ds_bpermute; keys outside the window take the per-lane gather (profiles/experiments/probe_stream_microbench.hip:
// 1.29 -> 1.05 ms per 600M sorted keys, unchanged for random keys).
template <typename T, bool HAS_MASK, bool HAS_VALID, bool FULL>
__device__ inline void probe_match_chunk(const T* keys, const uint64_t* key_valid, const uint64_t* mask, int64_t n, int64_t kmin, uint64_t range,
                                         const uint64_t* bitmap, uint64_t* match_bits, int64_t base, int lane) {
  // Every load of the chunk that does not depend on another is issued before the first wait: the four 16-byte key loads and the four selection words (round 4: the kernel
  // used to load keys[base] on its own, wait, and only then issue the rest -- two memory round trips per wave in a row -- and fetched every selection word inside the loop
  // behind the previous iteration's store).  The window's first key is lane 0's first key.
  T k[PM_ROWS / 2][2];
#pragma unroll
  for (int r = 0; r < PM_ROWS / 2; r++) {                      // lane l owns rows base + 128 r + 2l, +1
    int64_t j = base + r * 2 * WAVE + 2 * lane;
    if (FULL || j + 1 < n) { struct alignas(2 * sizeof(T)) P { T a, b; }; P p = *(const P*)(keys + j); k[r][0] = p.a; k[r][1] = p.b; }
    else { k[r][0] = j < n ? keys[j] : (T)0; k[r][1] = 0; }
  }
  uint64_t mws[PM_ROWS / 2];
#pragma unroll
  for (int r = 0; r < PM_ROWS / 2; r++) { int64_t j = base + r * 2 * WAVE + 2 * lane; mws[r] = (HAS_MASK && (FULL || j < n)) ? mask[j >> 6] >> (j & 63) : 3ull; }
  uint64_t d0 = (uint64_t)((int64_t)k[0][0] - kmin);           // lane 0: row `base` (the caller guarantees base < n)
  d0 = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)d0) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(d0 >> 32)) << 32);
  int64_t w0i = d0 < range ? (int64_t)(d0 >> 6) : 0;
  uint64_t win = (uint64_t)(w0i + lane) * 64 < range ? bitmap[w0i + lane] : 0ull;
#pragma unroll
  for (int r = 0; r < PM_ROWS / 2; r++) {
    int64_t j = base + r * 2 * WAVE + 2 * lane;
    const uint64_t mw = mws[r];
    bool h[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
      uint64_t d = (uint64_t)((int64_t)k[r][e] - kmin);
      bool go = (FULL || j + e < n) && ((mw >> e) & 1) && (!HAS_VALID || valid_at(key_valid, j + e)) && d < range;
      int64_t rel = (int64_t)(d >> 6) - w0i;
      uint64_t word = __shfl(win, (int)(rel & 63), 64);
      if (go && (rel < 0 || rel >= WAVE)) word = bitmap[d >> 6];
      h[e] = go && ((word >> (d & 63)) & 1ull);
    }
    uint64_t be = ballot64(h[0]), bo = ballot64(h[1]);           // wave-uniform: the interleave below runs on the scalar unit
    uint64_t w0 = spread32(be) | (spread32(bo) << 1), w1 = spread32(be >> 32) | (spread32(bo >> 32) << 1);
    int64_t wbase = (base >> 6) + 2 * r;
    if (lane == 0) { if (FULL || base + r * 2 * WAVE < n) match_bits[wbase] = w0; if (FULL || base + r * 2 * WAVE + WAVE < n) match_bits[wbase + 1] = w1; }
  }
}
template <typename T, bool HAS_MASK, bool HAS_VALID>
__global__ void __launch_bounds__(BLOCK) k_probe_match_bitmap(const T* keys, const uint64_t* key_valid, const uint64_t* mask, int64_t n, int64_t kmin, uint64_t range,
                                                              const uint64_t* bitmap, uint64_t* match_bits) {
  int lane = lane_id();
  int64_t base = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * (WAVE * PM_ROWS);
  if (base + WAVE * PM_ROWS <= n) probe_match_chunk<T, HAS_MASK, HAS_VALID, true>(keys, key_valid, mask, n, kmin, range, bitmap, match_bits, base, lane);
  else if (base < n) probe_match_chunk<T, HAS_MASK, HAS_VALID, false>(keys, key_valid, mask, n, kmin, range, bitmap, match_bits, base, lane);
}
// ---- probe pass 2: the matched probe rows (ascending) look their key group up; unique builds emit the build row directly
__global__ void __launch_bounds__(BLOCK) k_probe_lookup(KeySet bks, KeySet pks, const uint32_t* rows, int64_t m, int null_eq, int force_zero,
                                                        const uint64_t* slots, const uint32_t* slot_count, uint64_t cap_mask, int unique,
                                                        uint64_t* out_build, uint32_t* out_slot, uint32_t* out_cnt, uint32_t* flags) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= m) return;
  int64_t j = rows[i];
  bool any_null; uint64_t h = keyset_hash(pks, j, 0, &any_null);
  if (force_zero) h = 0;
  uint64_t tag = h >> 32, s = h & cap_mask; bool found = false;
  for (uint64_t step = 0; step <= cap_mask; step++) {
    uint64_t cur = slots[s];
    if (cur == SLOT_EMPTY) break;
    if ((cur >> 32) == tag && keyset_equal(bks, (int64_t)(cur & 0xFFFFFFFFull), pks, j, null_eq != 0)) { found = true; break; }
    s = (s + 1) & cap_mask;
  }
  if (!found) { atomicOr(flags, DFGPU_FLAG_TABLE_FULL); s = 0; }       // cannot happen: pass 1 is exact
  if (unique) out_build[i] = found ? (slots[s] & 0xFFFFFFFFull) : 0;
  else { out_slot[i] = (uint32_t)s; out_cnt[i] = found ? slot_count[s] : 0; }
}
// ---- probe pass 3 (repeated build keys): every matched probe row emits its key group in build input order
// Load-balanced (the same walk as k_pj_expand, pjoin.hip): a wave owns 64 consecutive matches, whose pairs are consecutive in the output; the lanes take 64 pairs at a time and find
// the match a pair belongs to by a 6-step search over the wave's offsets (shuffles).  A thread per match looping over its run writes count-strided and a hot key serialises a wave.
__global__ void __launch_bounds__(BLOCK) k_probe_expand(const uint32_t* rows, const uint32_t* slot_of, const uint32_t* cnt, const uint64_t* offsets, int64_t m,
                                                        const uint32_t* slot_start, const uint32_t* csr_rows, uint64_t* out_build, uint32_t* out_probe) {
  const int lane = lane_id(); const int64_t w0 = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) - lane;
  if (w0 >= m) return;
  const int64_t i = w0 + lane; const bool on = i < m;
  const uint32_t c = on ? cnt[i] : 0u, st = on ? slot_start[slot_of[i]] : 0u, j = on ? rows[i] : 0u;
  const uint64_t o = on ? offsets[i] : 0ull;
  const uint64_t base = (uint64_t)__shfl((long long)o, 0, 64);
  const uint64_t rel = o - base;
  const int last = 63 - __clzll((long long)ballot64(on));
  const uint64_t total = (uint64_t)__shfl((long long)(rel + c), last, 64);
  for (uint64_t k0 = 0; k0 < total; k0 += WAVE) {
    const uint64_t k = k0 + lane;
    // the last lane in [0, last] with rel <= k and a non-empty run: runs of zero pairs (cnt 0 never reaches here, but keep the search right) share rel with their successor, and
    // "last lane with rel <= k" then lands on the successor, which is the one that owns k
    int lo = 0, hi = last;
#pragma unroll
    for (int step = 0; step < 6; step++) { const int mid = (lo + hi + 1) >> 1; const uint64_t r = (uint64_t)__shfl((long long)rel, mid, 64); if (r <= k) lo = mid; else hi = mid - 1; }
    const uint64_t r0 = (uint64_t)__shfl((long long)rel, lo, 64); const uint32_t s0 = (uint32_t)__shfl((int)st, lo, 64), p0 = (uint32_t)__shfl((int)j, lo, 64);
    if (k < total) { const uint32_t at = s0 + (uint32_t)(k - r0); out_build[base + k] = csr_rows ? csr_rows[at] : at; out_probe[base + k] = p0; }      // csr_rows == null: the run is contiguous (rank_runs)
  }
}
// ---- membership bitmap of the build keys
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_key_minmax(const T* keys, const uint32_t* row_slot, int64_t n, long long* mn, long long* mx) {
  long long lo = INT64_MAX, hi = INT64_MIN;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
    if (row_slot[i] != NO_SLOT) { long long v = (long long)keys[i]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { long long a = __shfl_xor(lo, d, 64), b = __shfl_xor(hi, d, 64); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
  __shared__ long long slo[BLOCK / WAVE], shi[BLOCK / WAVE];
  if (lane_id() == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {       // one atomic pair per workgroup (atomics on one word serialise)
    for (int w = 1; w < BLOCK / WAVE; w++) { lo = slo[w] < lo ? slo[w] : lo; hi = shi[w] > hi ? shi[w] : hi; }
    if (lo <= hi) { atomicMin(mn, lo); atomicMax(mx, hi); }
  }
}
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_key_setbits(const T* keys, const uint32_t* row_slot, int64_t n, int64_t kmin, uint64_t* bitmap) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n || row_slot[i] == NO_SLOT) return;
  uint64_t d = (uint64_t)((int64_t)keys[i] - kmin);
  atomicOr((unsigned long long*)&bitmap[d >> 6], 1ull << (d & 63));
}
// ---- rank index: the build keys are strictly increasing, so the membership bitmap alone locates the build row
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_check_increasing(const T* keys, int64_t n, unsigned long long* out /* [0] not sorted, [1] first key, [2] last key, [3] repeats */) {
  bool bad = false, dup = false;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x + 1; i < n; i += (int64_t)gridDim.x * BLOCK) { T a = keys[i - 1], b = keys[i]; bad |= b < a; dup |= a == b; }
  if (ballot64(bad) && lane_id() == 0) out[0] = 1ull;
  if (ballot64(dup) && lane_id() == 0) out[3] = 1ull;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[1] = (unsigned long long)(long long)keys[0]; out[2] = (unsigned long long)(long long)keys[n - 1]; }   // one read-back for all of them
}
// ---- rank index over UNSORTED unique keys (a key column after a hash repartition, a filtered dimension table in arrival order): the membership bitmap still ranks the
// keys; one more array, rank -> build row, takes the place of the sort order.  Setting the bits finds repeated keys (the bit is already there).
// rows != null: the selected rows as a list whose length is the device word *d_count (mask_to_indices_uncounted) -- a selection that keeps one row in ten costs a tenth of the lanes
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_key_setbits_unique(const T* keys, const uint64_t* mask, const uint32_t* rows, const unsigned long long* d_count, int64_t n, int64_t kmin, uint64_t* bitmap, unsigned long long* dup, const T* ckeys = nullptr) {
  const int64_t m = rows ? (int64_t)*d_count : n;
  for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < m; j += (int64_t)gridDim.x * BLOCK) {
    const int64_t i = rows ? (int64_t)rows[j] : j;
    if (!rows && mask && !bit_get(mask, i)) continue;
    const uint64_t d = (uint64_t)((int64_t)(ckeys ? ckeys[j] : keys[i]) - kmin), bit = 1ull << (d & 63);
    const unsigned long long old = atomicOr((unsigned long long*)&bitmap[d >> 6], (unsigned long long)bit);
    if (old & bit) *dup = 1ull;
  }
}
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_rank_rows(const T* keys, const uint64_t* mask, const uint32_t* rows, const unsigned long long* d_count, int64_t n, int64_t kmin, const uint64_t* bitmap, const uint32_t* prefix, uint32_t* row_of_rank, const T* ckeys = nullptr) {
  const int64_t m = rows ? (int64_t)*d_count : n;
  for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < m; j += (int64_t)gridDim.x * BLOCK) {
    const int64_t i = rows ? (int64_t)rows[j] : j;
    if (!rows && mask && !bit_get(mask, i)) continue;
    const uint64_t d = (uint64_t)((int64_t)(ckeys ? ckeys[j] : keys[i]) - kmin);
    row_of_rank[prefix[d >> 6] + (uint32_t)__popcll(bitmap[d >> 6] & ((1ull << (d & 63)) - 1ull))] = (uint32_t)i;
  }
}
// heads bit i = row i starts a run of equal keys
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_key_run_heads(const T* keys, int64_t n, uint64_t* heads) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  bool h = i < n && (i == 0 || keys[i - 1] != keys[i]);
  uint64_t m = ballot64(h);
  if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) heads[i >> 6] = m;
}
// matched probe rows of a rank_runs build: rank of the key -> (run id, run length)
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_probe_lookup_runs(const T* pkeys, const uint32_t* rows, int64_t m, int64_t kmin, const uint64_t* bitmap, const uint32_t* prefix,
                                                             const uint32_t* run_starts, int64_t n_runs, int64_t n_build, uint32_t* out_run, uint32_t* out_cnt) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= m) return;
  uint64_t d = (uint64_t)((int64_t)pkeys[rows[i]] - kmin);
  uint32_t r = prefix[d >> 6] + (uint32_t)__popcll(bitmap[d >> 6] & ((1ull << (d & 63)) - 1ull));
  uint32_t st = run_starts[r], en = (int64_t)r + 1 < n_runs ? run_starts[r + 1] : (uint32_t)n_build;
  out_run[i] = r; out_cnt[i] = en - st;
}
// set the bit of every selected row's key.  Equal bitmap words of neighbouring lanes are OR-combined first (segmented
// scan over runs of the same word; sorted keys put 16+ lanes on one word) so one atomic per run reaches L2.
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_key_setbits_masked(const T* keys, const uint64_t* mask, int64_t n, int64_t kmin, uint64_t* bitmap) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  int lane = lane_id();
  bool on = i < n && row_selected(mask, i);
  uint64_t d = on ? (uint64_t)((int64_t)keys[i] - kmin) : 0;
  int64_t w = on ? (int64_t)(d >> 6) : -1 - lane;            // unselected lanes never combine
  uint64_t b = on ? 1ull << (d & 63) : 0ull;
#pragma unroll
  for (int s = 1; s < WAVE; s <<= 1) {
    uint64_t ob = __shfl_up(b, s, 64); int64_t ow = __shfl_up(w, s, 64);
    if (lane >= s && ow == w) b |= ob;
  }
  int64_t nxt = __shfl_down(w, 1, 64);
  if (on && (lane == WAVE - 1 || nxt != w)) atomicOr((unsigned long long*)&bitmap[w], (unsigned long long)b);
}
// The same over a LIST of rows whose length is a device word (mask_to_indices_uncounted): a selection that keeps one row in ten of a 150 M-row column would spend the
// kernel above on 150 M lanes for 15 M bits (SF100 Q3's join of lineitem with the orders the first join selected: 0.68 ms); here the 15 M selected rows fill the lanes.
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_key_setbits_rows(const T* keys, const uint32_t* rows, const unsigned long long* d_count, int64_t kmin, uint64_t* bitmap) {
  const int64_t m = (int64_t)*d_count; const int lane = lane_id();
  for (int64_t j0 = (int64_t)blockIdx.x * BLOCK; j0 < m; j0 += (int64_t)gridDim.x * BLOCK) {
    const int64_t j = j0 + threadIdx.x; const bool on = j < m;
    uint64_t d = on ? (uint64_t)((int64_t)keys[rows[j]] - kmin) : 0;
    int64_t w = on ? (int64_t)(d >> 6) : -1 - lane;
    uint64_t b = on ? 1ull << (d & 63) : 0ull;
#pragma unroll
    for (int s2 = 1; s2 < WAVE; s2 <<= 1) {
      uint64_t ob = __shfl_up(b, s2, 64); int64_t ow = __shfl_up(w, s2, 64);
      if (lane >= s2 && ow == w) b |= ob;
    }
    int64_t nxt = __shfl_down(w, 1, 64);
    if (on && (lane == WAVE - 1 || nxt != w)) atomicOr((unsigned long long*)&bitmap[w], (unsigned long long)b);
  }
}
__global__ void __launch_bounds__(BLOCK) k_popc_words(const uint64_t* words, int64_t nw, uint32_t* out) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < nw) out[i] = (uint32_t)__popcll(words[i]);
}
// LR_ROWS matches per lane, each level of the dependent chain (row -> key -> bitmap word + prefix -> selected row) issued for all of them
// before the next level; ALL = every probe row matched (the match list is 0 .. m-1 and is not read); SEL = a build selection is fused.
constexpr int LR_ROWS = 4;
template <typename T, bool ALL, bool SEL>
__global__ void __launch_bounds__(BLOCK) k_probe_lookup_rank(const T* pkeys, const uint32_t* rows, int64_t m, int64_t kmin, const uint64_t* bitmap,
                                                             const uint32_t* prefix, const uint32_t* sel_rows, int identity, uint64_t* out_build,
                                                             const uint64_t* rows_valid = nullptr, int64_t n_probe = 0) {
  const int64_t base = (int64_t)blockIdx.x * BLOCK * LR_ROWS + threadIdx.x;
  int64_t ic[LR_ROWS]; uint64_t d[LR_ROWS], r[LR_ROWS];
#pragma unroll
  for (int q = 0; q < LR_ROWS; q++) { int64_t i = base + (int64_t)q * BLOCK; ic[q] = i < m ? i : m - 1; }
  bool live[LR_ROWS];
#pragma unroll
  for (int q = 0; q < LR_ROWS; q++) live[q] = rows_valid == nullptr || valid_at(rows_valid, ic[q]);
  if constexpr (!ALL) {
#pragma unroll
    for (int q = 0; q < LR_ROWS; q++) ic[q] = rows[ic[q]];
  }
  if (rows_valid) {       // a NULL entry of `rows` (an unmatched row of an outer join above) names no probe row: read row 0, give back key_min's slot, the entry stays NULL
#pragma unroll
    for (int q = 0; q < LR_ROWS; q++) if (!live[q] || ic[q] >= n_probe) ic[q] = 0;
  }
#pragma unroll
  for (int q = 0; q < LR_ROWS; q++) d[q] = live[q] ? (uint64_t)((int64_t)pkeys[ic[q]] - kmin) : 0;      // pass 1 proved d < range and the bit set
  if (identity) {
#pragma unroll
    for (int q = 0; q < LR_ROWS; q++) r[q] = d[q];
  } else {
    uint64_t w[LR_ROWS]; uint32_t p[LR_ROWS];
#pragma unroll
    for (int q = 0; q < LR_ROWS; q++) { w[q] = bitmap[d[q] >> 6]; p[q] = prefix[d[q] >> 6]; }
#pragma unroll
    for (int q = 0; q < LR_ROWS; q++) r[q] = p[q] + (uint32_t)__popcll(w[q] & ((1ull << (d[q] & 63)) - 1ull));
    if constexpr (SEL) {
#pragma unroll
      for (int q = 0; q < LR_ROWS; q++) r[q] = sel_rows[r[q]];
    }
  }
#pragma unroll
  for (int q = 0; q < LR_ROWS; q++) { int64_t i = base + (int64_t)q * BLOCK; if (i < m) out_build[i] = r[q]; }
}
#define DFGPU_INT_KEY_DISPATCH(TYPE, CALL)                                                                      \
  switch (TYPE) {                                                                                               \
    case DFGPU_INT8: { using T = int8_t; CALL; break; } case DFGPU_INT16: { using T = int16_t; CALL; break; }   \
    case DFGPU_INT32: case DFGPU_DATE32: { using T = int32_t; CALL; break; } case DFGPU_INT64: { using T = int64_t; CALL; break; } \
    case DFGPU_UINT8: { using T = uint8_t; CALL; break; } case DFGPU_UINT16: { using T = uint16_t; CALL; break; } \
    case DFGPU_UINT32: { using T = uint32_t; CALL; break; } default: break; }
static bool int_key_type(int32_t t) { return t == DFGPU_INT8 || t == DFGPU_INT16 || t == DFGPU_INT32 || t == DFGPU_INT64 || t == DFGPU_DATE32 || t == DFGPU_UINT8 || t == DFGPU_UINT16 || t == DFGPU_UINT32; }

__global__ void k_mark_bits_u64idx(const uint64_t* idx, const uint64_t* idx_valid, int64_t n, uint64_t* bits, int64_t nbits, uint32_t* flags) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !valid_at(idx_valid, i)) return;
  uint64_t r = idx[i];
  if (r >= (uint64_t)nbits) { atomicOr(flags, DFGPU_FLAG_OOB); return; }
  atomicOr((unsigned long long*)&bits[r >> 6], 1ull << (r & 63));
}
__global__ void k_mark_bits_range(const uint32_t* idx, int64_t n, int64_t lo, int64_t hi, uint64_t* bits) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t r = idx[i];
  if (r >= lo && r < hi) atomicOr((unsigned long long*)&bits[(r - lo) >> 6], 1ull << ((r - lo) & 63));
}
// out = (a ^ flip) & (b or all ones)
__global__ void k_combine_words(const uint64_t* a, uint64_t flip, const uint64_t* b, uint64_t* out, int64_t nw) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nw) out[i] = (a[i] ^ flip) & (b ? b[i] : ~0ull);
}
__global__ void k_u32_to_u64(const uint32_t* in, uint64_t* out, int64_t n, uint64_t add) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint64_t)in[i] + add;
}
__global__ void k_add_u32(uint32_t* v, int64_t n, uint32_t add) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] += add;
}

// ---- key packing
struct PackCols { int32_t n; const void* v[MAX_KEYS]; const uint64_t* valid[MAX_KEYS]; int32_t type[MAX_KEYS]; int64_t mn[MAX_KEYS]; uint64_t range[MAX_KEYS], stride[MAX_KEYS]; };
__global__ void __launch_bounds__(BLOCK) k_cols_minmax(PackCols pc, int64_t n, long long* out /* [2c] min, [2c+1] max */) {
  for (int c = 0; c < MAX_KEYS; c++) {
    if (c >= pc.n) break;
    long long lo = INT64_MAX, hi = INT64_MIN;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
      if (valid_at(pc.valid[c], i)) { long long v = key_at(pc.v[c], pc.type[c], i); lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { long long a = __shfl_xor(lo, d, 64), b = __shfl_xor(hi, d, 64); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
    if (lane_id() == 0 && lo <= hi) { atomicMin(&out[2 * c], lo); atomicMax(&out[2 * c + 1], hi); }
  }
}
__global__ void __launch_bounds__(BLOCK) k_pack_keys(PackCols pc, int64_t n, int64_t* out, uint64_t* out_valid) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  bool ok = i < n; uint64_t p = 0;
#pragma unroll
  for (int c = 0; c < MAX_KEYS; c++) {
    if (c >= pc.n) break;
    if (ok) {
      if (!valid_at(pc.valid[c], i)) ok = false;
      else { uint64_t d = (uint64_t)(key_at(pc.v[c], pc.type[c], i) - pc.mn[c]); if (d >= pc.range[c]) ok = false; else p += d * pc.stride[c]; }
    }
  }
  if (i < n) out[i] = ok ? (int64_t)p : 0;
  if (out_valid) { uint64_t m = ballot64(ok); if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) out_valid[i >> 6] = m; }
}
static PackCols pack_cols(const dfgpu_join_table* t, const dfgpu_array* const* cols) {
  PackCols pc{}; pc.n = t->pack_n;
  for (int c = 0; c < t->pack_n; c++) { pc.v[c] = cols[c]->values->ptr; pc.valid[c] = cols[c]->validity ? (const uint64_t*)cols[c]->validity->ptr : nullptr; pc.type[c] = cols[c]->type;
                                        pc.mn[c] = t->pack_min[c]; pc.range[c] = t->pack_range[c]; pc.stride[c] = t->pack_stride[c]; }
  return pc;
}
// packed Int64 key column of `cols` (validity only when a tuple can be NULL / out of the build's ranges)
static dfgpu_array* pack_key_array(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* const* cols, bool need_valid) {
  int64_t n = cols[0]->length;
  ArrayHolder h(new_fixed(ctx, DFGPU_INT64, n, 0, 0, need_valid));
  if (n) { KernelTimer kt_(ctx, "k_pack_keys");
    hipLaunchKernelGGL(k_pack_keys, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, pack_cols(t, cols), n, (int64_t*)h.get()->values->ptr, need_valid ? (uint64_t*)h.get()->validity->ptr : nullptr);
    KERNEL_CHECK(); }
  h.get()->null_count = need_valid ? -1 : 0;
  return h.release();
}

static void check_key_types(const dfgpu_join_table* t, const dfgpu_array* const* pk, int32_t nkeys) {
  if (t->pack_n) {
    if (nkeys != t->pack_n) fail(DFGPU_INVALID_ARGUMENT, "probe has %d key columns, build has %d", nkeys, t->pack_n);
    for (int c = 0; c < nkeys; c++) if (logical_type(pk[c]) != t->pack_types[c]) fail(DFGPU_INVALID_ARGUMENT, "join key %d: build type %d vs probe type %d (the planner coerces first)", c, t->pack_types[c], logical_type(pk[c]));
    return;
  }
  if (nkeys != t->nkeys) fail(DFGPU_INVALID_ARGUMENT, "probe has %d key columns, build has %d", nkeys, t->nkeys);
  for (int c = 0; c < nkeys; c++)
    if (logical_type(pk[c]) != logical_type(t->keys[c])) fail(DFGPU_INVALID_ARGUMENT, "join key %d: build type %d vs probe type %d (the planner coerces first)", c, logical_type(t->keys[c]), logical_type(pk[c]));
}

// The general table: open addressing over all key types (and the CSR of repeated keys), plus the membership bitmap when the
// key is one integer column with a dense domain.
static void build_hash_table(dfgpu_ctx* ctx, dfgpu_join_table* t, bool with_bitmap) {
  int64_t n = t->n_build; const bool null_equals_null = t->null_equals_null;
  uint64_t cap = 64; int bits = 6; while (cap < (uint64_t)n * 2) { cap <<= 1; bits++; }
  if (cap > (1ull << 31)) fail(DFGPU_RESOURCES_EXHAUSTED, "build side of %lld rows exceeds the 2^30-row hash table limit", (long long)n);
  t->capacity = cap; t->cap_bits = bits;
  t->slots = alloc_buffer(ctx, cap * 8); HIP_CHECK(hipMemsetAsync(t->slots->ptr, 0xFF, cap * 8, ctx->stream));
  BufferPtr row_slot = alloc_buffer(ctx, (size_t)(n + 1) * 4);
  zero_scratch(ctx);
  if (n) { KernelTimer kt_(ctx, "k_join_build"); hipLaunchKernelGGL(k_join_build, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, t->ks, n,
                            t->build_mask ? (const uint64_t*)t->build_mask->ptr : nullptr, null_equals_null ? 1 : 0, ctx->force_hash_collisions ? 1 : 0,
                            (uint64_t*)t->slots->ptr, (uint32_t*)nullptr, cap - 1, (uint32_t*)row_slot->ptr, (unsigned long long*)ctx->d_scratch64); }
  KERNEL_CHECK();
  t->unique = read_scratch(ctx, 0) == 0;
  t->mem += (int64_t)(cap * 8);
  const dfgpu_array* key0 = t->keys[0];
  if (with_bitmap && n && t->nkeys == 1 && !null_equals_null && key0->type != DFGPU_DICTIONARY && int_key_type(key0->type)) {
    KernelTimer kt_(ctx, "join_build_bitmap");
    long long init[2] = { INT64_MAX, INT64_MIN };
    HIP_CHECK(hipMemcpyAsync(ctx->d_scratch64 + 4, init, 16, hipMemcpyHostToDevice, ctx->stream));
    const void* kv = key0->values->ptr; const uint32_t* rs = (const uint32_t*)row_slot->ptr;
    DFGPU_INT_KEY_DISPATCH(key0->type, hipLaunchKernelGGL((k_key_minmax<T>), dim3(grid_for(n, BLOCK * 8, ctx->num_cus * 2)), dim3(BLOCK), 0, ctx->stream, (const T*)kv, rs, n,
                                                          (long long*)(ctx->d_scratch64 + 4), (long long*)(ctx->d_scratch64 + 5)));
    KERNEL_CHECK();
    fetch_to_pinned(ctx, 4, ctx->d_scratch64 + 4, 16);
    long long lo = (long long)ctx->h_pinned[4], hi = (long long)ctx->h_pinned[5];
    if (lo <= hi) {
      uint64_t range = (uint64_t)hi - (uint64_t)lo + 1;
      if (range != 0 && range <= (1ull << 32) && range <= (uint64_t)n * 4096 + 65536) {      // <= 512 MB and not absurdly sparse
        t->bitmap = alloc_buffer(ctx, bitmap_bytes((int64_t)range), true); t->key_min = lo; t->range = range;
        DFGPU_INT_KEY_DISPATCH(key0->type, hipLaunchKernelGGL((k_key_setbits<T>), dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const T*)kv, rs, n, (int64_t)lo, (uint64_t*)t->bitmap->ptr));
        KERNEL_CHECK();
        t->mem += (int64_t)bitmap_bytes((int64_t)range);
      } else if (range != 0 && range <= (1ull << 32)) {     // too sparse to pay for up front; a probe batch large enough to amortise it builds it (join_probe)
        t->lazy_bitmap = true; t->key_min = lo; t->range = range;
        t->lazy_row_slot = alloc_buffer(ctx, (size_t)(n + 1) * 4);
        HIP_CHECK(hipMemcpyAsync(t->lazy_row_slot->ptr, row_slot->ptr, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
      }
    }
  }
  if (!t->unique) {
    t->slot_count = alloc_buffer(ctx, cap * 4, true);
    hipLaunchKernelGGL(k_count_slots, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)row_slot->ptr, n, (uint32_t*)t->slot_count->ptr);
    // CSR of build rows per key group: stable radix sort of (slot, row) then exclusive scan of group sizes
    hipLaunchKernelGGL(k_fix_unslotted, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (uint32_t*)row_slot->ptr, n, (uint32_t)cap);
    BufferPtr rows = alloc_buffer(ctx, (size_t)n * 4);
    launch_iota_u32(ctx, (uint32_t*)rows->ptr, n, 0);
    radix_sort_pairs_u32(ctx, (uint32_t*)row_slot->ptr, (uint32_t*)rows->ptr, n, bits + 1);
    t->csr_rows = rows;
    t->slot_start = alloc_buffer(ctx, cap * 4);
    HIP_CHECK(hipMemcpyAsync(t->slot_start->ptr, t->slot_count->ptr, cap * 4, hipMemcpyDeviceToDevice, ctx->stream));
    exclusive_scan_u32_inplace32(ctx, (uint32_t*)t->slot_start->ptr, (int64_t)cap, nullptr);
    t->mem += (int64_t)(cap * 8 + (size_t)n * 4);
  }
}

std::shared_ptr<const OrderStats> order_stats_measure(dfgpu_ctx* ctx, const dfgpu_array* a) {
  if (!a || a->type == DFGPU_DICTIONARY || !int_key_type(a->type) || a->validity || a->length < 1) return nullptr;
  const int64_t n = a->length; const void* kv = a->values->ptr;
  zero_scratch(ctx);
  DFGPU_INT_KEY_DISPATCH(a->type, hipLaunchKernelGGL((k_check_increasing<T>), dim3(grid_for(n, BLOCK * 8, ctx->num_cus * 8)), dim3(BLOCK), 0, ctx->stream, (const T*)kv, n, (unsigned long long*)ctx->d_scratch64));
  KERNEL_CHECK();
  ctx->count_sync("sync:rank_index_check"); fetch_to_pinned(ctx, 0, ctx->d_scratch64, 32);
  OrderStats st; st.sorted = ctx->h_pinned[0] == 0; st.repeats = ctx->h_pinned[3] != 0; st.exact = true; st.lo = (int64_t)ctx->h_pinned[1]; st.hi = (int64_t)ctx->h_pinned[2];       // sign/zero-extended by the kernel
  order_stats_set(a, st);
  return order_stats_get(a);
}
// Rank index: taken when the single integer key column is strictly increasing (checked on the device, one streaming pass)
// and its domain is dense enough for a bitmap.  Returns false (nothing built) otherwise.
static bool build_rank_index(dfgpu_ctx* ctx, dfgpu_join_table* t) {
  int64_t n = t->n_build;
  if (!ctx->join_rank_index || ctx->force_hash_collisions || n < 2 || t->nkeys != 1 || t->null_equals_null) return false;
  const dfgpu_array* key0 = t->keys[0];
  if (key0->type == DFGPU_DICTIONARY || !int_key_type(key0->type) || key0->validity) return false;
  const void* kv = key0->values->ptr;
  KernelTimer kt_(ctx, "join_build_rank");
  // the column's order statistics: its memo (a base-table key column after its first build), a derivation (a sorted column gathered through ascending row numbers: bounds
  // of the source, which may be wide), or the pass itself
  auto st = order_stats_get(key0);
  if (!st) st = order_stats_measure(ctx, key0);
  auto dense = [&](const OrderStats& s2) { const uint64_t r = (uint64_t)s2.hi - (uint64_t)s2.lo + 1; return r != 0 && r <= (1ull << 32) && r <= (uint64_t)n * 4096 + 65536; };
  if (st && st->sorted && !st->exact && !dense(*st)) st = order_stats_measure(ctx, key0);      // the inherited bounds are too wide for a bitmap: the exact ones may not be
  if (!st || !st->sorted) return false;
  const bool runs = st->repeats;                           // sorted with repeats (an inherited "repeats" may overstate: the run path answers unique keys too)
  if (runs && (t->build_mask || n > 0xFFFFFFF0ll)) return false;
  long long lo = (long long)st->lo, hi = (long long)st->hi;
  uint64_t range = (uint64_t)hi - (uint64_t)lo + 1;
  if (!dense(*st)) return false;
  const uint64_t* mk = t->build_mask ? (const uint64_t*)t->build_mask->ptr : nullptr;
  int64_t nw = (int64_t)((range + 63) / 64);
  t->key_min = lo; t->range = range;
  t->rank_mode = true; t->unique = !runs; t->rank_runs = runs; t->rank_identity = !runs && range == (uint64_t)n;
  BufferPtr heads;
  if (runs) {             // one bit per run head: it is the selection for the key bitmap, and its indices are the run starts
    heads = alloc_buffer(ctx, bitmap_bytes(n));
    DFGPU_INT_KEY_DISPATCH(key0->type, hipLaunchKernelGGL((k_key_run_heads<T>), dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const T*)kv, n, (uint64_t*)heads->ptr));
    KERNEL_CHECK();
    mk = (const uint64_t*)heads->ptr;
  }
  if (t->rank_identity && mk) t->bitmap = t->build_mask;       // key = key_min + row: the build selection IS the membership bitmap
  else {
    t->bitmap = alloc_buffer(ctx, bitmap_bytes((int64_t)range)); t->mem += (int64_t)bitmap_bytes((int64_t)range);
    HIP_CHECK(hipMemsetAsync(t->bitmap->ptr, t->rank_identity ? 0xFF : 0, bitmap_bytes((int64_t)range), ctx->stream));   // bits >= range are never read as set: probes test d < range
  }
  if (!t->rank_identity) {
    // masked build: rank -> build row; runs: rank -> first row of the run.  A plain masked build only ever reads entries below the number of set bits (a rank), so the table
    // is written without the host learning that number (one read-back less per build; the count stays in a device word); the run path walks to sel_rows[r + 1] and needs the exact length.
    if (mk && !runs && n <= 0xFFFFFFF0ll) {
      t->sel_rows = mask_to_indices_uncounted(ctx, mk, n, ctx->d_scratch64 + 15); t->mem += t->sel_rows->length * 4;
      DFGPU_INT_KEY_DISPATCH(key0->type, hipLaunchKernelGGL((k_key_setbits_rows<T>), dim3(grid_for(n, BLOCK, ctx->num_cus * 16)), dim3(BLOCK), 0, ctx->stream, (const T*)kv, (const uint32_t*)t->sel_rows->values->ptr,
                                                            (const unsigned long long*)(ctx->d_scratch64 + 15), (int64_t)lo, (uint64_t*)t->bitmap->ptr));
    } else {
      DFGPU_INT_KEY_DISPATCH(key0->type, hipLaunchKernelGGL((k_key_setbits_masked<T>), dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const T*)kv, mk, n, (int64_t)lo, (uint64_t*)t->bitmap->ptr));
      if (mk) { t->sel_rows = mask_to_indices_impl(ctx, mk, n); t->mem += t->sel_rows->length * 4; }
    }
    KERNEL_CHECK();
    t->rank_prefix = alloc_buffer(ctx, (size_t)nw * 4);
    hipLaunchKernelGGL(k_popc_words, dim3(grid_for(nw, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)t->bitmap->ptr, nw, (uint32_t*)t->rank_prefix->ptr);
    exclusive_scan_u32_inplace32(ctx, (uint32_t*)t->rank_prefix->ptr, nw, nullptr);
    KERNEL_CHECK();
    t->mem += nw * 4;
  }
  return true;
}

// The partitioned join (pjoin.hip) is for key domains the membership bitmap cannot prefilter: min / max of the selected build keys in one
// streaming pass.  true = range beyond 256 x rows (or beyond 2^32): a bitmap over it would be mostly empty lines.
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_key_minmax_masked(const T* keys, const uint64_t* valid, const uint64_t* mask, int64_t n, long long* mn, long long* mx, const uint32_t* rows = nullptr, const unsigned long long* d_count = nullptr) {
  long long lo = INT64_MAX, hi = INT64_MIN;
  if (rows) {            // the selected rows as a list (length on the device)
    const int64_t m = (int64_t)*d_count;
    for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < m; j += (int64_t)gridDim.x * BLOCK) { const int64_t i = rows[j]; if (valid_at(valid, i)) { long long v = (long long)keys[i]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; } }
  } else
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
    if (row_selected(mask, i) && valid_at(valid, i)) { long long v = (long long)keys[i]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { long long a = __shfl_xor(lo, d, 64), b = __shfl_xor(hi, d, 64); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
  __shared__ long long slo[BLOCK / WAVE], shi[BLOCK / WAVE];
  if (lane_id() == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < BLOCK / WAVE; w++) { lo = slo[w] < lo ? slo[w] : lo; hi = shi[w] > hi ? shi[w] : hi; }
    if (lo <= hi) { atomicMin(mn, lo); atomicMax(mx, hi); }
  }
}
// the selected rows' keys gathered once into a compact column (the passes that follow read it in order) and their min / max in the same pass
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_key_gather_minmax(const T* keys, const uint32_t* rows, const unsigned long long* d_count, T* ckeys, long long* mn, long long* mx) {
  const int64_t m = (int64_t)*d_count; long long lo = INT64_MAX, hi = INT64_MIN;
  // four gathers in flight per lane; the grid is at most 4 workgroups per CU because every workgroup that saw a row ends with two atomics on ONE address each (4096 workgroups:
  // 8192 serialised atomics = 105 us for 1.8 M rows, profiles/r04_shuffled_timeline_sf12.5.txt)
  for (int64_t j0 = ((int64_t)blockIdx.x * BLOCK) * 4 + threadIdx.x; j0 < m; j0 += (int64_t)gridDim.x * BLOCK * 4) {
    T k[4]; bool on[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const int64_t j = j0 + (int64_t)q * BLOCK; on[q] = j < m; k[q] = on[q] ? keys[rows[j]] : (T)0; }
#pragma unroll
    for (int q = 0; q < 4; q++) if (on[q]) { ckeys[j0 + (int64_t)q * BLOCK] = k[q]; const long long v = (long long)k[q]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { long long a = __shfl_xor(lo, d, 64), b = __shfl_xor(hi, d, 64); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
  __shared__ long long slo[BLOCK / WAVE], shi[BLOCK / WAVE];
  if (lane_id() == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < BLOCK / WAVE; w++) { lo = slo[w] < lo ? slo[w] : lo; hi = shi[w] > hi ? shi[w] : hi; }
    if (lo <= hi) { atomicMin(mn, lo); atomicMax(mx, hi); }
  }
}
// Rank index over unsorted unique keys (see k_key_setbits_unique).  Tried after build_rank_index found the keys unsorted: min / max of the selected keys, a domain of at most
// 256 slots per key (bitmap + prefix <= 48 bytes per build row; an eighth of a hash-partitioned TPC-H key column is 1 in 64), the bits set with repeat detection, then rank -> row.  false = not taken, nothing kept.
static bool build_rank_index_unsorted(dfgpu_ctx* ctx, dfgpu_join_table* t) {
  const int64_t n = t->n_build;
  if (!ctx->join_rank_index || !ctx->join_rank_index_unsorted || ctx->force_hash_collisions || n < 2 || n > 0xFFFFFFF0ll || t->nkeys != 1 || t->null_equals_null) return false;
  const dfgpu_array* key0 = t->keys[0];
  if (key0->type == DFGPU_DICTIONARY || !int_key_type(key0->type) || key0->validity) return false;
  KernelTimer kt_(ctx, "join_build_rank");
  const uint64_t* mk = t->build_mask ? (const uint64_t*)t->build_mask->ptr : nullptr;
  // a masked build (a FilterExec or a join's selection fused into it) first turns the mask into the list of its rows -- the count stays on the device -- and every pass below
  // walks that list: the work follows the selected rows, not the column (TPC-H Q3 behind a shuffle: 15 M of 150 M orders)
  ArrayHolder list; const uint32_t* rl = nullptr; const unsigned long long* dcount = (const unsigned long long*)(ctx->d_scratch64 + 15);
  if (mk) { list.a = mask_to_indices_uncounted(ctx, mk, n, ctx->d_scratch64 + 15); rl = (const uint32_t*)list.get()->values->ptr; }
  const int lgrid = grid_for(n, BLOCK, ctx->num_cus * 16);
  long long init[2] = { INT64_MAX, INT64_MIN };
  HIP_CHECK(hipMemcpyAsync(ctx->d_scratch64 + 4, init, 16, hipMemcpyHostToDevice, ctx->stream));
  // masked: the selected keys are gathered ONCE (with their min / max) into a compact column; setting the bits and ranking the rows then read it in order instead of
  // gathering the same scattered 8 bytes two more times (three random passes over 15 M of 150 M orders cost 0.8 ms, one costs 0.27)
  BufferPtr ck; const void* ckp = nullptr;
  if (rl) {
    ck = alloc_buffer(ctx, (size_t)n * type_width(key0->type) + 16); ckp = ck->ptr;
    DFGPU_INT_KEY_DISPATCH(key0->type, hipLaunchKernelGGL((k_key_gather_minmax<T>), dim3(grid_for(n, BLOCK * 4, ctx->num_cus * 4)), dim3(BLOCK), 0, ctx->stream, (const T*)key0->values->ptr, rl, dcount, (T*)ck->ptr,
                                                          (long long*)(ctx->d_scratch64 + 4), (long long*)(ctx->d_scratch64 + 5)));
  } else
  DFGPU_INT_KEY_DISPATCH(key0->type, hipLaunchKernelGGL((k_key_minmax_masked<T>), dim3(grid_for(n, BLOCK * 8, ctx->num_cus * 4)), dim3(BLOCK), 0, ctx->stream, (const T*)key0->values->ptr,
                                                        (const uint64_t*)nullptr, mk, n, (long long*)(ctx->d_scratch64 + 4), (long long*)(ctx->d_scratch64 + 5), rl, dcount));
  KERNEL_CHECK();
  ctx->count_sync("sync:rank_index_range"); fetch_to_pinned(ctx, 4, ctx->d_scratch64 + 4, 16);
  const long long lo = (long long)ctx->h_pinned[4], hi = (long long)ctx->h_pinned[5];
  t->have_minmax = true; t->sel_min = lo; t->sel_max = hi;
  if (lo > hi) return false;                               // no selected row
  const uint64_t range = (uint64_t)hi - (uint64_t)lo + 1;
  if (range == 0 || range > (1ull << 32) || range > (uint64_t)n * 256 + 65536) return false;       // sparser domains are the partitioned join's (pj_domain_is_sparse draws the same line)
  const int64_t nw = (int64_t)((range + 63) / 64);
  BufferPtr bitmap = alloc_buffer(ctx, bitmap_bytes((int64_t)range));
  HIP_CHECK(hipMemsetAsync(bitmap->ptr, 0, bitmap_bytes((int64_t)range), ctx->stream));
  HIP_CHECK(hipMemsetAsync(ctx->d_scratch64 + 6, 0, 8, ctx->stream));
  DFGPU_INT_KEY_DISPATCH(key0->type, hipLaunchKernelGGL((k_key_setbits_unique<T>), dim3(lgrid), dim3(BLOCK), 0, ctx->stream, (const T*)key0->values->ptr, mk, rl, dcount, n, (int64_t)lo, (uint64_t*)bitmap->ptr,
                                                        (unsigned long long*)(ctx->d_scratch64 + 6), (const T*)ckp));
  BufferPtr prefix = alloc_buffer(ctx, (size_t)nw * 4);
  hipLaunchKernelGGL(k_popc_words, dim3(grid_for(nw, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)bitmap->ptr, nw, (uint32_t*)prefix->ptr);
  exclusive_scan_u32_inplace32(ctx, (uint32_t*)prefix->ptr, nw, ctx->d_scratch64 + 7);
  KERNEL_CHECK();
  const uint64_t* hsc = read_scratch_range(ctx, 6, 2);
  ctx->count_sync("sync:rank_index_unique");
  if (hsc[0] != 0) return false;                           // a key repeats: the hash paths keep their CSR
  const int64_t nsel = (int64_t)hsc[1];
  ArrayHolder rows(new_fixed(ctx, DFGPU_UINT32, nsel));
  DFGPU_INT_KEY_DISPATCH(key0->type, hipLaunchKernelGGL((k_rank_rows<T>), dim3(lgrid), dim3(BLOCK), 0, ctx->stream, (const T*)key0->values->ptr, mk, rl, dcount, n, (int64_t)lo, (const uint64_t*)bitmap->ptr,
                                                        (const uint32_t*)prefix->ptr, (uint32_t*)rows.get()->values->ptr, (const T*)ckp));
  KERNEL_CHECK();
  t->key_min = lo; t->range = range; t->rank_mode = true; t->unique = true; t->rank_runs = false; t->rank_identity = false;
  t->bitmap = bitmap; t->rank_prefix = prefix; t->sel_rows = rows.release();
  t->mem += (int64_t)bitmap_bytes((int64_t)range) + nw * 4 + nsel * 4;
  return true;
}
static bool pj_domain_is_sparse(dfgpu_ctx* ctx, dfgpu_join_table* t) {
  int64_t n = t->n_build;
  if (!ctx->join_partitioned || ctx->force_hash_collisions || t->nkeys != 1 || t->null_equals_null || n < ctx->join_partitioned_min_build) return false;
  const dfgpu_array* key0 = t->keys[0];
  if (key0->type == DFGPU_UINT64) return true;
  if (key0->type == DFGPU_DICTIONARY || !int_key_type(key0->type)) return false;
  if (t->have_minmax && !key0->validity) {                 // build_rank_index_unsorted has been here (it takes columns without NULLs only)
    if (t->sel_min > t->sel_max) return false;
    const uint64_t range = (uint64_t)t->sel_max - (uint64_t)t->sel_min + 1;
    return range == 0 || range > (1ull << 32) || range > (uint64_t)n * 256;
  }
  long long init[2] = { INT64_MAX, INT64_MIN };
  HIP_CHECK(hipMemcpyAsync(ctx->d_scratch64 + 4, init, 16, hipMemcpyHostToDevice, ctx->stream));
  DFGPU_INT_KEY_DISPATCH(key0->type, hipLaunchKernelGGL((k_key_minmax_masked<T>), dim3(grid_for(n, BLOCK * 8, ctx->num_cus * 4)), dim3(BLOCK), 0, ctx->stream, (const T*)key0->values->ptr,
                                                        key0->validity ? (const uint64_t*)key0->validity->ptr : nullptr, t->build_mask ? (const uint64_t*)t->build_mask->ptr : nullptr, n,
                                                        (long long*)(ctx->d_scratch64 + 4), (long long*)(ctx->d_scratch64 + 5)));
  KERNEL_CHECK();
  ctx->count_sync("sync:pj_key_range"); fetch_to_pinned(ctx, 4, ctx->d_scratch64 + 4, 16);
  long long lo = (long long)ctx->h_pinned[4], hi = (long long)ctx->h_pinned[5];
  if (lo > hi) return false;
  uint64_t range = (uint64_t)hi - (uint64_t)lo + 1;
  return range == 0 || range > (1ull << 32) || range > (uint64_t)n * 256;
}

}  // namespace dfgpu

// ---- NestedLoopJoinExec: the candidate pairs of left rows [first, first + count) x every right row, left-major
// (build_join_indices, joins/nested_loop_join.rs:405-432: left = [i, i, .., i], right = [0, 1, .., n_right) for every left row i in order)
template <typename L, typename R>
__global__ void __launch_bounds__(BLOCK) k_cross_indices(int64_t first, int64_t n_right, int64_t total, L* __restrict__ left, R* __restrict__ right) {
  int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (k >= total) return;
  int64_t i = k / n_right; left[k] = (L)(first + i); right[k] = (R)(k - i * n_right);
}

extern "C" {

dfgpu_status dfgpu_join_build(dfgpu_ctx* ctx, const dfgpu_array* const* keys, int32_t nkeys, const dfgpu_array* opt_mask,
                              int32_t null_equals_null, dfgpu_join_table** out) {
  return guard(ctx, [&] {
    if (!keys || !out) fail(DFGPU_INVALID_ARGUMENT, "join_build: null argument");
    std::unique_ptr<dfgpu_join_table> t(new dfgpu_join_table());
    t->ctx = ctx; t->nkeys = nkeys; t->null_equals_null = null_equals_null != 0;
    int64_t n = keys[0]->length; t->n_build = n;
    ArrayHolder packed;
    bool packable = ctx->join_key_packing && nkeys >= 2 && nkeys <= 4 && !null_equals_null && !ctx->force_hash_collisions && n > 0;
    for (int c = 0; c < nkeys && packable; c++) packable = keys[c]->type != DFGPU_DICTIONARY && int_key_type(keys[c]->type) && keys[c]->length == n;
    if (packable) {
      t->pack_n = nkeys; for (int c = 0; c < nkeys; c++) t->pack_types[c] = logical_type(keys[c]);
      std::vector<long long> init((size_t)2 * nkeys); for (int c = 0; c < nkeys; c++) { init[(size_t)2 * c] = INT64_MAX; init[(size_t)2 * c + 1] = INT64_MIN; }
      HIP_CHECK(hipMemcpyAsync(ctx->d_scratch64 + 16, init.data(), init.size() * 8, hipMemcpyHostToDevice, ctx->stream));
      PackCols pc = pack_cols(t.get(), keys);
      hipLaunchKernelGGL(k_cols_minmax, dim3(grid_for(n, BLOCK * 8, ctx->num_cus * 4)), dim3(BLOCK), 0, ctx->stream, pc, n, (long long*)(ctx->d_scratch64 + 16));
      KERNEL_CHECK();
      ctx->count_sync("sync:key_packing_ranges"); fetch_to_pinned(ctx, 16, ctx->d_scratch64 + 16, init.size() * 8);
      unsigned __int128 prod = 1; bool any_valid = true, nullable = false;
      for (int c = nkeys - 1; c >= 0; c--) {
        long long lo = (long long)ctx->h_pinned[16 + 2 * c], hi = (long long)ctx->h_pinned[16 + 2 * c + 1];
        if (lo > hi) { any_valid = false; lo = hi = 0; }                       // a column of NULLs only: nothing can match
        t->pack_min[c] = lo; t->pack_range[c] = (uint64_t)hi - (uint64_t)lo + 1; t->pack_stride[c] = (uint64_t)prod;
        if (t->pack_range[c] == 0) { prod = (unsigned __int128)1 << 100; break; }
        prod *= t->pack_range[c]; nullable = nullable || keys[c]->validity != nullptr;
        if (prod > ((unsigned __int128)1 << 40)) break;
      }
      if (prod <= ((unsigned __int128)1 << 40)) {
        (void)any_valid;
        packed.a = pack_key_array(ctx, t.get(), keys, nullable);
        const dfgpu_array* pk1 = packed.get(); keys = &pk1; nkeys = 1; t->nkeys = 1;
        t->ks = make_keyset(keys, 1);
        t->keys.push_back(packed.release());                                   // the table owns the packed column
      } else t->pack_n = 0;
    }
    if (!t->pack_n) {
      t->ks = make_keyset(keys, nkeys);
      for (int c = 0; c < nkeys; c++) { t->keys.push_back(const_cast<dfgpu_array*>(keys[c])); dfgpu_array_retain(t->keys.back()); }
    }
    t->build_mask = effective_mask(ctx, opt_mask, n);
    t->mem = (int64_t)bitmap_bytes(n);               // the visited bitmap: accounted here, allocated and cleared when a join type that marks rows first does (an Inner join never)
    // order of preference: rank index (clustered keys: no table at all), radix-partitioned LDS tables (large builds on unsorted keys
    // whose domain is too sparse for the membership bitmap in front of the general table), general open-addressing table
    if (!build_rank_index(ctx, t.get()) && !build_rank_index_unsorted(ctx, t.get()) && !((pj_domain_is_sparse(ctx, t.get()) || pj_hashed_candidate(ctx, t.get())) && pj_build(ctx, t.get()))) build_hash_table(ctx, t.get(), true);
    *out = t.release();
  });
}
void dfgpu_join_table_free(dfgpu_join_table* t) { delete t; }
int64_t dfgpu_join_table_num_rows(const dfgpu_join_table* t) { return t ? t->n_build : 0; }
int64_t dfgpu_join_table_memory(const dfgpu_join_table* t) { return t ? t->mem : 0; }

static dfgpu_status join_probe_impl(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* const* probe_keys, int32_t nkeys,
                                    const dfgpu_array* opt_mask, dfgpu_array** out_build_idx, dfgpu_array** out_probe_idx, bool may_defer, dfgpu_array** out_selection);
dfgpu_status dfgpu_join_probe(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* const* probe_keys, int32_t nkeys,
                              const dfgpu_array* opt_mask, dfgpu_array** out_build_idx, dfgpu_array** out_probe_idx) {
  return join_probe_impl(ctx, t, probe_keys, nkeys, opt_mask, out_build_idx, out_probe_idx, false, nullptr);
}
/* see include/dfgpu.h */
dfgpu_status dfgpu_join_probe_deferred(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* const* probe_keys, int32_t nkeys,
                                       const dfgpu_array* opt_mask, dfgpu_array** out_build_idx, dfgpu_array** out_probe_idx) {
  return join_probe_impl(ctx, t, probe_keys, nkeys, opt_mask, out_build_idx, out_probe_idx, true, nullptr);
}
dfgpu_status dfgpu_join_probe_selection(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* const* probe_keys, int32_t nkeys, const dfgpu_array* opt_mask, dfgpu_array** out_selection) {
  if (!out_selection) return DFGPU_INVALID_ARGUMENT;
  return join_probe_impl(ctx, t, probe_keys, nkeys, opt_mask, nullptr, nullptr, false, out_selection);
}
dfgpu_status dfgpu_join_lookup(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* const* probe_keys, int32_t nkeys, const dfgpu_array* rows, dfgpu_array** out_build_idx) {
  return guard(ctx, [&] {
    if (!t || !probe_keys || !out_build_idx) fail(DFGPU_INVALID_ARGUMENT, "join_lookup: null argument");
    if (!(t->rank_mode && t->unique && !t->rank_runs && t->bitmap) || nkeys != 1 || probe_keys[0]->type != t->keys[0]->type || probe_keys[0]->validity)
      fail(DFGPU_INVALID_ARGUMENT, "join_lookup: only after dfgpu_join_probe_deferred left the build indices out (a unique rank-indexed build, one integer key column without NULLs)");
    if (rows && rows->type != DFGPU_UINT32) fail(DFGPU_INVALID_ARGUMENT, "join_lookup: rows must be UInt32");
    const dfgpu_array* pk = probe_keys[0];
    const int64_t m = rows ? rows->length : pk->length;
    ArrayHolder ob(new_fixed(ctx, DFGPU_UINT64, m));
    // NULL entries of `rows` (the NULL side of an outer join's indices above a deferred join) stay NULL in the answer: take() through it then yields NULL rows
    const uint64_t* rv = rows && rows->validity ? (const uint64_t*)rows->validity->ptr : nullptr;
    if (m) { KernelTimer kt_(ctx, "k_probe_lookup_rank");
      const bool all = !rv && (rows == nullptr || (rows->identity && rows->length == pk->length)), sel = t->sel_rows != nullptr;
      const uint32_t* rp = rows ? (const uint32_t*)rows->values->ptr : nullptr;
#define LR(ALL, SEL) DFGPU_INT_KEY_DISPATCH(pk->type, hipLaunchKernelGGL((k_probe_lookup_rank<T, ALL, SEL>), dim3(grid_for(m, BLOCK * LR_ROWS)), dim3(BLOCK), 0, ctx->stream, (const T*)pk->values->ptr, rp, m, t->key_min, \
                                                            (const uint64_t*)t->bitmap->ptr, t->rank_prefix ? (const uint32_t*)t->rank_prefix->ptr : nullptr, \
                                                            t->sel_rows ? (const uint32_t*)t->sel_rows->values->ptr : nullptr, t->rank_identity ? 1 : 0, (uint64_t*)ob.get()->values->ptr, rv, pk->length))
      if (all && sel) { LR(true, true); } else if (all) { LR(true, false); } else if (sel) { LR(false, true); } else { LR(false, false); }
#undef LR
      KERNEL_CHECK(); }
    if (rv) { ob.get()->validity = rows->validity; ob.get()->null_count = rows->null_count; }
    *out_build_idx = ob.release();
  });
}
static dfgpu_status join_probe_impl(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* const* probe_keys, int32_t nkeys,
                                    const dfgpu_array* opt_mask, dfgpu_array** out_build_idx, dfgpu_array** out_probe_idx, bool may_defer, dfgpu_array** out_selection) {
  return guard(ctx, [&] {
    if (!t || !probe_keys || (!out_selection && (!out_build_idx || !out_probe_idx))) fail(DFGPU_INVALID_ARGUMENT, "join_probe: null argument");
    check_key_types(t, probe_keys, nkeys);
    // the selection form (dfgpu_join_probe_selection): only for the tables whose probe would be deferred -- decided before any pass runs
    if (out_selection && !(t->rank_mode && t->unique && !t->rank_runs && t->bitmap && !t->pack_n && nkeys == 1 && probe_keys[0]->type == t->keys[0]->type && !probe_keys[0]->validity))
      fail(DFGPU_NOT_IMPLEMENTED, "join_probe_selection: only a unique rank-indexed build probed by one integer key column of the build's type without NULLs");
    ArrayHolder packed_probe; const dfgpu_array* pk1 = nullptr;
    if (t->pack_n) {
      for (int c = 0; c < nkeys; c++) if (probe_keys[c]->type == DFGPU_DICTIONARY || probe_keys[c]->length != probe_keys[0]->length) fail(DFGPU_NOT_IMPLEMENTED, "probe of a packed multi-key table with dictionary-encoded keys");
      packed_probe.a = pack_key_array(ctx, t, probe_keys, true);
      pk1 = packed_probe.get(); probe_keys = &pk1; nkeys = 1;
    }
    KeySet pks = make_keyset(probe_keys, nkeys);
    int64_t n = probe_keys[0]->length;
    BufferPtr mask = effective_mask(ctx, opt_mask, n);
    int64_t nw = (n + 63) / 64;
    const uint64_t* mk = mask ? (const uint64_t*)mask->ptr : nullptr;
    int nen = t->null_equals_null ? 1 : 0, fz = ctx->force_hash_collisions ? 1 : 0;
    if (!out_selection && pj_probe_eligible(ctx, t, probe_keys, nkeys, n)) {        // large batch against a partitioned build: partition by partition out of LDS
      pj_probe(ctx, t, probe_keys, nkeys, mk, out_build_idx, out_probe_idx);
      check_flags(ctx, "join_probe");
      return;
    }
    // pass 1: match bit per probe row
    BufferPtr match_bits = alloc_buffer(ctx, bitmap_bytes(n), n == 0);
    bool use_bitmap = false, fast_hash = false; BufferPtr found_slot;
    if (n) {
      const dfgpu_array* pk = probe_keys[0];
      if (!t->bitmap && t->lazy_bitmap && pk->type == t->keys[0]->type && t->range <= (uint64_t)n * 16) {
        auto* mt = const_cast<dfgpu_join_table*>(t); const dfgpu_array* key0 = t->keys[0];
        KernelTimer kt_(ctx, "join_build_bitmap");
        mt->bitmap = alloc_buffer(ctx, bitmap_bytes((int64_t)t->range), true); mt->mem += (int64_t)bitmap_bytes((int64_t)t->range);
        DFGPU_INT_KEY_DISPATCH(key0->type, hipLaunchKernelGGL((k_key_setbits<T>), dim3(grid_for(t->n_build, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const T*)key0->values->ptr,
                                                              (const uint32_t*)t->lazy_row_slot->ptr, t->n_build, t->key_min, (uint64_t*)mt->bitmap->ptr));
        KERNEL_CHECK();
        mt->lazy_bitmap = false; mt->lazy_row_slot.reset();
      }
      use_bitmap = t->bitmap && pk->type == t->keys[0]->type;      // same physical integer type, no dictionary
      if (!use_bitmap && !t->slots) build_hash_table(ctx, const_cast<dfgpu_join_table*>(t), false);   // rank index cannot serve this probe column
      if (use_bitmap && bp_probe(ctx, t, pk, mk, n, (uint64_t*)match_bits->ptr)) {
        // unclustered keys: probed partition by partition of the key range (pjoin.hip)
      } else if (use_bitmap) {
        KernelTimer kt_(ctx, "k_probe_match_bitmap");
        int64_t rows_per_block = (int64_t)BLOCK * PM_ROWS;
        const uint64_t* kvp = pk->validity ? (const uint64_t*)pk->validity->ptr : nullptr;
        // the kernel reads two keys per load (2 x sizeof(T) aligned): a zero-copy slice at an odd row offset (dfgpu_array_slice) is copied once
        const void* kptr = pk->values->ptr; BufferPtr aligned_keys; const size_t kw = (size_t)type_width(pk->type);
        if ((uintptr_t)kptr % (2 * kw)) { aligned_keys = alloc_buffer(ctx, (size_t)n * kw); HIP_CHECK(hipMemcpyAsync(aligned_keys->ptr, kptr, (size_t)n * kw, hipMemcpyDeviceToDevice, ctx->stream)); kptr = aligned_keys->ptr; }
#define PM_LAUNCH(HM, HV) DFGPU_INT_KEY_DISPATCH(pk->type, hipLaunchKernelGGL((k_probe_match_bitmap<T, HM, HV>), dim3(grid_for(n, (int)rows_per_block)), dim3(BLOCK), 0, ctx->stream, (const T*)kptr, \
                                                            kvp, mk, n, t->key_min, t->range, (const uint64_t*)t->bitmap->ptr, (uint64_t*)match_bits->ptr))
        if (mk && kvp) { PM_LAUNCH(true, true); } else if (mk) { PM_LAUNCH(true, false); } else if (kvp) { PM_LAUNCH(false, true); } else { PM_LAUNCH(false, false); }
#undef PM_LAUNCH
      } else {
        KernelTimer kt_(ctx, "k_probe_match_hash");
        auto plain8 = [](const dfgpu_array* a) { return (a->type == DFGPU_INT64 || a->type == DFGPU_UINT64) && !a->validity; };
        fast_hash = (nkeys == 1 || nkeys == 2) && !nen && !fz && t->n_build > 0;
        for (int c = 0; c < nkeys && fast_hash; c++) fast_hash = plain8(probe_keys[c]) && plain8(t->keys[(size_t)c]);
        if (fast_hash) {
          found_slot = alloc_buffer(ctx, (size_t)n * 4);
          const uint64_t *b0 = (const uint64_t*)t->keys[0]->values->ptr, *b1 = nkeys == 2 ? (const uint64_t*)t->keys[1]->values->ptr : nullptr;
          const uint64_t *p0 = (const uint64_t*)probe_keys[0]->values->ptr, *p1 = nkeys == 2 ? (const uint64_t*)probe_keys[1]->values->ptr : nullptr;
          dim3 hgrid(grid_for(n, BLOCK * HP_ROWS));
#define HP(NK, HM) hipLaunchKernelGGL((k_probe_hash_i64<NK, HM>), hgrid, dim3(BLOCK), 0, ctx->stream, b0, b1, p0, p1, n, mk, (const uint64_t*)t->slots->ptr, t->capacity - 1, (uint64_t*)match_bits->ptr, (uint32_t*)found_slot->ptr)
          if (nkeys == 1) { if (mk) HP(1, true); else HP(1, false); } else { if (mk) HP(2, true); else HP(2, false); }
#undef HP
        } else
        hipLaunchKernelGGL(k_probe_match_hash, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, t->ks, pks, n, mk, nen, fz, (const uint64_t*)t->slots->ptr, t->capacity - 1, (uint64_t*)match_bits->ptr);
      }
      KERNEL_CHECK();
    }
    (void)nw;
    if (out_selection) {            // the match bits ARE the answer: a Boolean column over the probe rows (bits of a partial last word beyond n are zero)
      ArrayHolder sel(new_array(ctx, DFGPU_BOOL, n)); sel.get()->values = match_bits; sel.get()->null_count = 0;
      check_flags(ctx, "join_probe");
      *out_selection = sel.release();
      return;
    }
    ArrayHolder rows(mask_to_indices_impl(ctx, (const uint64_t*)match_bits->ptr, n));      // matched probe rows, ascending
    int64_t m = rows.get()->length;
    const uint32_t* rp = (const uint32_t*)rows.get()->values->ptr;
    ArrayHolder ob, op;
    if (t->rank_mode && use_bitmap && t->rank_runs) {          // sorted build key with repeats: every match emits its contiguous run
      const dfgpu_array* pk = probe_keys[0];
      BufferPtr run_of = alloc_buffer(ctx, (size_t)(m + 1) * 4), cnt = alloc_buffer(ctx, (size_t)(m + 1) * 4), offs = alloc_buffer(ctx, (size_t)(m + 1) * 8);
      int64_t total = 0;
      if (m) {
        { KernelTimer kt_(ctx, "k_probe_lookup_rank");
          DFGPU_INT_KEY_DISPATCH(pk->type, hipLaunchKernelGGL((k_probe_lookup_runs<T>), dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const T*)pk->values->ptr, rp, m, t->key_min,
                                                              (const uint64_t*)t->bitmap->ptr, (const uint32_t*)t->rank_prefix->ptr, (const uint32_t*)t->sel_rows->values->ptr, t->sel_rows->length, t->n_build,
                                                              (uint32_t*)run_of->ptr, (uint32_t*)cnt->ptr)); }
        KERNEL_CHECK();
        exclusive_scan_u32(ctx, (const uint32_t*)cnt->ptr, (uint64_t*)offs->ptr, m, ctx->d_scratch64 + 8);
        total = (int64_t)read_scratch(ctx, 8);
      }
      if (total > 0xFFFFFFF0ll) fail(DFGPU_RESOURCES_EXHAUSTED, "join output of %lld rows for one probe batch; split the probe batch", (long long)total);
      ob.a = new_fixed(ctx, DFGPU_UINT64, total); op.a = new_fixed(ctx, DFGPU_UINT32, total);
      if (total) { KernelTimer kt_(ctx, "k_probe_expand");
        hipLaunchKernelGGL(k_probe_expand, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, rp, (const uint32_t*)run_of->ptr, (const uint32_t*)cnt->ptr, (const uint64_t*)offs->ptr, m,
                           (const uint32_t*)t->sel_rows->values->ptr, (const uint32_t*)nullptr, (uint64_t*)ob.get()->values->ptr, (uint32_t*)op.get()->values->ptr); }
      KERNEL_CHECK();
    } else if (t->rank_mode && use_bitmap && may_defer && t->unique && !probe_keys[0]->validity) {
      // the caller resolves build rows later, for the rows it still holds by then (dfgpu_join_lookup): only the matched probe rows leave
      *out_build_idx = nullptr; *out_probe_idx = rows.release();
      check_flags(ctx, "join_probe");
      return;
    } else if (t->rank_mode && use_bitmap) {
      ob.a = new_fixed(ctx, DFGPU_UINT64, m);
      const dfgpu_array* pk = probe_keys[0];
      if (m) { KernelTimer kt_(ctx, "k_probe_lookup_rank");
        const bool all = rows.get()->identity && m == n, sel = t->sel_rows != nullptr;
#define LR(ALL, SEL) DFGPU_INT_KEY_DISPATCH(pk->type, hipLaunchKernelGGL((k_probe_lookup_rank<T, ALL, SEL>), dim3(grid_for(m, BLOCK * LR_ROWS)), dim3(BLOCK), 0, ctx->stream, (const T*)pk->values->ptr, rp, m, t->key_min, \
                                                            (const uint64_t*)t->bitmap->ptr, t->rank_prefix ? (const uint32_t*)t->rank_prefix->ptr : nullptr, \
                                                            t->sel_rows ? (const uint32_t*)t->sel_rows->values->ptr : nullptr, t->rank_identity ? 1 : 0, (uint64_t*)ob.get()->values->ptr))
        if (all && sel) { LR(true, true); } else if (all) { LR(true, false); } else if (sel) { LR(false, true); } else { LR(false, false); }
#undef LR
      }
      KERNEL_CHECK();
      op.a = rows.release();
    } else if (t->unique) {
      ob.a = new_fixed(ctx, DFGPU_UINT64, m);
      if (m) { KernelTimer kt_(ctx, "k_probe_lookup");
        if (fast_hash) hipLaunchKernelGGL(k_probe_found, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, rp, m, (const uint32_t*)found_slot->ptr, (const uint64_t*)t->slots->ptr, (const uint32_t*)nullptr, 1,
                                          (uint64_t*)ob.get()->values->ptr, (uint32_t*)nullptr, (uint32_t*)nullptr);
        else
        hipLaunchKernelGGL(k_probe_lookup, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, t->ks, pks, rp, m, nen, fz, (const uint64_t*)t->slots->ptr, (const uint32_t*)nullptr,
                           t->capacity - 1, 1, (uint64_t*)ob.get()->values->ptr, (uint32_t*)nullptr, (uint32_t*)nullptr, ctx->d_flags); }
      KERNEL_CHECK();
      op.a = rows.release();
    } else {
      BufferPtr slot_of = alloc_buffer(ctx, (size_t)(m + 1) * 4), cnt = alloc_buffer(ctx, (size_t)(m + 1) * 4), offs = alloc_buffer(ctx, (size_t)(m + 1) * 8);
      int64_t total = 0;
      if (m) {
        { KernelTimer kt_(ctx, "k_probe_lookup");
        if (fast_hash) hipLaunchKernelGGL(k_probe_found, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, rp, m, (const uint32_t*)found_slot->ptr, (const uint64_t*)t->slots->ptr, (const uint32_t*)t->slot_count->ptr, 0,
                                          (uint64_t*)nullptr, (uint32_t*)slot_of->ptr, (uint32_t*)cnt->ptr);
        else
        hipLaunchKernelGGL(k_probe_lookup, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, t->ks, pks, rp, m, nen, fz, (const uint64_t*)t->slots->ptr, (const uint32_t*)t->slot_count->ptr,
                           t->capacity - 1, 0, (uint64_t*)nullptr, (uint32_t*)slot_of->ptr, (uint32_t*)cnt->ptr, ctx->d_flags); }
        KERNEL_CHECK();
        exclusive_scan_u32(ctx, (const uint32_t*)cnt->ptr, (uint64_t*)offs->ptr, m, ctx->d_scratch64 + 8);
        total = (int64_t)read_scratch(ctx, 8);
      }
      if (total > 0xFFFFFFF0ll) fail(DFGPU_RESOURCES_EXHAUSTED, "join output of %lld rows for one probe batch; split the probe batch", (long long)total);
      ob.a = new_fixed(ctx, DFGPU_UINT64, total); op.a = new_fixed(ctx, DFGPU_UINT32, total);
      if (total) { KernelTimer kt_(ctx, "k_probe_expand");
        hipLaunchKernelGGL(k_probe_expand, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, rp, (const uint32_t*)slot_of->ptr, (const uint32_t*)cnt->ptr, (const uint64_t*)offs->ptr, m,
                           (const uint32_t*)t->slot_start->ptr, (const uint32_t*)t->csr_rows->ptr, (uint64_t*)ob.get()->values->ptr, (uint32_t*)op.get()->values->ptr); }
      KERNEL_CHECK();
    }
    check_flags(ctx, "join_probe");
    *out_build_idx = ob.release(); *out_probe_idx = op.release();
  });
}

dfgpu_status dfgpu_join_mark_visited(dfgpu_ctx* ctx, dfgpu_join_table* t, const dfgpu_array* build_idx) {
  return guard(ctx, [&] {
    if (!t || !build_idx || build_idx->type != DFGPU_UINT64) fail(DFGPU_INVALID_ARGUMENT, "mark_visited: UINT64 build indices expected");
    int64_t n = build_idx->length; if (!n) return;
    if (!t->visited) t->visited = alloc_buffer(ctx, bitmap_bytes(t->n_build), true);
    hipLaunchKernelGGL(k_mark_bits_u64idx, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)build_idx->values->ptr,
                       build_idx->validity ? (const uint64_t*)build_idx->validity->ptr : nullptr, n, (uint64_t*)t->visited->ptr, t->n_build, ctx->d_flags);
    KERNEL_CHECK();
    check_flags(ctx, "join_mark_visited");
  });
}

dfgpu_status dfgpu_join_final_indices(dfgpu_ctx* ctx, const dfgpu_join_table* t, int32_t join_type, dfgpu_array** out_build_idx) {
  return guard(ctx, [&] {
    if (!t || !out_build_idx) fail(DFGPU_INVALID_ARGUMENT, "final_indices: null argument");
    bool semi = join_type == DFGPU_JOIN_LEFT_SEMI;
    if (!semi && join_type != DFGPU_JOIN_LEFT && join_type != DFGPU_JOIN_FULL && join_type != DFGPU_JOIN_LEFT_ANTI)
      fail(DFGPU_INVALID_ARGUMENT, "join type %d produces no final build-side batch (need_produce_result_in_final)", join_type);
    int64_t n = t->n_build, nw = (n + 63) / 64;
    if (!t->visited) const_cast<dfgpu_join_table*>(t)->visited = alloc_buffer(ctx, bitmap_bytes(n), true);      // nothing was ever marked
    BufferPtr sel = alloc_buffer(ctx, bitmap_bytes(n), true);
    if (nw) hipLaunchKernelGGL(k_combine_words, dim3(grid_for(nw, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)t->visited->ptr, semi ? 0ull : ~0ull,
                               t->build_mask ? (const uint64_t*)t->build_mask->ptr : nullptr, (uint64_t*)sel->ptr, nw);
    KERNEL_CHECK();
    ArrayHolder idx32(mask_to_indices_impl(ctx, (const uint64_t*)sel->ptr, n));
    int64_t m = idx32.get()->length;
    ArrayHolder o(new_fixed(ctx, DFGPU_UINT64, m));
    if (m) hipLaunchKernelGGL(k_u32_to_u64, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)idx32.get()->values->ptr, (uint64_t*)o.get()->values->ptr, m, 0ull);
    KERNEL_CHECK();
    *out_build_idx = o.release();
  });
}

dfgpu_status dfgpu_join_adjust_indices(dfgpu_ctx* ctx, const dfgpu_array* build_idx, const dfgpu_array* probe_idx, int64_t range_start,
                                       int64_t range_end, int32_t join_type, dfgpu_array** out_build_idx, dfgpu_array** out_probe_idx) {
  return guard(ctx, [&] {
    if (!build_idx || !probe_idx || !out_build_idx || !out_probe_idx) fail(DFGPU_INVALID_ARGUMENT, "adjust_indices: null argument");
    if (build_idx->type != DFGPU_UINT64 || probe_idx->type != DFGPU_UINT32 || build_idx->length != probe_idx->length) fail(DFGPU_INVALID_ARGUMENT, "adjust_indices: (UINT64, UINT32) index arrays of equal length expected");
    int64_t m = probe_idx->length;
    switch (join_type) {
      case DFGPU_JOIN_INNER: case DFGPU_JOIN_LEFT:
        dfgpu_array_retain(const_cast<dfgpu_array*>(build_idx)); dfgpu_array_retain(const_cast<dfgpu_array*>(probe_idx));
        *out_build_idx = const_cast<dfgpu_array*>(build_idx); *out_probe_idx = const_cast<dfgpu_array*>(probe_idx); return;
      case DFGPU_JOIN_LEFT_SEMI: case DFGPU_JOIN_LEFT_ANTI:
        *out_build_idx = new_fixed(ctx, DFGPU_UINT64, 0); *out_probe_idx = new_fixed(ctx, DFGPU_UINT32, 0); return;
      default: break;
    }
    if (range_end < range_start) range_end = range_start;
    int64_t rl = range_end - range_start, nw = (rl + 63) / 64;
    BufferPtr bm = alloc_buffer(ctx, bitmap_bytes(rl), true);
    if (m && rl) hipLaunchKernelGGL(k_mark_bits_range, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)probe_idx->values->ptr, m, range_start, range_end, (uint64_t*)bm->ptr);
    bool want_set = join_type == DFGPU_JOIN_RIGHT_SEMI;
    if (!want_set && nw) hipLaunchKernelGGL(k_combine_words, dim3(grid_for(nw, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)bm->ptr, ~0ull, (const uint64_t*)nullptr, (uint64_t*)bm->ptr, nw);
    KERNEL_CHECK();
    ArrayHolder extra(mask_to_indices_impl(ctx, (const uint64_t*)bm->ptr, rl));   // tail bits beyond rl are masked by mask_word
    int64_t e = extra.get()->length;
    if (e && range_start) hipLaunchKernelGGL(k_add_u32, dim3(grid_for(e, BLOCK)), dim3(BLOCK), 0, ctx->stream, (uint32_t*)extra.get()->values->ptr, e, (uint32_t)range_start);
    KERNEL_CHECK();
    if (join_type == DFGPU_JOIN_RIGHT_SEMI || join_type == DFGPU_JOIN_RIGHT_ANTI) {
      // left indices are unused for right semi/anti (joins/utils.rs:1257-1269): emit an all-NULL build column of matching length
      dfgpu_array* nb = nullptr;
      dfgpu_status st = dfgpu_array_new_null(ctx, DFGPU_UINT64, 0, 0, e, &nb); if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str());
      *out_build_idx = nb; *out_probe_idx = extra.release(); return;
    }
    // Right / Full: matched pairs followed by the unmatched probe rows with NULL build index (append_right_indices :1284-1306)
    ArrayHolder ob(new_fixed(ctx, DFGPU_UINT64, m + e, 0, 0, e > 0)), op(new_fixed(ctx, DFGPU_UINT32, m + e));
    if (m) { HIP_CHECK(hipMemcpyAsync(ob.get()->values->ptr, build_idx->values->ptr, (size_t)m * 8, hipMemcpyDeviceToDevice, ctx->stream));
             HIP_CHECK(hipMemcpyAsync(op.get()->values->ptr, probe_idx->values->ptr, (size_t)m * 4, hipMemcpyDeviceToDevice, ctx->stream)); }
    if (e) {
      HIP_CHECK(hipMemsetAsync((uint64_t*)ob.get()->values->ptr + m, 0, (size_t)e * 8, ctx->stream));
      HIP_CHECK(hipMemcpyAsync((uint32_t*)op.get()->values->ptr + m, extra.get()->values->ptr, (size_t)e * 4, hipMemcpyDeviceToDevice, ctx->stream));
      // validity: first m bits set
      launch_set_bits_prefix(ctx, (uint64_t*)ob.get()->validity->ptr, m);
      ob.get()->null_count = e;
    }
    *out_build_idx = ob.release(); *out_probe_idx = op.release();
  });
}

dfgpu_status dfgpu_cross_join_indices(dfgpu_ctx* ctx, int64_t first_left, int64_t count_left, int64_t n_right, int32_t left_is_u64, dfgpu_array** out_left, dfgpu_array** out_right) {
  return guard(ctx, [&] {
    if (!out_left || !out_right || first_left < 0 || count_left < 0 || n_right < 0) fail(DFGPU_INVALID_ARGUMENT, "cross_join_indices: bad argument");
    const int64_t total = count_left * n_right;
    if (total > (int64_t)1 << 31) fail(DFGPU_INVALID_ARGUMENT, "cross_join_indices: %lld pairs in one call, split the left rows", (long long)total);
    if ((left_is_u64 ? n_right : first_left + count_left) > 0xFFFFFFFFll) fail(DFGPU_NOT_IMPLEMENTED, "cross_join_indices: more than 2^32 rows on the UInt32 side");
    HIP_CHECK(hipSetDevice(ctx->device));
    ArrayHolder l(new_fixed(ctx, left_is_u64 ? DFGPU_UINT64 : DFGPU_UINT32, total)), r(new_fixed(ctx, left_is_u64 ? DFGPU_UINT32 : DFGPU_UINT64, total));
    if (total) {
      KernelTimer kt(ctx, "k_cross_indices");
      if (left_is_u64) hipLaunchKernelGGL((k_cross_indices<uint64_t, uint32_t>), dim3(grid_for(total, BLOCK)), dim3(BLOCK), 0, ctx->stream, first_left, n_right, total, (uint64_t*)l.get()->values->ptr, (uint32_t*)r.get()->values->ptr);
      else hipLaunchKernelGGL((k_cross_indices<uint32_t, uint64_t>), dim3(grid_for(total, BLOCK)), dim3(BLOCK), 0, ctx->stream, first_left, n_right, total, (uint32_t*)l.get()->values->ptr, (uint64_t*)r.get()->values->ptr);
      KERNEL_CHECK();
    }
    *out_left = l.release(); *out_right = r.release();
  });
}

}  // extern "C"

namespace dfgpu {
__global__ void k_set_bits_prefix(uint64_t* bits, int64_t m) {
  int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nw = (m + 63) >> 6;
  if (w >= nw) return;
  bits[w] = (w == nw - 1 && (m & 63)) ? ((1ull << (m & 63)) - 1ull) : ~0ull;
}
void launch_set_bits_prefix(dfgpu_ctx* ctx, uint64_t* bits, int64_t m) {
  if (m <= 0) return;
  hipLaunchKernelGGL(k_set_bits_prefix, dim3(grid_for((m + 63) / 64, BLOCK)), dim3(BLOCK), 0, ctx->stream, bits, m);
  KERNEL_CHECK();
}

}  // namespace dfgpu
