// radix_partition.h -- one-pass, deterministic hash partitioning of rows into P partitions on gfx950.
//
// Shared by the radix-partitioned hash join (pjoin.hip), the partitioned aggregation (pagg.hip) and RepartitionExec's
// BatchPartitioner (partition.hip; repartition/mod.rs:148-221).  Three launches:
//   k_rp_hist     every workgroup counts RP_G consecutive tiles into an LDS histogram and writes counts[p][t .. t + RP_G) as one burst
//   scan          ONE exclusive scan over the partition-major matrix counts[P][ntiles] gives goff[p][t] = first output slot of tile t's
//                 rows in partition p (partition-major output: partition p is the contiguous range [goff[p][0], goff[p + 1][0]))
//   k_rp_scatter  per tile: counting sort of the tile inside LDS, then every column is staged through LDS in sorted order and written
//                 out linearly, so consecutive lanes write consecutive slots of a run
// Ranking inside a tile: STABLE (P <= 256): a wave owns 512 consecutive rows of the tile; 64-wide match-any ballots rank a row among the lanes of its slab, a running
// count per (wave, partition) -- a row of a small LDS table only that wave touches -- ranks it across the wave's slabs, a prefix over the waves finishes the rank, so
// rows of a partition keep input order (what BatchPartitioner promises, repartition/mod.rs:196-214); otherwise ranks come from LDS atomics
// (rows of one (tile, partition) run land in arbitrary order inside the run -- consumers that only need the partition's row SET, the
// join and the aggregation, take that).  Tiles are always laid out in input order inside a partition.
// Write combining happens at two levels: a tile's run for one partition is contiguous (LDS-staged), and the runs of NEIGHBOURING tiles
// are adjacent in the output, so the blockIdx -> tile map gives each XCD a contiguous range of tiles: the workgroups an XCD runs
// concurrently extend the same cache lines inside that XCD's L2 (measured on MI355X, 150 M rows of 8 B key -> 12 B out, P = 1024:
// 1.88 ms with the plain map, 1.03 ms with the XCD-aware map; profiles/experiments/radix_partition_microbench.hip).
#pragma once
#include "device_utils.h"

namespace dfgpu {

constexpr int RP_R = 8;              // rows per lane and tile
constexpr int RP_G = 16;             // tiles per histogram workgroup (P <= 2048): 16 x 4 B = one 64-B burst per partition; 4 beyond
constexpr int RP_MAX_COLS = 12;
constexpr uint32_t RP_MAX_P = 8192;  // LDS: histogram [P][G] u32 <= 128 KB, scatter cnt + delta = 64 KB
constexpr uint32_t RP_MAX_STABLE_P = 256;

__device__ inline uint32_t rp_pid(uint64_t h, uint32_t P) { return (uint32_t)(((h >> 32) * (uint64_t)P) >> 32); }   // monotone in the top hash bits, any P

// hashers: selected(i) + partition id of row i
template <typename T> struct RpHashInt {        // one integer key column (widened to 64 bits, hashed with mix64), optional validity / selection bitmap
  const T* keys; const uint64_t* valid; const uint64_t* mask;
  __device__ inline bool operator()(int64_t i, uint32_t P, uint32_t* pid, uint64_t* key) const {
    *key = (uint64_t)(int64_t)keys[i];
    *pid = rp_pid(mix64(*key), P);
    return (!mask || bit_get(mask, i)) && (!valid || bit_get(valid, i));
  }
};
// rows already carrying a 32-bit row number in their low half (the join's (probe row, build row) matches): bucket = row >> shift.
// Partition ids are then monotone in the row number, so bucket order is row order.
struct RpHashRowBucket {
  const uint64_t* recs; int shift;
  __device__ inline bool operator()(int64_t i, uint32_t, uint32_t* pid, uint64_t* key) const { *key = recs[i]; *pid = (uint32_t)(*key & 0xFFFFFFFFull) >> shift; return true; }
};
struct RpHashKeySet {     // create_hashes % P (repartition/mod.rs:185) over any key columns
  KeySet ks; const uint64_t* mask; int force_zero;
  __device__ inline bool operator()(int64_t i, uint32_t P, uint32_t* pid, uint64_t* key) const {
    bool an; uint64_t h = force_zero ? 0 : keyset_hash(ks, i, 0, &an);
    *pid = (uint32_t)(h % P); *key = h;
    return !mask || bit_get(mask, i);
  }
};

struct RpHashDigit {      // partition = one digit of a 64-bit word (the LSD passes of sort.hip and of the first-seen ordering in pagg.hip)
  const uint64_t* keys; int shift; uint32_t mask;
  __device__ inline bool operator()(int64_t i, uint32_t, uint32_t* pid, uint64_t* key) const { *key = keys[i]; *pid = (uint32_t)(*key >> shift) & mask; return true; }
};

// columns moved by the scatter
enum { RP_RAW = 0, RP_KEY64 = 2, RP_HASHKEY = 3, RP_LO16 = 4, RP_CASTF64 = 5 };      // LO16: src holds 16-byte elements (Decimal128), the low 8 bytes move (width 8 on the destination side); CASTF64: src integer column of `type`, dst the double of every value (a CAST(.. AS DOUBLE) argument converted while it moves)
struct RpCol { const void* src; void* dst; int32_t width; int32_t kind; int32_t type; };   // RAW: width bytes per row (1, 2, 4, 8, 16); KEY64: src integer column of `type`, dst u64; HASHKEY: dst u64 = the 64-bit key the hasher produced for the row
// rowid_dst (optional): the original row number of every moved row (rides along with the first column's round).
// pack12_dst (optional; column 0 must be 8 bytes wide): column 0 and the row number leave as ONE array of 12-byte records
// (RpRec12: one global_store_dwordx3 per row, a (tile, partition) run is one contiguous piece instead of two)
struct RpRec12 { uint32_t lo, hi, row; };
struct RpCols { int32_t n; uint32_t* rowid_dst; RpRec12* pack12_dst; RpCol c[RP_MAX_COLS]; };

template <int NT, typename H>
__global__ void __launch_bounds__(NT) k_rp_hist(H hs, int64_t n, uint32_t P, int64_t ntiles, int G, uint32_t* counts /*[P][ntiles]*/) {
  extern __shared__ uint32_t rp_lds[];        // [P][G]
  const int64_t t0 = (int64_t)blockIdx.x * G;
  for (int x = threadIdx.x; x < (int)P * G; x += NT) rp_lds[x] = 0;
  __syncthreads();
  for (int g = 0; g < G; g++) {
    const int64_t base = (t0 + g) * (int64_t)(NT * RP_R);
    if (base >= n) break;
#pragma unroll
    for (int q = 0; q < RP_R; q++) {
      int64_t i = base + (int64_t)q * NT + threadIdx.x; uint32_t pid; uint64_t hk;
      if (i < n && hs(i, P, &pid, &hk)) atomicAdd(&rp_lds[pid * G + g], 1u);
    }
  }
  __syncthreads();
  for (int x = threadIdx.x; x < (int)P * G; x += NT) { int p = x / G, g = x % G; if (t0 + g < ntiles) counts[(int64_t)p * ntiles + t0 + g] = rp_lds[x]; }
}

// LDS of the scatter: cnt[P] u32 | delta[P] u32 | spid[TILE] u16 | slidx[TILE] u16 | stage[TILE] u64 | (STABLE) wcnt[NT / 64][P] u16
template <int NT> static inline size_t rp_scatter_lds(uint32_t P, bool stable, int rounds = 1) {
  return (size_t)P * 8 + (size_t)NT * RP_R * 4 + 8 + (size_t)NT * RP_R * 8 / (size_t)rounds + (stable ? (size_t)(NT / WAVE) * P * 2 : 0);
}

// ROUNDS = 2: a column is staged and written out in two rounds of half a tile each (rows whose staged position lies in the round's half), so the staging buffer is half
// as large: the 8192-row tile of the many-partition case then needs 74 KB instead of 106 KB of LDS and TWO workgroups fit a CU (one loading while the other writes) --
// what the partitioned join's own scatter does (pjoin.hip); twice the barriers per column.  Measured (round 4, call y) and left off by default (ctx option
// partition_two_round_staging): the pre-aggregation's 1024-way scatter of 100 M rows x three 8-byte columns 1.79 -> 2.44 ms.
template <int NT, bool STABLE, typename H, bool LO16 = false, int ROUNDS = 1>          // LO16: some column is RP_LO16 (an instantiation of its own: the extra branch in the column loop cost the others 10 %)
__global__ void __launch_bounds__(NT) k_rp_scatter(H hs, int64_t n, uint32_t P, int64_t ntiles, const uint32_t* goff, RpCols cols) {
  extern __shared__ uint32_t rp_lds[];
  constexpr int TILE = NT * RP_R, NW = NT / WAVE;
  uint32_t* cnt = rp_lds; uint32_t* delta = rp_lds + P; uint16_t* spid = (uint16_t*)(rp_lds + 2 * P);
  uint16_t* slidx = spid + TILE;                                    // tile-local index of the row staged at each position
  uint64_t* stage = (uint64_t*)(((uintptr_t)(slidx + TILE) + 7) & ~(uintptr_t)7);
  constexpr uint32_t HT = TILE / ROUNDS;                            // staged positions per round
  uint16_t* wcnt = (uint16_t*)(stage + HT);                         // STABLE only: rows of partition p in wave w (running over the wave's slabs while ranking, then exclusive over the waves)
  __shared__ uint32_t wsum[NW]; __shared__ uint32_t moved_sh; __shared__ RpCol scol[RP_MAX_COLS];
#pragma unroll
  for (int c = 0; c < RP_MAX_COLS; c++) if ((int)threadIdx.x == c) scol[c] = cols.c[c];       // static indexing of the by-value argument; the column loop reads LDS
  const int64_t per = (ntiles + 7) / 8, t = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);      // XCD x works on tiles [x * per, (x + 1) * per)
  if (t >= ntiles || (int64_t)(blockIdx.x >> 3) >= per) return;
  const int64_t base = t * (int64_t)TILE;
  const int wave = threadIdx.x >> 6;
  for (int p = threadIdx.x; p < (int)P; p += NT) cnt[p] = 0;
  if (STABLE) for (int x = threadIdx.x; x < NW * (int)P; x += NT) wcnt[x] = 0;
  __syncthreads();
  uint32_t pid[RP_R], rk[RP_R]; bool on[RP_R]; uint64_t hk[RP_R];          // hk: the key the hasher read (column kind RP_HASHKEY moves it without a second load)
  // STABLE: a wave owns RP_R * 64 CONSECUTIVE rows (slab q = 64 of them), so tile order is (wave, slab, lane) and a wave's running count per partition -- its own
  // row of wcnt, touched by no other wave, LDS operations of one wave execute in order -- ranks its rows across its slabs; otherwise slab q = rows q * NT ..
  const int64_t i0 = STABLE ? base + (int64_t)wave * (RP_R * WAVE) + lane_id() : base + threadIdx.x;
  constexpr int QS = STABLE ? WAVE : NT;
#pragma unroll
  for (int q = 0; q < RP_R; q++) {
    int64_t i = i0 + (int64_t)q * QS; pid[q] = 0; hk[q] = 0;
    on[q] = i < n && hs(i, P, &pid[q], &hk[q]);
    if (STABLE) {
      uint64_t peers = ballot64(on[q]);
      for (uint32_t b = 1; b < P; b <<= 1) { uint64_t mb = ballot64((pid[q] & b) != 0); peers &= (pid[q] & b) ? mb : ~mb; }
      const uint32_t below = (uint32_t)__popcll(peers & lanemask_lt());
      uint32_t seen = 0;
      if (on[q] && below == 0) { uint16_t* wc = wcnt + (size_t)wave * P + pid[q]; seen = *wc; *wc = (uint16_t)(seen + (uint32_t)__popcll(peers)); }
      seen = __shfl(seen, peers ? __ffsll((unsigned long long)peers) - 1 : 0, 64);       // the first lane of every peer group read the count before adding
      rk[q] = seen + below;
    } else rk[q] = on[q] ? atomicAdd(&cnt[pid[q]], 1u) : 0;
  }
  __syncthreads();
  if (STABLE) {       // thread p: counts of partition p per wave -> exclusive prefix over the waves, tile total into cnt
    for (int p = threadIdx.x; p < (int)P; p += NT) {
      uint32_t run = 0;
#pragma unroll
      for (int w = 0; w < NW; w++) { const uint32_t c = wcnt[(size_t)w * P + p]; wcnt[(size_t)w * P + p] = (uint16_t)run; run += c; }
      cnt[p] = run;
    }
    __syncthreads();
  }
  {   // exclusive scan of cnt[P]: thread t owns the PER consecutive bins from t * PER
    constexpr int PER = ((int)(STABLE ? RP_MAX_STABLE_P : RP_MAX_P) + NT - 1) / NT; uint32_t loc[PER]; uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) { int p = threadIdx.x * PER + j; loc[j] = p < (int)P ? cnt[p] : 0; s += loc[j]; }
    uint32_t inc = wave_inclusive_sum(s);
    __syncthreads();
    if (lane_id() == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - s; for (int w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
    for (int j = 0; j < PER; j++) { int p = threadIdx.x * PER + j; if (p < (int)P) { cnt[p] = run; delta[p] = goff[(int64_t)p * ntiles + t] - run; run += loc[j]; } }   // mod 2^32: slot = delta + staged position
    if (threadIdx.x == NT - 1) moved_sh = run;
  }
  __syncthreads();
  const uint32_t moved = moved_sh;
  uint32_t spos[RP_R];
#pragma unroll
  for (int q = 0; q < RP_R; q++) {
    spos[q] = on[q] ? cnt[pid[q]] + rk[q] + (STABLE ? (uint32_t)wcnt[(size_t)wave * P + pid[q]] : 0u) : 0u;
    if (on[q]) { spid[spos[q]] = (uint16_t)pid[q]; slidx[spos[q]] = (uint16_t)(i0 - base + q * QS); }
  }
  uint32_t* const rowid_dst = cols.rowid_dst;
  if (rowid_dst && cols.n == 0) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < moved; i += NT) rowid_dst[(int64_t)(uint32_t)(delta[spid[i]] + i)] = (uint32_t)(base + slidx[i]);
  }
  for (int c = 0; c < cols.n; c++) {
    const RpCol col = scol[c];
    const int halves = col.width == 16 ? 2 : 1;
    for (int hf = 0; hf < halves; hf++) for (int r = 0; r < ROUNDS; r++) {
      if (c || hf || r) __syncthreads();               // the previous column's (round's) write-out has read the staging buffer
      const uint32_t r0 = (uint32_t)r * HT, r1 = moved < r0 + HT ? moved : r0 + HT;          // staged positions of this round
      // the width switch sits outside the unrolled row loop: one load shape per column, RP_R loads in flight
#define RP_GATHER(EXPR) { _Pragma("unroll") for (int q = 0; q < RP_R; q++) if (on[q] && (ROUNDS == 1 || spos[q] - r0 < HT)) { const int64_t i = i0 + (int64_t)q * QS; stage[spos[q] - r0] = (uint64_t)(EXPR); } }
      if (col.kind == RP_HASHKEY) { _Pragma("unroll") for (int q = 0; q < RP_R; q++) if (on[q] && (ROUNDS == 1 || spos[q] - r0 < HT)) stage[spos[q] - r0] = hk[q]; }
      else if (LO16 && col.kind == RP_LO16) RP_GATHER(((const uint64_t*)col.src)[2 * i])
      else if (LO16 && col.kind == RP_CASTF64) {            // arrow-cast: integer -> Float64 is `as f64`; Int32 and Int64 sources (two load shapes: every further one is another unrolled gather in this instantiation's column loop)
        if (col.type == DFGPU_INT32) RP_GATHER(__double_as_longlong((double)((const int32_t*)col.src)[i]))
        else RP_GATHER(__double_as_longlong((double)((const int64_t*)col.src)[i]))
      }
      else if (col.kind == RP_KEY64) switch (col.type) {            // widened exactly as key_at() does
        case DFGPU_INT8: RP_GATHER((int64_t)((const int8_t*)col.src)[i]) break;
        case DFGPU_INT16: RP_GATHER((int64_t)((const int16_t*)col.src)[i]) break;
        case DFGPU_INT32: case DFGPU_DATE32: RP_GATHER((int64_t)((const int32_t*)col.src)[i]) break;
        case DFGPU_UINT8: RP_GATHER(((const uint8_t*)col.src)[i]) break;
        case DFGPU_UINT16: RP_GATHER(((const uint16_t*)col.src)[i]) break;
        case DFGPU_UINT32: RP_GATHER(((const uint32_t*)col.src)[i]) break;
        default: RP_GATHER(((const uint64_t*)col.src)[i]) break;
      }
      else switch (col.width) {
        case 1: RP_GATHER(((const uint8_t*)col.src)[i]) break;
        case 2: RP_GATHER(((const uint16_t*)col.src)[i]) break;
        case 4: RP_GATHER(((const uint32_t*)col.src)[i]) break;
        case 8: RP_GATHER(((const uint64_t*)col.src)[i]) break;
        default: RP_GATHER(((const uint64_t*)col.src)[2 * i + hf]) break;
      }
#undef RP_GATHER
      __syncthreads();
#define RP_WRITE(T, IDX) for (uint32_t i = r0 + threadIdx.x; i < r1; i += NT) { const int64_t pos = (int64_t)(uint32_t)(delta[spid[i]] + i); ((T*)col.dst)[IDX] = (T)stage[i - r0]; }
      if (cols.pack12_dst && c == 0) {
        RpRec12* const d12 = cols.pack12_dst;
        for (uint32_t i = r0 + threadIdx.x; i < r1; i += NT) { const int64_t pos = (int64_t)(uint32_t)(delta[spid[i]] + i); const uint64_t v = stage[i - r0]; d12[pos] = RpRec12{ (uint32_t)v, (uint32_t)(v >> 32), (uint32_t)(base + slidx[i]) }; }
      } else if (rowid_dst && c == 0 && hf == 0 && col.width == 8) {           // the common (key, row id) pair in one sweep
        for (uint32_t i = r0 + threadIdx.x; i < r1; i += NT) { const int64_t pos = (int64_t)(uint32_t)(delta[spid[i]] + i); ((uint64_t*)col.dst)[pos] = stage[i - r0]; rowid_dst[pos] = (uint32_t)(base + slidx[i]); }
      } else {
      if (rowid_dst && c == 0 && hf == 0) for (uint32_t i = r0 + threadIdx.x; i < r1; i += NT) rowid_dst[(int64_t)(uint32_t)(delta[spid[i]] + i)] = (uint32_t)(base + slidx[i]);
      switch (col.width) {
        case 1: RP_WRITE(uint8_t, pos) break;
        case 2: RP_WRITE(uint16_t, pos) break;
        case 4: RP_WRITE(uint32_t, pos) break;
        case 8: RP_WRITE(uint64_t, pos) break;
        default: RP_WRITE(uint64_t, 2 * pos + hf) break;
      }
      }
#undef RP_WRITE
    }
  }
}

// Few partitions (P <= 16, the RepartitionExec / exchange fan-out of one node): no data goes through LDS at all.  Every row's slot follows from the tile's global
// offset, the counts of its partition in the (slab, wave) pieces in front of it (a small LDS table) and its rank among the lanes of its wave (ballots); each lane
// then copies its row's columns straight from source to slot.  A wave's 64 rows land in at most P runs of consecutive slots (~64 / P rows each), reads are fully
// coalesced, 16-byte columns move in one piece.  Stable by construction.
template <int NT, int DP, typename H>          // DP: most partitions this instance takes (16 or 256: the (slab, wave, partition) count table is DP wide)
__global__ void __launch_bounds__(NT) k_rp_scatter_direct(H hs, int64_t n, uint32_t P, int64_t ntiles, const uint32_t* goff, RpCols cols) {
  constexpr int TILE = NT * RP_R, NW = NT / WAVE;
  __shared__ uint16_t wcnt[RP_R * NW * DP]; __shared__ uint32_t tbase[DP]; __shared__ RpCol scol[RP_MAX_COLS];
#pragma unroll
  for (int c = 0; c < RP_MAX_COLS; c++) if ((int)threadIdx.x == c) scol[c] = cols.c[c];
  const int64_t per = (ntiles + 7) / 8, t = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (t >= ntiles || (int64_t)(blockIdx.x >> 3) >= per) return;
  const int64_t base = t * (int64_t)TILE; const int wave = threadIdx.x >> 6;
  for (int x = threadIdx.x; x < RP_R * NW * DP; x += NT) wcnt[x] = 0;
  if (threadIdx.x < P) tbase[threadIdx.x] = goff[(int64_t)threadIdx.x * ntiles + t];
  __syncthreads();
  uint32_t pid[RP_R], rk[RP_R]; bool on[RP_R]; uint64_t hk[RP_R];
#pragma unroll
  for (int q = 0; q < RP_R; q++) {
    int64_t i = base + (int64_t)q * NT + threadIdx.x; pid[q] = 0; hk[q] = 0;
    on[q] = i < n && hs(i, P, &pid[q], &hk[q]);
    uint64_t peers = ballot64(on[q]);
    for (uint32_t b = 1; b < P; b <<= 1) { uint64_t mb = ballot64((pid[q] & b) != 0); peers &= (pid[q] & b) ? mb : ~mb; }
    rk[q] = (uint32_t)__popcll(peers & lanemask_lt());
    if (on[q] && rk[q] == 0) wcnt[((size_t)q * NW + wave) * DP + pid[q]] = (uint16_t)__popcll(peers);
  }
  __syncthreads();
  if (threadIdx.x < P) { uint32_t run = 0; for (int x = 0; x < RP_R * NW; x++) { uint16_t c = wcnt[(size_t)x * DP + threadIdx.x]; wcnt[(size_t)x * DP + threadIdx.x] = (uint16_t)run; run += c; } }
  __syncthreads();
  int64_t pos[RP_R];
#pragma unroll
  for (int q = 0; q < RP_R; q++) pos[q] = on[q] ? (int64_t)(tbase[pid[q]] + (uint32_t)wcnt[((size_t)q * NW + wave) * DP + pid[q]] + rk[q]) : 0;
  if (cols.rowid_dst) {
#pragma unroll
    for (int q = 0; q < RP_R; q++) if (on[q]) cols.rowid_dst[pos[q]] = (uint32_t)(base + (int64_t)q * NT + threadIdx.x);
  }
  const int64_t i0 = base + threadIdx.x;
  for (int c = 0; c < cols.n; c++) {
    const RpCol col = scol[c];
#define RP_MOVE(DT, EXPR) { _Pragma("unroll") for (int q = 0; q < RP_R; q++) if (on[q]) { const int64_t i = i0 + (int64_t)q * NT; ((DT*)col.dst)[pos[q]] = (DT)(EXPR); } }
    if (col.kind == RP_HASHKEY) { _Pragma("unroll") for (int q = 0; q < RP_R; q++) if (on[q]) ((uint64_t*)col.dst)[pos[q]] = hk[q]; }
    else if (col.kind == RP_KEY64) switch (col.type) {
      case DFGPU_INT8: RP_MOVE(uint64_t, (int64_t)((const int8_t*)col.src)[i]) break;
      case DFGPU_INT16: RP_MOVE(uint64_t, (int64_t)((const int16_t*)col.src)[i]) break;
      case DFGPU_INT32: case DFGPU_DATE32: RP_MOVE(uint64_t, (int64_t)((const int32_t*)col.src)[i]) break;
      case DFGPU_UINT8: RP_MOVE(uint64_t, ((const uint8_t*)col.src)[i]) break;
      case DFGPU_UINT16: RP_MOVE(uint64_t, ((const uint16_t*)col.src)[i]) break;
      case DFGPU_UINT32: RP_MOVE(uint64_t, ((const uint32_t*)col.src)[i]) break;
      default: RP_MOVE(uint64_t, ((const uint64_t*)col.src)[i]) break;
    }
    else switch (col.width) {
      case 1: RP_MOVE(uint8_t, ((const uint8_t*)col.src)[i]) break;
      case 2: RP_MOVE(uint16_t, ((const uint16_t*)col.src)[i]) break;
      case 4: RP_MOVE(uint32_t, ((const uint32_t*)col.src)[i]) break;
      case 8: RP_MOVE(uint64_t, ((const uint64_t*)col.src)[i]) break;
      default: { _Pragma("unroll") for (int q = 0; q < RP_R; q++) if (on[q]) ((ulonglong2*)col.dst)[pos[q]] = ((const ulonglong2*)col.src)[i0 + (int64_t)q * NT]; } break;
    }
#undef RP_MOVE
  }
}

// starts[p] = first slot of partition p, starts[P] = rows moved (the scan's total)
static __global__ void k_rp_starts(const uint32_t* goff, int64_t ntiles, uint32_t P, const uint64_t* d_total, uint32_t* starts) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) starts[p] = goff[(int64_t)p * ntiles];
  if (p == P) starts[P] = (uint32_t)*d_total;
}

struct RpResult { uint32_t P = 0; int64_t ntiles = 0; BufferPtr starts; };     // starts: u32[P + 1] on the device

// Partition rows 0 .. n-1 (those the hasher selects) into P partitions, moving `cols`.  Destination buffers must hold n rows.
// d_total: device u64 that receives the number of rows moved.  timer names: <prefix>_hist / _scan / _scatter.
template <typename H>
static RpResult rp_partition(dfgpu_ctx* ctx, H hs, int64_t n, uint32_t P, const RpCols& cols, bool stable, uint64_t* d_total,
                             const char* t_hist, const char* t_scan, const char* t_scatter, bool want_starts = true, bool wide_rows = false) {
  if (P < 1 || P > RP_MAX_P) fail(DFGPU_INTERNAL, "rp_partition: %u partitions (1..%u supported)", P, RP_MAX_P);
  if (stable && P > RP_MAX_STABLE_P) fail(DFGPU_INTERNAL, "rp_partition: stable order supports up to %u partitions, got %u", RP_MAX_STABLE_P, P);
  if (n > 0xFFFFFFF0ll) fail(DFGPU_NOT_IMPLEMENTED, "partitioning above 2^32-16 rows");
  RpResult r; r.P = P;
  if (want_starts) r.starts = alloc_buffer(ctx, (size_t)(P + 1) * 4);       // the passes of a sort only need the rows moved
  const bool big = P > 512 && P <= 2048 && !stable;        // 8192-row tiles halve the count matrix; 4096-row tiles give more workgroups per CU (and leave LDS for P > 2048)
  // the LDS-free scatter: always for <= 16 partitions; up to 256 when a row is one column (a sort pass: 3.0 against 3.2 ms per 100 M rows) or holds a 16-byte column (the staged
  // scatter moves those as two halves: 600 M rows x 44 B into 64 / 128 / 256: 16.9 / 18.0 / 22.5 ms against 22.8 / 23.9 / 25.9); rows of several narrow columns (the aggregation's
  // (key, row, value): 2.7 against 1.4 ms) stay with the staged one
  bool wide16 = false, lo16 = false; for (int c = 0; c < cols.n; c++) { wide16 |= cols.c[c].width == 16; lo16 |= cols.c[c].kind == RP_LO16 || cols.c[c].kind == RP_CASTF64; }      // lo16: the instantiation with the extra column kinds
  const bool direct = stable && !cols.pack12_dst && !lo16 && (P <= 16 || (P <= 256 && !wide_rows && (cols.n <= 1 || wide16)));
  const bool small_wg = stable && P > 16 && !wide_rows && !direct;    // wide_rows: several columns move per row (the aggregation's second level: 20 B) -- there the 4096-row tile's longer runs win (1.65 -> 1.44 ms),
                                                           // while the sort's single 8-byte column is faster with more workgroups per CU (3.2 against 3.7 ms)                  // stable with many partitions: 256-thread workgroups (the count table is 64 P bytes) keep several on a CU;
  const int nt = small_wg ? 256 : big ? 1024 : 512, tile = nt * RP_R;      // few partitions want the longer runs of a 4096-row tile
  const int64_t ntiles = n ? (n + tile - 1) / tile : 1; r.ntiles = ntiles;
  int G = P <= 2048 ? RP_G : 4; while (G > 1 && (ntiles + G - 1) / G < 1024) G >>= 1;      // a small input still wants ~1000 histogram workgroups (a few dozen of them counting 16 tiles each was 0.04 ms per sort pass of 1 M rows)
  BufferPtr counts = alloc_buffer(ctx, (size_t)P * ntiles * 4);
  const size_t hl = (size_t)P * G * 4;
  const int64_t nh = (ntiles + G - 1) / G;
  { KernelTimer kt_(ctx, t_hist);
    if (big) { HIP_CHECK(hipFuncSetAttribute((const void*)k_rp_hist<1024, H>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
      hipLaunchKernelGGL((k_rp_hist<1024, H>), dim3((unsigned)nh), dim3(1024), hl, ctx->stream, hs, n, P, ntiles, G, (uint32_t*)counts->ptr); }
    else if (small_wg) hipLaunchKernelGGL((k_rp_hist<256, H>), dim3((unsigned)nh), dim3(256), hl, ctx->stream, hs, n, P, ntiles, G, (uint32_t*)counts->ptr);
    else { HIP_CHECK(hipFuncSetAttribute((const void*)k_rp_hist<512, H>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
      hipLaunchKernelGGL((k_rp_hist<512, H>), dim3((unsigned)nh), dim3(512), hl, ctx->stream, hs, n, P, ntiles, G, (uint32_t*)counts->ptr); }
    KERNEL_CHECK(); }
  { KernelTimer kt_(ctx, t_scan); exclusive_scan_u32_inplace32(ctx, (uint32_t*)counts->ptr, (int64_t)P * ntiles, d_total); }
  if (want_starts) hipLaunchKernelGGL(k_rp_starts, dim3((P + 1 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)counts->ptr, ntiles, P, (const uint64_t*)d_total, (uint32_t*)r.starts->ptr);
  KERNEL_CHECK();
  if (n) { KernelTimer kt_(ctx, t_scatter);
    const unsigned grid = (unsigned)(((ntiles + 7) / 8) * 8);
#define RP_LAUNCH(NT_, ST_) { if (lo16) { HIP_CHECK(hipFuncSetAttribute((const void*)k_rp_scatter<NT_, ST_, H, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512)); \
      hipLaunchKernelGGL((k_rp_scatter<NT_, ST_, H, true>), dim3(grid), dim3(NT_), rp_scatter_lds<NT_>(P, ST_), ctx->stream, hs, n, P, ntiles, (const uint32_t*)counts->ptr, cols); } else { \
      HIP_CHECK(hipFuncSetAttribute((const void*)k_rp_scatter<NT_, ST_, H>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512)); \
      hipLaunchKernelGGL((k_rp_scatter<NT_, ST_, H>), dim3(grid), dim3(NT_), rp_scatter_lds<NT_>(P, ST_), ctx->stream, hs, n, P, ntiles, (const uint32_t*)counts->ptr, cols); } }
    if (direct && P <= 16) hipLaunchKernelGGL((k_rp_scatter_direct<512, 16, H>), dim3(grid), dim3(512), 0, ctx->stream, hs, n, P, ntiles, (const uint32_t*)counts->ptr, cols);
    else if (direct) hipLaunchKernelGGL((k_rp_scatter_direct<512, 256, H>), dim3(grid), dim3(512), 0, ctx->stream, hs, n, P, ntiles, (const uint32_t*)counts->ptr, cols);
    else if (small_wg) RP_LAUNCH(256, true) else if (stable) RP_LAUNCH(512, true)
    else if (big && ctx->partition_two_round_staging) {          // the 8192-row tile staged in two rounds: two workgroups per CU
      if (lo16) { HIP_CHECK(hipFuncSetAttribute((const void*)k_rp_scatter<1024, false, H, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
        hipLaunchKernelGGL((k_rp_scatter<1024, false, H, true, 2>), dim3(grid), dim3(1024), rp_scatter_lds<1024>(P, false, 2), ctx->stream, hs, n, P, ntiles, (const uint32_t*)counts->ptr, cols); }
      else { HIP_CHECK(hipFuncSetAttribute((const void*)k_rp_scatter<1024, false, H, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
        hipLaunchKernelGGL((k_rp_scatter<1024, false, H, false, 2>), dim3(grid), dim3(1024), rp_scatter_lds<1024>(P, false, 2), ctx->stream, hs, n, P, ntiles, (const uint32_t*)counts->ptr, cols); }
    }
    else if (big) RP_LAUNCH(1024, false) else RP_LAUNCH(512, false)
#undef RP_LAUNCH
    KERNEL_CHECK(); }
  return r;
}

}  // namespace dfgpu
