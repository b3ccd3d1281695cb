// Experiment (not product code), round 3: candidates for the next radix-partitioned hash join pipeline on gfx950.
//   scatter  E  exact: histogram [P][tiles] -> scan -> LDS-staged scatter (what csrc/radix_partition.h does)
//            A  single pass: per-tile LDS counting sort, every (shard, partition) region has a fixed capacity, a tile reserves its runs with
//               one returning atomic per partition on sharded cursors (shard = blockIdx % S ~ the XCD), no histogram, no count matrix
//   join     F  found[probe row] = build row scatter + order-restoring compaction (round 2)
//            H  hits (probe row, build row) leave per partition, compacted per wave through an LDS counter; order restored by a sort on
//               the probe row (here: hipcub radix sort as a stand-in for the in-tree passes)
// Build: hipcc --offload-arch=gfx950 -O3 -o pjoin2_microbench.bin pjoin2_microbench.hip
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <type_traits>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__host__ __device__ inline uint64_t mix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL; return x ^ (x >> 31);
}
__device__ inline uint32_t pid_of(uint64_t h, uint32_t P) { return (uint32_t)(((h >> 32) * (uint64_t)P) >> 32); }
struct Rec12 { uint32_t lo, hi, row; };
constexpr uint32_t EMPTY = 0xFFFFFFFFu;

// ------------------------------------------------------------------------------------------------ exact path: hist
template <int NT, int R, int G>
__global__ void __launch_bounds__(NT) k_hist(const uint64_t* keys, long n, uint32_t P, long ntiles, uint32_t* counts /*[P][ntiles]*/) {
  extern __shared__ uint32_t lds[];        // [P][G]
  const long t0 = (long)blockIdx.x * G;
  for (int x = threadIdx.x; x < (int)P * G; x += NT) lds[x] = 0;
  __syncthreads();
  for (int g = 0; g < G; g++) {
    const long base = (t0 + g) * (long)(NT * R);
    if (base >= n) break;
    uint64_t k[R];
#pragma unroll
    for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; k[q] = i < n ? keys[i] : 0; }
#pragma unroll
    for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; if (i < n) atomicAdd(&lds[pid_of(mix64(k[q]), P) * G + g], 1u); }
  }
  __syncthreads();
  for (int x = threadIdx.x; x < (int)P * G; x += NT) { int p = x / G, g = x % G; if (t0 + g < ntiles) counts[(long)p * ntiles + t0 + g] = lds[x]; }
}

// ------------------------------------------------------------------------------------------------ staged scatter, exact (MODE 0) or atomic cursors (MODE 1)
// LDS: cnt[P] | delta[P] | spid[TILE] u16 | slidx[TILE] u16 | skey[TILE] u64
template <int NT, int R, int MODE>
__global__ void __launch_bounds__(NT) k_scatter(const uint64_t* keys, long n, uint32_t P, long ntiles, const uint32_t* goff, int S, uint32_t cap, uint32_t* cursors, uint32_t* overflow, Rec12* out) {
  extern __shared__ uint32_t lds[];
  constexpr int TILE = NT * R;
  uint32_t* cnt = lds; uint32_t* delta = lds + P; uint16_t* spid = (uint16_t*)(lds + 2 * P); uint16_t* slidx = spid + TILE;
  uint64_t* skey = (uint64_t*)(((uintptr_t)(slidx + TILE) + 7) & ~(uintptr_t)7);
  __shared__ uint32_t wsum[NT / 64]; __shared__ uint32_t moved_sh;
  long t; int shard = 0;
  if (MODE == 0) { long per = (ntiles + 7) / 8; t = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3); if (t >= ntiles || (long)(blockIdx.x >> 3) >= per) return; }
  else { t = blockIdx.x; shard = blockIdx.x % S; if (t >= ntiles) return; }
  const long base = t * (long)TILE;
  for (int p = threadIdx.x; p < (int)P; p += NT) cnt[p] = 0;
  __syncthreads();
  uint64_t k[R]; uint32_t pid[R], rk[R]; bool on[R];
#pragma unroll
  for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; on[q] = i < n; k[q] = on[q] ? keys[i] : 0; }
#pragma unroll
  for (int q = 0; q < R; q++) { pid[q] = pid_of(mix64(k[q]), P); rk[q] = on[q] ? atomicAdd(&cnt[pid[q]], 1u) : 0; }
  __syncthreads();
  {
    constexpr int PER = 8192 / NT; uint32_t loc[PER]; uint32_t s = 0; uint32_t res[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) { int p = threadIdx.x * PER + j; loc[j] = p < (int)P ? cnt[p] : 0; s += loc[j]; }
    if (MODE == 1) {      // reserve the runs: one returning atomic per non-empty (tile, partition); 64 consecutive cursors share a 4 KB page of their own
#pragma unroll
      for (int j = 0; j < PER; j++) { int p = threadIdx.x * PER + j; res[j] = 0;
        if (p < (int)P && loc[j]) { uint32_t* c = cursors + ((size_t)(shard * (P >> 6) + (p >> 6)) << 10) + (p & 63); res[j] = atomicAdd(c, loc[j]); if (res[j] + loc[j] > cap) { *overflow = 1; res[j] = 0xFFFFFFFFu; } } }
    }
    uint32_t inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if ((threadIdx.x & 63) >= d) inc += o; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t run = inc - s; for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += wsum[w];
#pragma unroll
    for (int j = 0; j < PER; j++) { int p = threadIdx.x * PER + j; if (p < (int)P) {
      cnt[p] = run;
      if (MODE == 0) delta[p] = goff[(long)p * ntiles + t] - run;
      else delta[p] = res[j] == 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)(((size_t)p * S + shard) * cap) + res[j] - run;    // region sizes keep positions below 2^32
      run += loc[j]; } }
    if (threadIdx.x == NT - 1) moved_sh = run;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < R; q++) if (on[q]) { uint32_t s = cnt[pid[q]] + rk[q]; spid[s] = (uint16_t)pid[q]; slidx[s] = (uint16_t)(q * NT + threadIdx.x); skey[s] = k[q]; }
  __syncthreads();
  const uint32_t moved = moved_sh;
  for (uint32_t i = threadIdx.x; i < moved; i += NT) {
    const uint32_t d = delta[spid[i]]; if (MODE == 1 && d == 0xFFFFFFFFu) continue;
    const uint32_t pos = d + i; const uint64_t v = skey[i];
    out[pos] = Rec12{ (uint32_t)v, (uint32_t)(v >> 32), (uint32_t)(base + slidx[i]) };
  }
}


// ------------------------------------------------------------------------------------------------ v2: u16 histogram with the next tile's keys in flight
template <int NT, int R, int G>
__global__ void __launch_bounds__(NT) k_hist2(const uint64_t* keys, long n, uint32_t P, long ntiles, uint16_t* counts /*[P][ntiles]*/) {
  extern __shared__ uint32_t lds[];        // [P][G] u16, two counters per word
  const long t0 = (long)blockIdx.x * G;
  for (int x = threadIdx.x; x < (int)P * G / 2; x += NT) lds[x] = 0;
  uint64_t k[R], kn[R];
  { const long base = t0 * (long)(NT * R);
#pragma unroll
    for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; kn[q] = i < n ? keys[i] : 0; } }
  __syncthreads();
  for (int g = 0; g < G; g++) {
    const long base = (t0 + g) * (long)(NT * R);
    if (base >= n) break;
#pragma unroll
    for (int q = 0; q < R; q++) k[q] = kn[q];
    if (g + 1 < G) { const long nb2 = base + (long)(NT * R);
#pragma unroll
      for (int q = 0; q < R; q++) { long i = nb2 + (long)q * NT + threadIdx.x; kn[q] = i < n ? keys[i] : 0; } }
#pragma unroll
    for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; if (i < n) { uint32_t c = pid_of(mix64(k[q]), P) * G + g; atomicAdd(&lds[c >> 1], 1u << ((c & 1) * 16)); } }
  }
  __syncthreads();
  // counts[p][t0 .. t0 + G): G u16 = 32 B per partition
  const uint16_t* l16 = (const uint16_t*)lds;
  for (int x = threadIdx.x; x < (int)P * G; x += NT) { int p = x / G, g = x % G; if (t0 + g < ntiles) counts[(long)p * ntiles + t0 + g] = l16[x]; }
}
struct U16ToU32 { __host__ __device__ uint32_t operator()(uint16_t v) const { return v; } };

// persistent LDS-staged scatter: 8 x W workgroups, XCD x (= blockIdx & 7 under round-robin placement) owns a contiguous range of tiles and its W workgroups
// take them round robin, so the tiles in flight on one XCD are neighbours; the next tile's keys and offsets are loaded while this tile is sorted and written
template <int NT, int R, int ABL>
__global__ void __launch_bounds__(NT) k_scatter2(const uint64_t* keys, long n, uint32_t P, long ntiles, const uint32_t* goff, int W, Rec12* out) {
  extern __shared__ uint32_t lds[];
  constexpr int TILE = NT * R;
  uint32_t* cnt = lds; uint32_t* delta = lds + P; uint16_t* spid = (uint16_t*)(lds + 2 * P); uint16_t* slidx = spid + TILE;
  uint64_t* skey = (uint64_t*)(((uintptr_t)(slidx + TILE) + 7) & ~(uintptr_t)7);
  __shared__ uint32_t wsum[NT / 64]; __shared__ uint32_t moved_sh;
  constexpr int PER = 2048 / NT < 1 ? 1 : 2048 / NT;        // P <= 2048 here
  const long per = (ntiles + 7) / 8; const long x = blockIdx.x & 7, j = blockIdx.x >> 3;
  const long tend = (x + 1) * per < ntiles ? (x + 1) * per : ntiles;
  long t = x * per + j;
  if (t >= tend) return;
  uint64_t kn[R]; uint32_t gn[PER];
#pragma unroll
  for (int q = 0; q < R; q++) { long i = t * (long)TILE + (long)q * NT + threadIdx.x; kn[q] = i < n ? keys[i] : 0; }
#pragma unroll
  for (int jj = 0; jj < PER; jj++) { int p = threadIdx.x * PER + jj; gn[jj] = p < (int)P ? goff[(long)p * ntiles + t] : 0; }
  for (; t < tend; t += W) {
    const long base = t * (long)TILE;
    uint64_t k[R]; uint32_t gc[PER]; uint32_t pid[R], rk[R]; bool on[R];
#pragma unroll
    for (int q = 0; q < R; q++) { k[q] = kn[q]; on[q] = base + (long)q * NT + threadIdx.x < n; }
#pragma unroll
    for (int jj = 0; jj < PER; jj++) gc[jj] = gn[jj];
    for (int p = threadIdx.x; p < (int)P; p += NT) cnt[p] = 0;
    __syncthreads();                                  // also: the previous tile's write-out has finished reading the staging arrays
    const long tn = t + W;
    if (tn < tend) {
#pragma unroll
      for (int q = 0; q < R; q++) { long i = tn * (long)TILE + (long)q * NT + threadIdx.x; kn[q] = i < n ? keys[i] : 0; }
#pragma unroll
      for (int jj = 0; jj < PER; jj++) { int p = threadIdx.x * PER + jj; gn[jj] = p < (int)P ? goff[(long)p * ntiles + tn] : 0; }
    }
#pragma unroll
    for (int q = 0; q < R; q++) { pid[q] = pid_of(mix64(k[q]), P); rk[q] = on[q] ? atomicAdd(&cnt[pid[q]], 1u) : 0; }
    __syncthreads();
    {
      uint32_t loc[PER]; uint32_t s = 0;
#pragma unroll
      for (int jj = 0; jj < PER; jj++) { int p = threadIdx.x * PER + jj; loc[jj] = p < (int)P ? cnt[p] : 0; s += loc[jj]; }
      uint32_t inc = s;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if ((threadIdx.x & 63) >= d) inc += o; }
      if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
      __syncthreads();
      uint32_t run = inc - s; for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += wsum[w];
#pragma unroll
      for (int jj = 0; jj < PER; jj++) { int p = threadIdx.x * PER + jj; if (p < (int)P) { cnt[p] = run; delta[p] = gc[jj] - run; run += loc[jj]; } }
      if (threadIdx.x == NT - 1) moved_sh = run;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R; q++) if (on[q] && ABL != 3) { uint32_t s = cnt[pid[q]] + rk[q]; spid[s] = (uint16_t)pid[q]; slidx[s] = (uint16_t)(q * NT + threadIdx.x); skey[s] = k[q]; }
    __syncthreads();
    const uint32_t moved = moved_sh;
    if (ABL == 0) for (uint32_t i = threadIdx.x; i < moved; i += NT) {
      const uint32_t pos = delta[spid[i]] + i; const uint64_t v = skey[i];
      out[pos] = Rec12{ (uint32_t)v, (uint32_t)(v >> 32), (uint32_t)(base + slidx[i]) };
    }
    if (ABL == 2) for (uint32_t i = threadIdx.x; i < moved; i += NT) {
      const uint32_t pos = (uint32_t)base + i + (delta[spid[i]] & 1u); const uint64_t v = skey[i];
      out[pos] = Rec12{ (uint32_t)v, (uint32_t)(v >> 32), (uint32_t)(base + slidx[i]) };
    }
    if (ABL == 1 || ABL == 3) { if (moved == 0xFFFFFFFFu) out[threadIdx.x] = Rec12{ cnt[threadIdx.x], delta[threadIdx.x], (uint32_t)skey[threadIdx.x] }; }
  }
}

// ------------------------------------------------------------------------------------------------ join
constexpr int IDX_BITS = 14; constexpr uint32_t IDX_MASK = (1u << IDX_BITS) - 1u, TAG_MASK = (1u << (31 - IDX_BITS)) - 1u;
__device__ inline void pj_hash(uint32_t lo, uint32_t hi, uint32_t M, int sbits, uint32_t* group, uint32_t* tagsh) {
  uint32_t a = lo ^ (hi * 0x9E3779B1u), x = a * 0x85EBCA6Bu; x ^= x >> 13;
  uint32_t y = x * 0xC2B2AE35u;
  *group = (y >> (32 - sbits)) & M & ~3u;
  *tagsh = ((y ^ (y >> 16) ^ a) & TAG_MASK) << IDX_BITS;
}
__device__ inline bool pj_group(const uint4 v, uint32_t tagsh, uint32_t s, uint32_t* cand, uint32_t* pos) {
  const uint32_t c0 = ((v.x & ~IDX_MASK) == tagsh ? 2u : 0u) | (v.x >> 31), c1 = ((v.y & ~IDX_MASK) == tagsh ? 2u : 0u) | (v.y >> 31);
  const uint32_t c2 = ((v.z & ~IDX_MASK) == tagsh ? 2u : 0u) | (v.z >> 31), c3 = ((v.w & ~IDX_MASK) == tagsh ? 2u : 0u) | (v.w >> 31);
  uint32_t code = c3, val = v.w, j = 3;
  code = c2 ? c2 : code; val = c2 ? v.z : val; j = c2 ? 2u : j;
  code = c1 ? c1 : code; val = c1 ? v.y : val; j = c1 ? 1u : j;
  code = c0 ? c0 : code; val = c0 ? v.x : val; j = c0 ? 0u : j;
  *cand = (code & 2u) ? val : EMPTY; *pos = s + j;
  return code != 0;
}
// probe records of partition p live in S pieces: piece s = prec[pbase(p, s) .. + pcnt[p * S + s]); EMIT 0: found[row] = brow; EMIT 1: hits leave compacted per wave
template <int NT, int U, int EMIT>
__global__ void __launch_bounds__(NT) k_join(const Rec12* brec, const uint32_t* bstart, const Rec12* prec, const uint32_t* pstart /*S == 0: [P + 1]*/, int S, uint32_t cap, const uint32_t* pcnt, int sbits,
                                            uint32_t* found, uint64_t* hits, uint32_t* hcount) {
  extern __shared__ uint4 tab4[];
  uint32_t* const tab = (uint32_t*)tab4;
  __shared__ uint32_t piece_end[64]; __shared__ uint32_t hit_cursor;
  const uint32_t Sl = 1u << sbits, M = Sl - 1;
  const int p = blockIdx.x;
  for (uint32_t s = threadIdx.x; s < Sl; s += NT) tab[s] = EMPTY;
  if (threadIdx.x == 0) { hit_cursor = 0; uint32_t run = 0; if (S) for (int s = 0; s < S; s++) { uint32_t c = pcnt[((size_t)(s * (gridDim.x >> 6) + (p >> 6)) << 10) + (p & 63)]; run += c < cap ? c : cap; piece_end[s] = run; } }
  __syncthreads();
  const uint32_t b0 = bstart[p], nb = bstart[p + 1] - b0;
  const Rec12* br = brec + b0;
  for (uint32_t j = threadIdx.x; j < nb; j += NT) {
    Rec12 k = br[j]; uint32_t s, tagsh; pj_hash(k.lo, k.hi, M, sbits, &s, &tagsh); const uint32_t ent = tagsh | j;
    for (;;) { uint32_t old = atomicCAS(&tab[s], EMPTY, ent); if (old == EMPTY) break; s = (s + 1) & M; }
  }
  __syncthreads();
  uint32_t q0, q1;
  if (S) { q0 = 0; q1 = piece_end[S - 1]; } else { q0 = pstart[p]; q1 = pstart[p + 1]; }
  const size_t hbase = S ? (size_t)p * S * cap : (size_t)q0;
  auto rec_at = [&](uint32_t i) -> const Rec12* {
    if (!S) return prec + i;
    int s = 0; uint32_t lo = 0;
#pragma unroll 1
    for (; s < S - 1 && i >= piece_end[s]; s++) lo = piece_end[s];
    return prec + ((size_t)p * S + s) * cap + (i - lo);
  };
  for (uint32_t i0 = q0 + threadIdx.x; i0 < q1; i0 += NT * U) {
    Rec12 rc[U]; bool on[U];
#pragma unroll
    for (int u = 0; u < U; u++) { uint32_t i = i0 + u * NT; on[u] = i < q1; rc[u] = *rec_at(on[u] ? i : q1 - 1); }
    uint32_t s[U], tagsh[U], cand[U], pos[U]; uint4 v[U]; bool walking[U]; bool more = false;
#pragma unroll
    for (int u = 0; u < U; u++) { pj_hash(rc[u].lo, rc[u].hi, M, sbits, &s[u], &tagsh[u]); v[u] = tab4[s[u] >> 2]; }
#pragma unroll
    for (int u = 0; u < U; u++) { const bool done = pj_group(v[u], tagsh[u], s[u], &cand[u], &pos[u]); walking[u] = on[u] & !done; cand[u] = on[u] ? cand[u] : EMPTY; more |= walking[u]; }
    while (__ballot(more)) {
      more = false;
#pragma unroll
      for (int u = 0; u < U; u++) if (walking[u]) { s[u] = (s[u] + 4) & M; walking[u] = !pj_group(tab4[s[u] >> 2], tagsh[u], s[u], &cand[u], &pos[u]); more |= walking[u]; }
    }
    Rec12 vb[U];
#pragma unroll
    for (int u = 0; u < U; u++) { vb[u] = Rec12{0, 0, 0}; if (cand[u] != EMPTY) vb[u] = br[cand[u] & IDX_MASK]; }
    uint32_t hrow[U]; bool hit[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      hit[u] = false; hrow[u] = 0;
      if (cand[u] == EMPTY) continue;
      if (vb[u].lo == rc[u].lo && vb[u].hi == rc[u].hi) { hit[u] = true; hrow[u] = vb[u].row; continue; }
      uint32_t s1 = (pos[u] + 1) & M, c = tab[s1];
      while (c != EMPTY) {
        if ((c & ~IDX_MASK) == tagsh[u]) { Rec12 w = br[c & IDX_MASK]; if (w.lo == rc[u].lo && w.hi == rc[u].hi) { hit[u] = true; hrow[u] = w.row; break; } }
        s1 = (s1 + 1) & M; c = tab[s1];
      }
    }
    if (EMIT == 0) {
#pragma unroll
      for (int u = 0; u < U; u++) if (hit[u]) found[rc[u].row] = hrow[u];
    } else {
      uint32_t mine = 0, tot = 0; uint32_t pre[U];
#pragma unroll
      for (int u = 0; u < U; u++) { uint64_t b = __ballot(hit[u]); pre[u] = tot + (uint32_t)__popcll(b & ((1ull << (threadIdx.x & 63)) - 1ull)); tot += (uint32_t)__popcll(b); mine += hit[u]; }
      uint32_t wbase = 0;
      if ((threadIdx.x & 63) == 0 && tot) wbase = atomicAdd(&hit_cursor, tot);
      wbase = __shfl(wbase, 0, 64);
#pragma unroll
      for (int u = 0; u < U; u++) if (hit[u]) hits[hbase + wbase + pre[u]] = ((uint64_t)rc[u].row << 32) | hrow[u];
    }
  }
  if (EMIT == 1) { __syncthreads(); if (threadIdx.x == 0) hcount[p] = hit_cursor; }
}


constexpr int CH_SHIFT = 17;          // chunk = 16 tiles of 8192 probe rows
// column sums of the hit matrix -> hits per chunk
__global__ void __launch_bounds__(256) k_chunk_tot(const uint32_t* hstart, int P, int NC, uint32_t* ctot) {
  const int c = blockIdx.x; uint32_t s = 0;
  for (int p = threadIdx.x; p < P; p += 256) s += hstart[(size_t)p * (NC + 1) + c + 1] - hstart[(size_t)p * (NC + 1) + c];
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
  __shared__ uint32_t w[4]; if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s; __syncthreads();
  if (threadIdx.x == 0) ctot[c] = w[0] + w[1] + w[2] + w[3];
}
// ------------------------------------------------------------------------------------------------ v4: tile-major offset matrices (coalesced reads of a tile's 2048 offsets)
// histogram: pre[t][p] = rows of partition p in the tiles of t's chunk in front of t (u32, tile-major), tot[c][p] = rows of partition p in chunk c (chunk-major)
template <int NT, int R, int G>
__global__ void __launch_bounds__(NT) k_hist4(const uint64_t* keys, long n, uint32_t P, long ntiles, long nchunks, uint32_t* pre /*[ntiles][P]*/, uint32_t* tot /*[nchunks][P]*/) {
  extern __shared__ uint32_t lds[];        // [P][G] u16
  const long t0 = (long)blockIdx.x * G;
  for (int x = threadIdx.x; x < (int)P * G / 2; x += NT) lds[x] = 0;
  uint64_t k[R], kn[R];
  { const long base = t0 * (long)(NT * R);
#pragma unroll
    for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; kn[q] = i < n ? keys[i] : 0; } }
  __syncthreads();
  for (int g = 0; g < G; g++) {
    const long base = (t0 + g) * (long)(NT * R);
    if (base >= n) break;
#pragma unroll
    for (int q = 0; q < R; q++) k[q] = kn[q];
    if (g + 1 < G) { const long nb2 = base + (long)(NT * R);
#pragma unroll
      for (int q = 0; q < R; q++) { long i = nb2 + (long)q * NT + threadIdx.x; kn[q] = i < n ? keys[i] : 0; } }
#pragma unroll
    for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; if (i < n) { uint32_t c = pid_of(mix64(k[q]), P) * G + g; atomicAdd(&lds[c >> 1], 1u << ((c & 1) * 16)); } }
  }
  __syncthreads();
  static_assert(G == 16, "two uint4 per partition");
  for (int p = threadIdx.x; p < (int)P; p += NT) {
    const uint4* s = (const uint4*)(lds + p * (G / 2)); uint4 a = s[0], b = s[1];
    uint32_t w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w }; uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (t0 + 2 * j < ntiles) pre[(t0 + 2 * j) * (long)P + p] = sum;
      sum += w[j] & 0xFFFFu;
      if (t0 + 2 * j + 1 < ntiles) pre[(t0 + 2 * j + 1) * (long)P + p] = sum;
      sum += w[j] >> 16;
    }
    tot[(long)blockIdx.x * P + p] = sum;
  }
}
// within-partition exclusive prefix of the chunk totals, in place (chunk-major storage), and the partition totals: 64 partitions per workgroup, one wave per range of chunks
__global__ void __launch_bounds__(1024) k_chunk_prefix(uint32_t* tot /*[nchunks][P] -> exclusive prefix along the chunks*/, long nchunks, uint32_t P, uint32_t* ptot /*[P]*/) {
  __shared__ uint32_t part[16][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63; const uint32_t p = blockIdx.x * 64 + lane;
  const long per = (nchunks + 15) / 16, c0 = wave * per, c1 = c0 + per < nchunks ? c0 + per : nchunks;
  uint32_t s = 0;
  if (p < P) for (long c = c0; c < c1; c++) s += tot[c * P + p];
  part[wave][lane] = s;
  __syncthreads();
  uint32_t run = 0; for (int w = 0; w < wave; w++) run += part[w][lane];
  if (wave == 15 && p < P) ptot[p] = run + s;
  if (p < P) for (long c = c0; c < c1; c++) { uint32_t v = tot[c * P + p]; tot[c * P + p] = run; run += v; }
}
// pstart[p] = exclusive scan of ptot (P <= 2048, one workgroup), pstart[P] = total
__global__ void __launch_bounds__(1024) k_pstart(const uint32_t* ptot, uint32_t P, uint32_t* pstart) {
  __shared__ uint32_t wsum[16];
  uint32_t a = threadIdx.x * 2 < P ? ptot[threadIdx.x * 2] : 0, b = threadIdx.x * 2 + 1 < P ? ptot[threadIdx.x * 2 + 1] : 0, s = a + b, inc = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if ((threadIdx.x & 63) >= d) inc += o; }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint32_t run = inc - s, tot = 0; for (int w = 0; w < 16; w++) { if (w < (int)(threadIdx.x >> 6)) run += wsum[w]; tot += wsum[w]; }
  if (threadIdx.x * 2 < P) pstart[threadIdx.x * 2] = run;
  if (threadIdx.x * 2 + 1 < P) pstart[threadIdx.x * 2 + 1] = run + a;
  if (threadIdx.x == 0) pstart[P] = tot;
}

template <int NT, int R, int ROUNDS, int G>
__global__ void __launch_bounds__(NT, ROUNDS >= 2 ? 2048 / 256 : NT / 256) k_scatter4(const uint64_t* keys, long n, uint32_t P, long ntiles, const uint32_t* pre, const uint32_t* cpre, const uint32_t* pstart, Rec12* out) {
  extern __shared__ uint32_t lds[];
  constexpr int TILE = NT * R, PIECE = TILE / ROUNDS;
  uint32_t* cnt = lds; uint32_t* delta = lds + P; uint16_t* spid = (uint16_t*)(lds + 2 * P); uint16_t* slidx = spid + PIECE;
  uint64_t* skey = (uint64_t*)(((uintptr_t)(slidx + PIECE) + 7) & ~(uintptr_t)7);
  __shared__ uint32_t wsum[NT / 64]; __shared__ uint32_t moved_sh;
  constexpr int PER = 2048 / NT < 1 ? 1 : 2048 / NT;
  const long per = (ntiles + 7) / 8; const long t = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (t >= ntiles || (long)(blockIdx.x >> 3) >= per) return;
  const long base = t * (long)TILE; const long chunk = t / G;
  uint64_t k[R]; uint32_t pid[R], rk[R]; bool on[R]; uint32_t gc[PER];
#pragma unroll
  for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; on[q] = i < n; k[q] = on[q] ? keys[i] : 0; }
#pragma unroll
  for (int jj = 0; jj < PER; jj++) { int p = threadIdx.x * PER + jj; gc[jj] = p < (int)P ? pre[t * (long)P + p] + cpre[chunk * (long)P + p] + pstart[p] : 0; }
  for (int p = threadIdx.x; p < (int)P; p += NT) cnt[p] = 0;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < R; q++) { pid[q] = pid_of(mix64(k[q]), P); rk[q] = on[q] ? atomicAdd(&cnt[pid[q]], 1u) : 0; }
  __syncthreads();
  {
    uint32_t loc[PER]; uint32_t s = 0;
#pragma unroll
    for (int jj = 0; jj < PER; jj++) { int p = threadIdx.x * PER + jj; loc[jj] = p < (int)P ? cnt[p] : 0; s += loc[jj]; }
    uint32_t inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if ((threadIdx.x & 63) >= d) inc += o; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t run = inc - s; for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += wsum[w];
#pragma unroll
    for (int jj = 0; jj < PER; jj++) { int p = threadIdx.x * PER + jj; if (p < (int)P) { cnt[p] = run; delta[p] = gc[jj] - run; run += loc[jj]; } }
    if (threadIdx.x == NT - 1) moved_sh = run;
  }
  __syncthreads();
  const uint32_t moved = moved_sh;
  uint32_t spos[R];
#pragma unroll
  for (int q = 0; q < R; q++) spos[q] = on[q] ? cnt[pid[q]] + rk[q] : 0xFFFFFFFFu;
#pragma unroll 1
  for (int h = 0; h < ROUNDS; h++) {
    const uint32_t lo = (uint32_t)h * PIECE;
    if (lo >= moved) break;
    if (h) __syncthreads();
#pragma unroll
    for (int q = 0; q < R; q++) { const uint32_t s = spos[q] - lo; if (s < (uint32_t)PIECE) { spid[s] = (uint16_t)pid[q]; slidx[s] = (uint16_t)(q * NT + threadIdx.x); skey[s] = k[q]; } }
    __syncthreads();
    const uint32_t m = moved - lo < (uint32_t)PIECE ? moved - lo : (uint32_t)PIECE;
    for (uint32_t i = threadIdx.x; i < m; i += NT) {
      const uint32_t pos = delta[spid[i]] + lo + i; const uint64_t v = skey[i];
      out[pos] = Rec12{ (uint32_t)v, (uint32_t)(v >> 32), (uint32_t)(base + slidx[i]) };
    }
  }
}

// join, ordered emission; MINW = waves per SIMD the register allocation must allow (8 = two 1024-thread workgroups per CU)
template <int NT, int U, int MINW, int PF>
__global__ void __launch_bounds__(NT, MINW) k_join4(const Rec12* brec, const uint32_t* bstart, const Rec12* prec, const uint32_t* pstart, int sbits, int NC, int chs,
                                                   uint64_t* hits, uint32_t* hstart /*[P][NC + 1]*/) {
  extern __shared__ uint4 tab4[];
  uint32_t* const tab = (uint32_t*)tab4;
  __shared__ uint32_t wc2[2][64]; int par = 0; uint32_t run_reg = 0;
  const uint32_t Sl = 1u << sbits, M = Sl - 1;
  const int p = blockIdx.x; const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (uint32_t s = threadIdx.x; s < Sl; s += NT) tab[s] = EMPTY;
  __syncthreads();
  const uint32_t b0 = bstart[p], nb = bstart[p + 1] - b0;
  const Rec12* br = brec + b0;
  for (uint32_t j = threadIdx.x; j < nb; j += NT) {
    Rec12 k = br[j]; uint32_t s, tagsh; pj_hash(k.lo, k.hi, M, sbits, &s, &tagsh); const uint32_t ent = tagsh | j;
    for (;;) { uint32_t old = atomicCAS(&tab[s], EMPTY, ent); if (old == EMPTY) break; s = (s + 1) & M; }
  }
  __syncthreads();
  const uint32_t q0 = pstart[p], q1 = pstart[p + 1];
  uint32_t* const hs = hstart + (size_t)p * (NC + 1);
  Rec12 rn[U];
  if (PF) {
#pragma unroll
    for (int u = 0; u < U; u++) { uint32_t i = q0 + threadIdx.x + u * NT; rn[u] = prec[i < q1 ? i : (q1 ? q1 - 1 : 0)]; }
  }
  for (uint32_t i0 = q0 + threadIdx.x; i0 - threadIdx.x < q1; i0 += NT * U) {
    Rec12 rc[U]; bool on[U];
#pragma unroll
    for (int u = 0; u < U; u++) { uint32_t i = i0 + u * NT; on[u] = i < q1; if (PF) rc[u] = rn[u]; else rc[u] = prec[on[u] ? i : q1 - 1]; }
    if (PF) { const uint32_t i2 = i0 + NT * U;
#pragma unroll
      for (int u = 0; u < U; u++) { uint32_t i = i2 + u * NT; rn[u] = prec[i < q1 ? i : q1 - 1]; } }
    uint32_t s[U], tagsh[U], cand[U], pos[U]; uint4 v[U]; bool walking[U]; bool more = false;
#pragma unroll
    for (int u = 0; u < U; u++) { pj_hash(rc[u].lo, rc[u].hi, M, sbits, &s[u], &tagsh[u]); v[u] = tab4[s[u] >> 2]; }
#pragma unroll
    for (int u = 0; u < U; u++) { const bool done = pj_group(v[u], tagsh[u], s[u], &cand[u], &pos[u]); walking[u] = on[u] & !done; cand[u] = on[u] ? cand[u] : EMPTY; more |= walking[u]; }
    while (__ballot(more)) {
      more = false;
#pragma unroll
      for (int u = 0; u < U; u++) if (walking[u]) { s[u] = (s[u] + 4) & M; walking[u] = !pj_group(tab4[s[u] >> 2], tagsh[u], s[u], &cand[u], &pos[u]); more |= walking[u]; }
    }
    Rec12 vb[U];
#pragma unroll
    for (int u = 0; u < U; u++) { vb[u] = Rec12{0, 0, 0}; if (cand[u] != EMPTY) vb[u] = br[cand[u] & IDX_MASK]; }
    uint32_t hrow[U]; bool hit[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      hit[u] = false; hrow[u] = 0;
      if (cand[u] == EMPTY) continue;
      if (vb[u].lo == rc[u].lo && vb[u].hi == rc[u].hi) { hit[u] = true; hrow[u] = vb[u].row; continue; }
      uint32_t s1 = (pos[u] + 1) & M, c = tab[s1];
      while (c != EMPTY) {
        if ((c & ~IDX_MASK) == tagsh[u]) { Rec12 w = br[c & IDX_MASK]; if (w.lo == rc[u].lo && w.hi == rc[u].hi) { hit[u] = true; hrow[u] = w.row; break; } }
        s1 = (s1 + 1) & M; c = tab[s1];
      }
    }
    uint32_t* const wc = wc2[par]; par ^= 1;
    uint32_t pre[U];
#pragma unroll
    for (int u = 0; u < U; u++) { uint64_t b = __ballot(hit[u]); pre[u] = (uint32_t)__popcll(b & ((1ull << lane) - 1ull)); if (lane == 0) wc[u * (NT / 64) + wave] = (uint32_t)__popcll(b); }
    __syncthreads();
    const uint32_t run0 = run_reg;
    uint32_t mine = lane < U * (NT / 64) ? wc[lane] : 0, inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    const uint32_t exl = inc - mine;
    run_reg = run0 + __shfl(inc, 63, 64);
    uint32_t wb[U];
#pragma unroll
    for (int u = 0; u < U; u++) wb[u] = __shfl(exl, u * (NT / 64) + wave, 64);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t o = run0 + wb[u] + pre[u];
      if (hit[u]) hits[(size_t)q0 + o] = ((uint64_t)rc[u].row << 32) | hrow[u];
      const uint32_t i = i0 + u * NT; const int c = (int)(rc[u].row >> chs);
      int cp = __shfl_up(c, 1, 64);
      if (lane == 0) cp = i > q0 && on[u] ? (int)(prec[i - 1].row >> chs) : -1;
      if (on[u] && c != cp) for (int x = cp + 1; x <= c; x++) hs[x] = o;
    }
  }
  if (threadIdx.x == 0) { const int cl = q1 > q0 ? (int)(prec[q1 - 1].row >> chs) : -1; const uint32_t tot = run_reg; for (int x = cl + 1; x <= NC; x++) hs[x] = tot; }
}
// restore order: one workgroup per chunk; 16 lanes per run, four runs per wave instruction; rank by probe row through a bitmap of the chunk's rows
template <int NT>
__global__ void __launch_bounds__(NT) k_restore4(const uint64_t* hits, const uint32_t* pstart, const uint32_t* hstart, int P, int NC, const uint32_t* coff, uint64_t* out) {
  __shared__ uint32_t bits[1 << (CH_SHIFT - 5)]; __shared__ uint32_t pref[1 << (CH_SHIFT - 5)]; __shared__ uint32_t rsrc[2048]; __shared__ uint16_t rlen[2048]; __shared__ uint32_t wsum[NT / 64];
  const int c = blockIdx.x; const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int x = threadIdx.x; x < (1 << (CH_SHIFT - 5)); x += NT) bits[x] = 0;
  for (int p = threadIdx.x; p < P; p += NT) { const uint32_t a = hstart[(size_t)p * (NC + 1) + c], b = hstart[(size_t)p * (NC + 1) + c + 1]; rsrc[p] = pstart[p] + a; rlen[p] = (uint16_t)(b - a); }
  __syncthreads();
  const uint32_t rmask = (1u << CH_SHIFT) - 1u;
  constexpr int NW = NT / 64, E = 8;      // E groups of four runs per wave and step, every load of a step issued before the first use
  for (int g0 = wave * 4; g0 < P; g0 += NW * 4 * E) {
    uint64_t h[E]; uint32_t n_[E];
#pragma unroll
    for (int e = 0; e < E; e++) { const int p = g0 + e * NW * 4 + (lane >> 4); n_[e] = 0; h[e] = ~0ull;
      if (p < P) { n_[e] = rlen[p]; if ((uint32_t)(lane & 15) < n_[e]) h[e] = hits[(size_t)rsrc[p] + (lane & 15)]; } }
#pragma unroll
    for (int e = 0; e < E; e++) {
      if ((uint32_t)(lane & 15) < n_[e]) { const uint32_t rr = (uint32_t)(h[e] >> 32) & rmask; atomicOr(&bits[rr >> 5], 1u << (rr & 31)); }
      if (n_[e] > 16) { const int p = g0 + e * NW * 4 + (lane >> 4); const uint64_t* src = hits + rsrc[p];
        for (uint32_t j = 16 + (lane & 15); j < n_[e]; j += 16) { const uint32_t rr = (uint32_t)(src[j] >> 32) & rmask; atomicOr(&bits[rr >> 5], 1u << (rr & 31)); } }
    }
  }
  __syncthreads();
  { constexpr int WPT = (1 << (CH_SHIFT - 5)) / NT; uint32_t c2[WPT]; uint32_t s2 = 0;
#pragma unroll
    for (int j = 0; j < WPT; j++) { c2[j] = __popc(bits[threadIdx.x * WPT + j]); s2 += c2[j]; }
    uint32_t inc2 = s2;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc2, d, 64); if (lane >= d) inc2 += o; }
    if (lane == 63) wsum[wave] = inc2;
    __syncthreads();
    uint32_t r2 = inc2 - s2; for (int w = 0; w < wave; w++) r2 += wsum[w];
#pragma unroll
    for (int j = 0; j < WPT; j++) { pref[threadIdx.x * WPT + j] = r2; r2 += c2[j]; } }
  __syncthreads();
  uint64_t* const o = out + coff[c];
  for (int g0 = wave * 4; g0 < P; g0 += NW * 4 * E) {
    uint64_t h[E]; uint32_t n_[E];
#pragma unroll
    for (int e = 0; e < E; e++) { const int p = g0 + e * NW * 4 + (lane >> 4); n_[e] = 0; h[e] = ~0ull;
      if (p < P) { n_[e] = rlen[p]; if ((uint32_t)(lane & 15) < n_[e]) h[e] = hits[(size_t)rsrc[p] + (lane & 15)]; } }
#pragma unroll
    for (int e = 0; e < E; e++) {
      if ((uint32_t)(lane & 15) < n_[e]) { const uint32_t rr = (uint32_t)(h[e] >> 32) & rmask; o[pref[rr >> 5] + __popc(bits[rr >> 5] & ((1u << (rr & 31)) - 1u))] = h[e]; }
      if (n_[e] > 16) { const int p = g0 + e * NW * 4 + (lane >> 4); const uint64_t* src = hits + rsrc[p];
        for (uint32_t j = 16 + (lane & 15); j < n_[e]; j += 16) { const uint64_t hv = src[j]; const uint32_t rr = (uint32_t)(hv >> 32) & rmask; o[pref[rr >> 5] + __popc(bits[rr >> 5] & ((1u << (rr & 31)) - 1u))] = hv; } }
    }
  }
}


// restore order, staged: one workgroup per chunk of 2^chs probe rows (chs <= 15); the chunk's hits are gathered from every partition's list (4 lanes per
// run, 16 runs per wave instruction), ranked by probe row through a bitmap of the chunk's rows and leave through an LDS window of WIN hits in rank order
template <int NT, int WIN>
__global__ void __launch_bounds__(NT) k_restoreX(const uint64_t* hits, const uint32_t* pstart, const uint32_t* hstart, int P, int NC, int chs, const uint32_t* coff, uint32_t* out_probe, uint64_t* out_build) {
  __shared__ uint32_t bits[1024]; __shared__ uint32_t pref[1024]; __shared__ uint32_t rsrc[2048]; __shared__ uint16_t rlen[2048]; __shared__ uint32_t wsum[NT / 64]; __shared__ uint64_t stage[WIN];
  const long per = (NC + 7) / 8; const long cc = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (cc >= NC || (long)(blockIdx.x >> 3) >= per) return;
  const int c = (int)cc; const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nwords = 1 << (chs - 5);
  for (int x = threadIdx.x; x < nwords; x += NT) bits[x] = 0;
  for (int p = threadIdx.x; p < P; p += NT) { const uint32_t a = hstart[(size_t)p * (NC + 1) + c], b = hstart[(size_t)p * (NC + 1) + c + 1]; rsrc[p] = pstart[p] + a; rlen[p] = (uint16_t)(b - a); }
  __syncthreads();
  const uint32_t rmask = (1u << chs) - 1u;
  constexpr int NW = NT / 64, E = 8;
  for (int g0 = wave * 16; g0 < P; g0 += NW * 16 * E) {
    uint64_t h[E]; uint32_t n_[E];
#pragma unroll
    for (int e = 0; e < E; e++) { const int p = g0 + e * NW * 16 + (lane >> 2); n_[e] = 0; h[e] = ~0ull;
      if (p < P) { n_[e] = rlen[p]; if ((uint32_t)(lane & 3) < n_[e]) h[e] = hits[(size_t)rsrc[p] + (lane & 3)]; } }
#pragma unroll
    for (int e = 0; e < E; e++) {
      if ((uint32_t)(lane & 3) < n_[e]) { const uint32_t rr = (uint32_t)(h[e] >> 32) & rmask; atomicOr(&bits[rr >> 5], 1u << (rr & 31)); }
      if (n_[e] > 4) { const int p = g0 + e * NW * 16 + (lane >> 2); const uint64_t* src = hits + rsrc[p];
        for (uint32_t j = 4 + (lane & 3); j < n_[e]; j += 4) { const uint32_t rr = (uint32_t)(src[j] >> 32) & rmask; atomicOr(&bits[rr >> 5], 1u << (rr & 31)); } }
    }
  }
  __syncthreads();
  uint32_t tot;
  { uint32_t cw = threadIdx.x < (unsigned)nwords ? __popc(bits[threadIdx.x]) : 0, inc2 = cw;       // nwords <= 1024 = NT
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc2, d, 64); if (lane >= d) inc2 += o; }
    if (lane == 63) wsum[wave] = inc2;
    __syncthreads();
    uint32_t r2 = inc2 - cw; tot = 0; for (int w = 0; w < NW; w++) { if (w < wave) r2 += wsum[w]; tot += wsum[w]; }
    if (threadIdx.x < (unsigned)nwords) pref[threadIdx.x] = r2; }
  __syncthreads();
  const uint32_t o0 = coff[c];
  for (uint32_t lo = 0; lo < tot; lo += WIN) {
    if (lo) __syncthreads();
    for (int g0 = wave * 16; g0 < P; g0 += NW * 16 * E) {
      uint64_t h[E]; uint32_t n_[E];
#pragma unroll
      for (int e = 0; e < E; e++) { const int p = g0 + e * NW * 16 + (lane >> 2); n_[e] = 0; h[e] = ~0ull;
        if (p < P) { n_[e] = rlen[p]; if ((uint32_t)(lane & 3) < n_[e]) h[e] = hits[(size_t)rsrc[p] + (lane & 3)]; } }
#pragma unroll
      for (int e = 0; e < E; e++) {
        if ((uint32_t)(lane & 3) < n_[e]) { const uint32_t rr = (uint32_t)(h[e] >> 32) & rmask; const uint32_t rk = pref[rr >> 5] + __popc(bits[rr >> 5] & ((1u << (rr & 31)) - 1u)) - lo; if (rk < (uint32_t)WIN) stage[rk] = h[e]; }
        if (n_[e] > 4) { const int p = g0 + e * NW * 16 + (lane >> 2); const uint64_t* src = hits + rsrc[p];
          for (uint32_t j = 4 + (lane & 3); j < n_[e]; j += 4) { const uint64_t hv = src[j]; const uint32_t rr = (uint32_t)(hv >> 32) & rmask; const uint32_t rk = pref[rr >> 5] + __popc(bits[rr >> 5] & ((1u << (rr & 31)) - 1u)) - lo; if (rk < (uint32_t)WIN) stage[rk] = hv; } }
      }
    }
    __syncthreads();
    const uint32_t m = tot - lo < (uint32_t)WIN ? tot - lo : (uint32_t)WIN;
    for (uint32_t i = threadIdx.x; i < m; i += NT) { const uint64_t hv = stage[i]; out_probe[o0 + lo + i] = (uint32_t)(hv >> 32); out_build[o0 + lo + i] = (uint32_t)hv; }
  }
}
__global__ void k_checksum2(const uint32_t* op, const uint64_t* ob, long n, const uint64_t* bk, const uint64_t* pk, unsigned long long* out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  uint32_t pr = op[i]; uint32_t br = (uint32_t)ob[i];
  atomicAdd(&out[0], (unsigned long long)mix64(((uint64_t)pr << 32) | br));
  if (bk[br] != pk[pr]) atomicAdd(&out[1], 1ull);
  if (i && op[i - 1] >= pr) atomicAdd(&out[2], 1ull);
}


// ------------------------------------------------------------------------------------------------ v5: no barrier in the probe loop
// every wave owns a contiguous range of chunks (of 2^chs probe rows) of its partition = a contiguous slice of the partition's records (tiles lie in row order
// inside a partition; the slice bounds come from the partition pass's offsets) and emits its hits in record order behind the slice's first record.
// hstart[p][c] = position of chunk c's first hit (relative to pstart[p]); send[p][w] = end of wave w's hits
template <int NT, int U, int MINW>
__global__ void __launch_bounds__(NT, MINW) k_join5(const Rec12* brec, const uint32_t* bstart, const Rec12* prec, const uint32_t* pstart, int sbits, int NC, int chs, uint32_t P,
                                                   const uint32_t* pre, const uint32_t* cpre, long ntiles, uint64_t* hits, uint32_t* hstart /*[P][NC]*/, uint32_t* send /*[P][NT / 64]*/) {
  extern __shared__ uint4 tab4[];
  uint32_t* const tab = (uint32_t*)tab4;
  const uint32_t Sl = 1u << sbits, M = Sl - 1;
  const int p = blockIdx.x; const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NW = NT / 64;
  for (uint32_t s = threadIdx.x; s < Sl; s += NT) tab[s] = EMPTY;
  __syncthreads();
  const uint32_t b0 = bstart[p], nb = bstart[p + 1] - b0;
  const Rec12* br = brec + b0;
  for (uint32_t j = threadIdx.x; j < nb; j += NT) {
    Rec12 k = br[j]; uint32_t s, tagsh; pj_hash(k.lo, k.hi, M, sbits, &s, &tagsh); const uint32_t ent = tagsh | j;
    for (;;) { uint32_t old = atomicCAS(&tab[s], EMPTY, ent); if (old == EMPTY) break; s = (s + 1) & M; }
  }
  __syncthreads();
  const uint32_t q0 = pstart[p], q1 = pstart[p + 1];
  const int K = (NC + NW - 1) / NW, ca = wave * K, cb = ca + K < NC ? ca + K : NC;       // this wave's chunks [ca, cb)
  // record index of the first record of chunk c in this partition: offset of tile c << (chs - 13)
  auto rec_of_chunk = [&](int c) -> uint32_t { const long t = (long)c << (chs - 13); if (t >= ntiles) return q1; return q0 + pre[t * (long)P + p] + cpre[(t >> 4) * (long)P + p]; };
  uint32_t* const hs = hstart + (size_t)p * NC;
  if (ca >= NC) { if (lane == 0) send[(size_t)p * NW + wave] = q1 - q0; return; }
  const uint32_t r0 = rec_of_chunk(ca), r1 = cb < NC ? rec_of_chunk(cb) : q1;
  uint32_t run = r0 - q0;       // hits of this wave go to [r0 - q0, ...) of the partition's hit region
  int clast = ca - 1;           // wave-uniform: last chunk whose start has been written
  for (uint32_t i0 = r0 + lane; i0 - lane < r1; i0 += 64 * U) {
    Rec12 rc[U]; bool on[U];
#pragma unroll
    for (int u = 0; u < U; u++) { uint32_t i = i0 + u * 64; on[u] = i < r1; rc[u] = prec[on[u] ? i : r1 - 1]; }
    uint32_t s[U], tagsh[U], cand[U], pos[U]; uint4 v[U]; bool walking[U]; bool more = false;
#pragma unroll
    for (int u = 0; u < U; u++) { pj_hash(rc[u].lo, rc[u].hi, M, sbits, &s[u], &tagsh[u]); v[u] = tab4[s[u] >> 2]; }
#pragma unroll
    for (int u = 0; u < U; u++) { const bool done = pj_group(v[u], tagsh[u], s[u], &cand[u], &pos[u]); walking[u] = on[u] & !done; cand[u] = on[u] ? cand[u] : EMPTY; more |= walking[u]; }
    while (__ballot(more)) {
      more = false;
#pragma unroll
      for (int u = 0; u < U; u++) if (walking[u]) { s[u] = (s[u] + 4) & M; walking[u] = !pj_group(tab4[s[u] >> 2], tagsh[u], s[u], &cand[u], &pos[u]); more |= walking[u]; }
    }
    Rec12 vb[U];
#pragma unroll
    for (int u = 0; u < U; u++) { vb[u] = Rec12{0, 0, 0}; if (cand[u] != EMPTY) vb[u] = br[cand[u] & IDX_MASK]; }
    uint32_t hrow[U]; bool hit[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      hit[u] = false; hrow[u] = 0;
      if (cand[u] == EMPTY) continue;
      if (vb[u].lo == rc[u].lo && vb[u].hi == rc[u].hi) { hit[u] = true; hrow[u] = vb[u].row; continue; }
      uint32_t s1 = (pos[u] + 1) & M, c = tab[s1];
      while (c != EMPTY) {
        if ((c & ~IDX_MASK) == tagsh[u]) { Rec12 w = br[c & IDX_MASK]; if (w.lo == rc[u].lo && w.hi == rc[u].hi) { hit[u] = true; hrow[u] = w.row; break; } }
        s1 = (s1 + 1) & M; c = tab[s1];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint64_t b = __ballot(hit[u]);
      const uint32_t o = run + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
      if (hit[u]) hits[(size_t)q0 + o] = ((uint64_t)rc[u].row << 32) | hrow[u];
      // chunk starts: the first record of a chunk writes the start of every chunk since the previous record's
      const int c = on[u] ? (int)(rc[u].row >> chs) : cb;
      int cp = __shfl_up(c, 1, 64); if (lane == 0) cp = clast;
      if (on[u] && c != cp) for (int x = cp + 1; x <= c; x++) hs[x] = o;
      clast = __shfl(c, 63, 64); if (clast >= cb) { const uint64_t onb = __ballot(on[u]); clast = onb ? __shfl(c, 63 - __clzll(onb), 64) : cp; clast = __shfl(clast, 0, 64); }
      run += (uint32_t)__popcll(b);
    }
  }
  if (lane == 0) { for (int x = clast + 1; x < cb; x++) hs[x] = run; send[(size_t)p * NW + wave] = run; }
}
// hstart [P][NC] -> hT [NC][P] + lengths: len(p, c) = next start (or the owning wave's end) - start; chunk totals by atomics
__global__ void __launch_bounds__(1024) k_transpose_len(const uint32_t* hstart, const uint32_t* send, int P, int NC, int NW, uint32_t* hT /*[NC][P] starts*/, uint16_t* lT /*[NC][P]*/, uint32_t* ctot) {
  __shared__ uint32_t ts[32][33], tl[32][33];
  const int c0 = blockIdx.x * 32, p0 = blockIdx.y * 32; const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int K = (NC + NW - 1) / NW;
  { const int p = p0 + ty, c = c0 + tx;
    if (p < P && c < NC) { const uint32_t a = hstart[(size_t)p * NC + c]; const bool last = (c + 1) % K == 0 || c + 1 == NC; const uint32_t b = last ? send[(size_t)p * NW + c / K] : hstart[(size_t)p * NC + c + 1]; ts[ty][tx] = a; tl[ty][tx] = b - a; }
    else { ts[ty][tx] = 0; tl[ty][tx] = 0; } }
  __syncthreads();
  { const int c = c0 + ty, p = p0 + tx;
    uint32_t l = tl[tx][ty];
    if (c < NC && p < P) { hT[(size_t)c * P + p] = ts[tx][ty]; lT[(size_t)c * P + p] = (uint16_t)l; }
    for (int d = 16; d > 0; d >>= 1) l += __shfl_xor(l, d, 64);
    if (tx == 0 && c < NC && l) atomicAdd(&ctot[c], l); }
}
template <int NT, int WIN>
__global__ void __launch_bounds__(NT) k_restore5(const uint64_t* hits, const uint32_t* pstart, const uint32_t* hT, const uint16_t* lT, int P, int NC, int chs, const uint32_t* coff, uint32_t* out_probe, uint64_t* out_build) {
  __shared__ uint32_t bits[1024]; __shared__ uint32_t pref[1024]; __shared__ uint32_t rsrc[2048]; __shared__ uint16_t rlen[2048]; __shared__ uint32_t wsum[NT / 64]; __shared__ uint64_t stage[WIN];
  const long per = (NC + 7) / 8; const long cc = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (cc >= NC || (long)(blockIdx.x >> 3) >= per) return;
  const int c = (int)cc; const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nwords = 1 << (chs - 5);
  for (int x = threadIdx.x; x < nwords; x += NT) bits[x] = 0;
  for (int p = threadIdx.x; p < P; p += NT) { rsrc[p] = pstart[p] + hT[(size_t)c * P + p]; rlen[p] = lT[(size_t)c * P + p]; }
  __syncthreads();
  const uint32_t rmask = (1u << chs) - 1u;
  constexpr int NW = NT / 64, E = 8;
  for (int g0 = wave * 16; g0 < P; g0 += NW * 16 * E) {
    uint64_t h[E]; uint32_t n_[E];
#pragma unroll
    for (int e = 0; e < E; e++) { const int p = g0 + e * NW * 16 + (lane >> 2); n_[e] = 0; h[e] = ~0ull;
      if (p < P) { n_[e] = rlen[p]; if ((uint32_t)(lane & 3) < n_[e]) h[e] = hits[(size_t)rsrc[p] + (lane & 3)]; } }
#pragma unroll
    for (int e = 0; e < E; e++) {
      if ((uint32_t)(lane & 3) < n_[e]) { const uint32_t rr = (uint32_t)(h[e] >> 32) & rmask; atomicOr(&bits[rr >> 5], 1u << (rr & 31)); }
      if (n_[e] > 4) { const int p = g0 + e * NW * 16 + (lane >> 2); const uint64_t* src = hits + rsrc[p];
        for (uint32_t j = 4 + (lane & 3); j < n_[e]; j += 4) { const uint32_t rr = (uint32_t)(src[j] >> 32) & rmask; atomicOr(&bits[rr >> 5], 1u << (rr & 31)); } }
    }
  }
  __syncthreads();
  uint32_t tot;
  { uint32_t cw = threadIdx.x < (unsigned)nwords ? __popc(bits[threadIdx.x]) : 0, inc2 = cw;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc2, d, 64); if (lane >= d) inc2 += o; }
    if (lane == 63) wsum[wave] = inc2;
    __syncthreads();
    uint32_t r2 = inc2 - cw; tot = 0; for (int w = 0; w < NW; w++) { if (w < wave) r2 += wsum[w]; tot += wsum[w]; }
    if (threadIdx.x < (unsigned)nwords) pref[threadIdx.x] = r2; }
  __syncthreads();
  const uint32_t o0 = coff[c];
  for (uint32_t lo = 0; lo < tot; lo += WIN) {
    if (lo) __syncthreads();
    for (int g0 = wave * 16; g0 < P; g0 += NW * 16 * E) {
      uint64_t h[E]; uint32_t n_[E];
#pragma unroll
      for (int e = 0; e < E; e++) { const int p = g0 + e * NW * 16 + (lane >> 2); n_[e] = 0; h[e] = ~0ull;
        if (p < P) { n_[e] = rlen[p]; if ((uint32_t)(lane & 3) < n_[e]) h[e] = hits[(size_t)rsrc[p] + (lane & 3)]; } }
#pragma unroll
      for (int e = 0; e < E; e++) {
        if ((uint32_t)(lane & 3) < n_[e]) { const uint32_t rr = (uint32_t)(h[e] >> 32) & rmask; const uint32_t rk = pref[rr >> 5] + __popc(bits[rr >> 5] & ((1u << (rr & 31)) - 1u)) - lo; if (rk < (uint32_t)WIN) stage[rk] = h[e]; }
        if (n_[e] > 4) { const int p = g0 + e * NW * 16 + (lane >> 2); const uint64_t* src = hits + rsrc[p];
          for (uint32_t j = 4 + (lane & 3); j < n_[e]; j += 4) { const uint64_t hv = src[j]; const uint32_t rr = (uint32_t)(hv >> 32) & rmask; const uint32_t rk = pref[rr >> 5] + __popc(bits[rr >> 5] & ((1u << (rr & 31)) - 1u)) - lo; if (rk < (uint32_t)WIN) stage[rk] = hv; } }
      }
    }
    __syncthreads();
    const uint32_t m = tot - lo < (uint32_t)WIN ? tot - lo : (uint32_t)WIN;
    for (uint32_t i = threadIdx.x; i < m; i += NT) { const uint64_t hv = stage[i]; out_probe[o0 + lo + i] = (uint32_t)(hv >> 32); out_build[o0 + lo + i] = (uint32_t)hv; }
  }
}


// restore order v6: flattened gather (hit i of the chunk -> its run by a search over the prefix of the run lengths), values of a chunk that fits one window stay in registers
template <int NT, int K6>
__global__ void __launch_bounds__(NT) k_restore6(const uint64_t* hits, const uint32_t* pstart, const uint32_t* hT, const uint16_t* lT, int P, int NC, int chs, const uint32_t* coff, uint32_t* out_probe, uint64_t* out_build) {
  constexpr int WIN = NT * K6;
  __shared__ uint32_t bits[1024]; __shared__ uint32_t pref[1024]; __shared__ uint32_t rsrc[2048]; __shared__ uint32_t roff[2049]; __shared__ uint32_t wsum[NT / 64]; __shared__ uint64_t stage[WIN];
  const long per = (NC + 7) / 8; const long cc = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (cc >= NC || (long)(blockIdx.x >> 3) >= per) return;
  const int c = (int)cc; const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63; constexpr int NW = NT / 64;
  const int nwords = 1 << (chs - 5);
  for (int x = threadIdx.x; x < nwords; x += NT) bits[x] = 0;
  uint32_t tot;
  { constexpr int PER = 2048 / NT; uint32_t len[PER]; uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) { const int p = threadIdx.x * PER + j; len[j] = 0; if (p < P) { len[j] = lT[(size_t)c * P + p]; rsrc[p] = pstart[p] + hT[(size_t)c * P + p]; } s += len[j]; }
    uint32_t inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - s; tot = 0; for (int w = 0; w < NW; w++) { if (w < wave) run += wsum[w]; tot += wsum[w]; }
#pragma unroll
    for (int j = 0; j < PER; j++) { const int p = threadIdx.x * PER + j; if (p < P) roff[p] = run; run += len[j]; }
    if (threadIdx.x == 0) roff[P] = tot; }
  __syncthreads();
  auto src_of = [&](uint32_t i) -> size_t { int lo = 0, hi = P;      // last run with roff <= i (empty runs share their successor's offset)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (roff[mid] <= i) lo = mid; else hi = mid; }
    return (size_t)rsrc[lo] + (i - roff[lo]); };
  const uint32_t rmask = (1u << chs) - 1u;
  const bool one = tot <= (uint32_t)WIN;
  uint64_t hv[K6];
  for (uint32_t i0 = 0; i0 < tot; i0 += WIN) {
    size_t a[K6];
#pragma unroll
    for (int k = 0; k < K6; k++) { const uint32_t i = i0 + k * NT + threadIdx.x; a[k] = i < tot ? src_of(i) : (size_t)0; }
#pragma unroll
    for (int k = 0; k < K6; k++) { const uint32_t i = i0 + k * NT + threadIdx.x; hv[k] = i < tot ? hits[a[k]] : ~0ull; }
#pragma unroll
    for (int k = 0; k < K6; k++) if (hv[k] != ~0ull) { const uint32_t rr = (uint32_t)(hv[k] >> 32) & rmask; atomicOr(&bits[rr >> 5], 1u << (rr & 31)); }
  }
  __syncthreads();
  { uint32_t cw = threadIdx.x < (unsigned)nwords ? __popc(bits[threadIdx.x]) : 0, inc2 = cw;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc2, d, 64); if (lane >= d) inc2 += o; }
    __syncthreads();
    if (lane == 63) wsum[wave] = inc2;
    __syncthreads();
    uint32_t r2 = inc2 - cw; for (int w = 0; w < wave; w++) r2 += wsum[w];
    if (threadIdx.x < (unsigned)nwords) pref[threadIdx.x] = r2; }
  __syncthreads();
  const uint32_t o0 = coff[c];
  for (uint32_t lo = 0; lo < tot; lo += WIN) {            // window of ranks [lo, lo + WIN)
    if (lo) __syncthreads();
    if (one) {
#pragma unroll
      for (int k = 0; k < K6; k++) if (hv[k] != ~0ull) { const uint32_t rr = (uint32_t)(hv[k] >> 32) & rmask; stage[pref[rr >> 5] + __popc(bits[rr >> 5] & ((1u << (rr & 31)) - 1u))] = hv[k]; }
    } else {
      for (uint32_t i0 = 0; i0 < tot; i0 += WIN) {
        size_t a[K6]; uint64_t h2[K6];
#pragma unroll
        for (int k = 0; k < K6; k++) { const uint32_t i = i0 + k * NT + threadIdx.x; a[k] = i < tot ? src_of(i) : (size_t)0; }
#pragma unroll
        for (int k = 0; k < K6; k++) { const uint32_t i = i0 + k * NT + threadIdx.x; h2[k] = i < tot ? hits[a[k]] : ~0ull; }
#pragma unroll
        for (int k = 0; k < K6; k++) if (h2[k] != ~0ull) { const uint32_t rr = (uint32_t)(h2[k] >> 32) & rmask; const uint32_t rk = pref[rr >> 5] + __popc(bits[rr >> 5] & ((1u << (rr & 31)) - 1u)) - lo; if (rk < (uint32_t)WIN) stage[rk] = h2[k]; }
      }
    }
    __syncthreads();
    const uint32_t m = tot - lo < (uint32_t)WIN ? tot - lo : (uint32_t)WIN;
    for (uint32_t i = threadIdx.x; i < m; i += NT) { const uint64_t v = stage[i]; out_probe[o0 + lo + i] = (uint32_t)(v >> 32); out_build[o0 + lo + i] = (uint32_t)v; }
  }
}

__global__ void __launch_bounds__(256) k_found_count(const uint32_t* found, long n, uint32_t* counts) {
  long base = ((long)blockIdx.x * 256 + threadIdx.x) * 16; uint32_t c = 0;
  if (base + 16 <= n) { const uint4* p = (const uint4*)(found + base);
#pragma unroll
    for (int q = 0; q < 4; q++) { uint4 v = p[q]; c += (v.x != EMPTY) + (v.y != EMPTY) + (v.z != EMPTY) + (v.w != EMPTY); } }
  else for (long i = base; i < n; i++) c += found[i] != EMPTY;
  for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
  __shared__ uint32_t w[4]; if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = c; __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}
__global__ void __launch_bounds__(256) k_found_write(const uint32_t* found, long n, const uint32_t* offs, uint64_t* out) {
  long base = ((long)blockIdx.x * 256 + threadIdx.x) * 16; uint32_t v[16]; uint32_t c = 0;
  if (base + 16 <= n) { const uint4* p = (const uint4*)(found + base);
#pragma unroll
    for (int q = 0; q < 4; q++) { uint4 x = p[q]; v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w; } }
  else for (int q = 0; q < 16; q++) v[q] = base + q < n ? found[base + q] : EMPTY;
#pragma unroll
  for (int q = 0; q < 16; q++) c += v[q] != EMPTY;
  uint32_t inc = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if ((threadIdx.x & 63) >= d) inc += o; }
  __shared__ uint32_t w[4]; if ((threadIdx.x & 63) == 63) w[threadIdx.x >> 6] = inc; __syncthreads();
  uint32_t ex = offs[blockIdx.x] + inc - c; for (int i = 0; i < (int)(threadIdx.x >> 6); i++) ex += w[i];
#pragma unroll
  for (int q = 0; q < 16; q++) if (v[q] != EMPTY) { out[ex] = ((uint64_t)(base + q) << 32) | v[q]; ex++; }
}
// ragged hit segments -> dense array (one workgroup per partition)
__global__ void __launch_bounds__(256) k_hits_dense(const uint64_t* hits, const uint32_t* pstart, int S, uint32_t cap, const uint32_t* hcount, const uint32_t* hoff, uint64_t* dense) {
  const int p = blockIdx.x; const size_t hbase = S ? (size_t)p * S * cap : (size_t)pstart[p]; const uint32_t c = hcount[p], o = hoff[p];
  for (uint32_t i = threadIdx.x; i < c; i += 256) dense[o + i] = hits[hbase + i];
}
__global__ void k_checksum(const uint64_t* pairs, long n, const uint64_t* bk, const uint64_t* pk, unsigned long long* out /*[0] xor-sum, [1] key mismatches, [2] order violations*/) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  uint64_t v = pairs[i]; uint32_t pr = (uint32_t)(v >> 32), br = (uint32_t)v;
  atomicAdd(&out[0], (unsigned long long)mix64(v));
  if (bk[br] != pk[pr]) atomicAdd(&out[1], 1ull);
  if (i && (uint32_t)(pairs[i - 1] >> 32) >= pr) atomicAdd(&out[2], 1ull);
}

__global__ void k_starts(const uint32_t* goff, long ntiles, int P, long n, uint32_t* start) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) start[p] = goff[(long)p * ntiles];
  if (p == P) start[P] = (uint32_t)n;
}
__global__ void k_fill(uint64_t* k, long n, uint64_t seed) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) k[i] = mix64(seed + (uint64_t)i * 0x9E3779B97F4A7C15ull) >> 2; }
__global__ void k_pick(uint64_t* pk, long n, const uint64_t* bk, long nb) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n && i % 5 == 0) pk[i] = bk[mix64((uint64_t)i) % (uint64_t)nb];
}

struct Ev { hipEvent_t ev; Ev() { CK(hipEventCreate(&ev)); } void rec() { CK(hipEventRecord(ev)); } };
static float ms(Ev& a, Ev& b) { float m; CK(hipEventElapsedTime(&m, a.ev, b.ev)); return m; }

int main(int argc, char** argv) {
  long nb = argc > 1 ? atol(argv[1]) : 15000000, np = argc > 2 ? atol(argv[2]) : 150000000;
  uint64_t *bk, *pk; CK(hipMalloc(&bk, nb * 8)); CK(hipMalloc(&pk, np * 8));
  hipLaunchKernelGGL(k_fill, dim3((nb + 255) / 256), dim3(256), 0, 0, bk, nb, 1ull);
  hipLaunchKernelGGL(k_fill, dim3((np + 255) / 256), dim3(256), 0, 0, pk, np, 0x1234567ull << 20);
  hipLaunchKernelGGL(k_pick, dim3((np + 255) / 256), dim3(256), 0, 0, pk, np, bk, nb);
  CK(hipDeviceSynchronize());
  const long maxP = 2048; long max_cells = maxP * ((np + 4095) / 4096);
  uint32_t *counts, *goff; CK(hipMalloc(&counts, max_cells * 4)); CK(hipMalloc(&goff, max_cells * 4));
  size_t tmp_bytes = 0; CK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, counts, goff, (int)max_cells));
  { size_t sb = 0; uint64_t* d = nullptr; CK(hipcub::DeviceRadixSort::SortKeys(nullptr, sb, d, d, (int)(np / 4), 32, 60)); if (sb > tmp_bytes) tmp_bytes = sb; }
  void* tmp; CK(hipMalloc(&tmp, tmp_bytes));
  const size_t rec_slots = (size_t)np + (size_t)np / 8 + (1u << 22);
  Rec12 *brec, *prec; uint32_t *bstart, *pstart; CK(hipMalloc(&brec, (nb + 4096) * 12)); CK(hipMalloc(&prec, rec_slots * 12)); CK(hipMalloc(&bstart, (maxP + 1) * 4)); CK(hipMalloc(&pstart, (maxP + 1) * 4));
  uint32_t* found; CK(hipMalloc(&found, np * 4)); uint64_t* hits; CK(hipMalloc(&hits, rec_slots * 8));
  uint32_t *hcount, *hoff; CK(hipMalloc(&hcount, (maxP + 1) * 4)); CK(hipMalloc(&hoff, (maxP + 1) * 4));
  uint32_t* fcnt; long nfb = (np + 4095) / 4096; CK(hipMalloc(&fcnt, nfb * 4)); uint32_t* foff; CK(hipMalloc(&foff, nfb * 4));
  uint64_t *pairs, *pairs2; CK(hipMalloc(&pairs, np * 8)); CK(hipMalloc(&pairs2, np * 8));
  uint32_t *cursors, *overflow; const int maxS = 64; CK(hipMalloc(&cursors, (size_t)maxS * (maxP / 64) * 4096)); CK(hipMalloc(&overflow, 4));
  unsigned long long* chk; CK(hipMalloc(&chk, 32));
  const int LDSMAX = 160 * 1024 - 1024;
#define ATTR(K) CK(hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, LDSMAX))
  ATTR((k_hist<1024, 8, 16>)); ATTR((k_hist4<1024, 8, 16>)); ATTR((k_scatter4<1024, 8, 2, 16>)); ATTR((k_scatter4<1024, 8, 1, 16>)); ATTR((k_join5<1024, 4, 8>)); ATTR((k_join5<1024, 2, 8>)); ATTR((k_join5<1024, 6, 8>));  ATTR((k_hist2<1024, 8, 16>)); ATTR((k_scatter2<1024, 8, 0>)); ATTR((k_scatter2<1024, 8, 1>)); ATTR((k_scatter2<1024, 8, 2>)); ATTR((k_scatter2<1024, 8, 3>)); ATTR((k_hist<512, 8, 16>));
  ATTR((k_scatter<1024, 8, 0>)); ATTR((k_scatter<1024, 8, 1>)); ATTR((k_scatter<512, 8, 0>)); ATTR((k_scatter<512, 8, 1>)); ATTR((k_scatter<256, 8, 1>)); ATTR((k_scatter<512, 4, 1>));
  ATTR((k_join<1024, 4, 0>)); ATTR((k_join<1024, 4, 1>)); ATTR((k_join<512, 4, 1>)); ATTR((k_join<1024, 2, 1>)); ATTR((k_join<512, 8, 1>));

  auto report_pairs = [&](const char* what, const uint64_t* prs, long total) {
    CK(hipMemset(chk, 0, 32));
    if (total) hipLaunchKernelGGL(k_checksum, dim3((total + 255) / 256), dim3(256), 0, 0, prs, total, bk, pk, chk);
    unsigned long long h[3]; CK(hipMemcpy(h, chk, 24, hipMemcpyDeviceToHost));
    printf("    %s: %ld pairs, checksum %016llx, key mismatches %llu, order violations %llu\n", what, total, h[0], h[1], h[2]);
  };

  for (uint32_t P : {2048u, 1024u}) {
    int sbits = 10; while ((1u << sbits) < 2 * (nb / P) + 1024) sbits++;
    if (sbits > 15) sbits = 15;
    printf("==== P = %u, LDS table %u slots\n", P, 1u << sbits);
    // ---- build side, exact path (512 x 8)
    {
      const int NT = 512, R = 8, G = 16, TILE = NT * R; long ntiles = (nb + TILE - 1) / TILE, nh = (ntiles + G - 1) / G;
      hipLaunchKernelGGL((k_hist<512, 8, 16>), dim3(nh), dim3(NT), P * G * 4, 0, bk, nb, P, ntiles, counts);
      size_t tb = tmp_bytes; CK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, counts, goff, (int)((long)P * ntiles)));
      size_t lds = (size_t)P * 8 + (size_t)TILE * 12 + 16;
      hipLaunchKernelGGL((k_scatter<512, 8, 0>), dim3(((ntiles + 7) / 8) * 8), dim3(NT), lds, 0, bk, nb, P, ntiles, goff, 0, 0u, nullptr, nullptr, brec);
      hipLaunchKernelGGL(k_starts, dim3((P + 256) / 256), dim3(256), 0, 0, goff, ntiles, (int)P, nb, bstart);
      CK(hipDeviceSynchronize());
    }
    // ---- probe side, exact path
    auto exact = [&](auto nt_c, const char* what) {
      constexpr int NT = decltype(nt_c)::value; const int R = 8, G = 16, TILE = NT * R; long ntiles = (np + TILE - 1) / TILE, nh = (ntiles + G - 1) / G;
      Ev e0, e1, e2, e3;
      for (int rep = 0; rep < 2; rep++) {
        e0.rec();
        hipLaunchKernelGGL((k_hist<NT, 8, 16>), dim3(nh), dim3(NT), P * G * 4, 0, pk, np, P, ntiles, counts);
        e1.rec();
        size_t tb = tmp_bytes; CK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, counts, goff, (int)((long)P * ntiles)));
        e2.rec();
        size_t lds = (size_t)P * 8 + (size_t)TILE * 12 + 16;
        hipLaunchKernelGGL((k_scatter<NT, 8, 0>), dim3(((ntiles + 7) / 8) * 8), dim3(NT), lds, 0, pk, np, P, ntiles, goff, 0, 0u, nullptr, nullptr, prec);
        e3.rec(); CK(hipEventSynchronize(e3.ev)); CK(hipGetLastError());
      }
      hipLaunchKernelGGL(k_starts, dim3((P + 256) / 256), dim3(256), 0, 0, goff, ntiles, (int)P, np, pstart);
      CK(hipDeviceSynchronize());
      printf("  scatter E %-9s: hist %.3f  scan %.3f  scatter %.3f  = %.3f ms\n", what, ms(e0, e1), ms(e1, e2), ms(e2, e3), ms(e0, e3));
    };
    exact(std::integral_constant<int, 1024>{}, "1024x8");
    // ---- joins over the exact layout
    auto join_exact = [&](int variant, const char* what) {
      Ev e0, e1, e2, e3, e4, e5; size_t lds = (size_t)(1u << sbits) * 4; long total = 0;
      for (int rep = 0; rep < 2; rep++) {
        if (variant == 0) {
          e0.rec(); CK(hipMemsetAsync(found, 0xFF, np * 4, 0)); e1.rec();
          hipLaunchKernelGGL((k_join<1024, 4, 0>), dim3(P), dim3(1024), lds, 0, brec, bstart, prec, pstart, 0, 0u, nullptr, sbits, found, nullptr, nullptr);
          e2.rec();
          hipLaunchKernelGGL(k_found_count, dim3(nfb), dim3(256), 0, 0, found, np, fcnt);
          e3.rec(); size_t tb = tmp_bytes; CK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, fcnt, foff, (int)nfb)); e4.rec();
          hipLaunchKernelGGL(k_found_write, dim3(nfb), dim3(256), 0, 0, found, np, foff, pairs);
          e5.rec(); CK(hipEventSynchronize(e5.ev)); CK(hipGetLastError());
          uint32_t a, b; CK(hipMemcpy(&a, foff + nfb - 1, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b, fcnt + nfb - 1, 4, hipMemcpyDeviceToHost)); total = (long)a + b;
          if (rep) printf("  join F %s: memset %.3f  join %.3f  count %.3f  scan %.3f  write %.3f  = %.3f ms\n", what, ms(e0, e1), ms(e1, e2), ms(e2, e3), ms(e3, e4), ms(e4, e5), ms(e0, e5));
        } else {
          e0.rec();
          if (variant == 1) hipLaunchKernelGGL((k_join<1024, 4, 1>), dim3(P), dim3(1024), lds, 0, brec, bstart, prec, pstart, 0, 0u, nullptr, sbits, nullptr, hits, hcount);
          else if (variant == 2) hipLaunchKernelGGL((k_join<512, 4, 1>), dim3(P), dim3(512), lds, 0, brec, bstart, prec, pstart, 0, 0u, nullptr, sbits, nullptr, hits, hcount);
          else if (variant == 3) hipLaunchKernelGGL((k_join<1024, 2, 1>), dim3(P), dim3(1024), lds, 0, brec, bstart, prec, pstart, 0, 0u, nullptr, sbits, nullptr, hits, hcount);
          else hipLaunchKernelGGL((k_join<512, 8, 1>), dim3(P), dim3(512), lds, 0, brec, bstart, prec, pstart, 0, 0u, nullptr, sbits, nullptr, hits, hcount);
          e1.rec();
          size_t tb = tmp_bytes; CK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, hcount, hoff, (int)P + 1));
          hipLaunchKernelGGL(k_hits_dense, dim3(P), dim3(256), 0, 0, hits, pstart, 0, 0u, hcount, hoff, pairs2);
          e2.rec();
          uint32_t tt; CK(hipMemcpy(&tt, hoff + P, 4, hipMemcpyDeviceToHost)); total = tt;
          e3.rec();
          tb = tmp_bytes; CK(hipcub::DeviceRadixSort::SortKeys(tmp, tb, pairs2, pairs, (int)total, 32, 60));
          e4.rec(); CK(hipEventSynchronize(e4.ev)); CK(hipGetLastError());
          if (rep) printf("  join H %s: join %.3f  dense copy %.3f  sort(hipcub, 28 bits) %.3f  = %.3f ms\n", what, ms(e0, e1), ms(e1, e2), ms(e3, e4), ms(e0, e2) + ms(e3, e4));
        }
      }
      report_pairs(what, pairs, total);
    };
    join_exact(0, "found 1024x4");
    join_exact(1, "hits 1024x4");
    // ---- v4: tile-major offsets, two-round staging, ordered hit emission + bitmap-rank restore
    {
      constexpr int NT = 1024; const int R = 8, G = 16, TILE = NT * R; long ntiles = (np + TILE - 1) / TILE, nchunks = (ntiles + G - 1) / G; const int NC = (int)nchunks;
      uint32_t* pre = counts; uint32_t* cpre = goff; uint32_t* ptot = goff + (size_t)P * nchunks;
      uint32_t* hstart = ptot + P + 64; uint32_t* ctot = hstart + (size_t)P * (NC + 1); uint32_t* coff = ctot + NC + 1;
      Ev e0, e1, e2, e3, e4, e5, e6;
      for (int rounds : {2, 1}) for (int rep = 0; rep < 2; rep++) {
        e0.rec();
        hipLaunchKernelGGL((k_hist4<NT, 8, 16>), dim3(nchunks), dim3(NT), P * G * 2, 0, pk, np, P, ntiles, nchunks, pre, cpre);
        e1.rec();
        hipLaunchKernelGGL(k_chunk_prefix, dim3((P + 63) / 64), dim3(1024), 0, 0, cpre, nchunks, P, ptot);
        hipLaunchKernelGGL(k_pstart, dim3(1), dim3(1024), 0, 0, ptot, P, pstart);
        e2.rec();
        if (rounds == 2) hipLaunchKernelGGL((k_scatter4<NT, 8, 2, 16>), dim3(((ntiles + 7) / 8) * 8), dim3(NT), (size_t)P * 8 + (size_t)TILE / 2 * 12 + 16, 0, pk, np, P, ntiles, pre, cpre, pstart, prec);
        else hipLaunchKernelGGL((k_scatter4<NT, 8, 1, 16>), dim3(((ntiles + 7) / 8) * 8), dim3(NT), (size_t)P * 8 + (size_t)TILE * 12 + 16, 0, pk, np, P, ntiles, pre, cpre, pstart, prec);
        e3.rec(); CK(hipEventSynchronize(e3.ev)); CK(hipGetLastError());
        if (rep) printf("  scatter E4 rounds=%d: hist %.3f  prefix %.3f  scatter %.3f  = %.3f ms\n", rounds, ms(e0, e1), ms(e1, e2), ms(e2, e3), ms(e0, e3));
      }
      long total = 0;
      uint32_t* out_p = (uint32_t*)pairs2; uint64_t* out_b = pairs;
      for (int chs : {15, 14}) {
        const int NCx = (int)((np + (1l << chs) - 1) >> chs); const int NW = 16;
        uint32_t* hstart = ptot + P + 64; uint32_t* send = hstart + (size_t)P * (NCx + 1); uint32_t* hT = send + (size_t)P * NW; uint16_t* lT = (uint16_t*)(hT + (size_t)P * NCx);
        uint32_t* ctot = (uint32_t*)(lT + (size_t)P * NCx + 64); uint32_t* coff = ctot + NCx + 1;
        for (int variant = 0; variant < 3; variant++) for (int rep = 0; rep < 2; rep++) {
          e3.rec();
          size_t lds = (size_t)(1u << sbits) * 4;
          if (variant == 0) hipLaunchKernelGGL((k_join5<1024, 4, 8>), dim3(P), dim3(1024), lds, 0, brec, bstart, prec, pstart, sbits, NCx, chs, P, pre, cpre, ntiles, hits, hstart, send);
          else if (variant == 1) hipLaunchKernelGGL((k_join5<1024, 2, 8>), dim3(P), dim3(1024), lds, 0, brec, bstart, prec, pstart, sbits, NCx, chs, P, pre, cpre, ntiles, hits, hstart, send);
          else hipLaunchKernelGGL((k_join5<1024, 4, 8>), dim3(P), dim3(1024), lds, 0, brec, bstart, prec, pstart, sbits, NCx, chs, P, pre, cpre, ntiles, hits, hstart, send);
          e4.rec();
          CK(hipMemsetAsync(ctot, 0, (size_t)(NCx + 1) * 4, 0));
          hipLaunchKernelGGL(k_transpose_len, dim3((NCx + 31) / 32, (P + 31) / 32), dim3(1024), 0, 0, hstart, send, (int)P, NCx, NW, hT, lT, ctot);
          size_t tb = tmp_bytes; CK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, ctot, coff, NCx + 1));
          e5.rec();
          if (variant == 2) hipLaunchKernelGGL((k_restore5<1024, 6144>), dim3(((NCx + 7) / 8) * 8), dim3(1024), 0, 0, hits, pstart, hT, lT, (int)P, NCx, chs, coff, out_p, out_b);
          else if (variant == 1) hipLaunchKernelGGL((k_restore6<512, 8>), dim3(((NCx + 7) / 8) * 8), dim3(512), 0, 0, hits, pstart, hT, lT, (int)P, NCx, chs, coff, out_p, out_b);
          else hipLaunchKernelGGL((k_restore6<1024, 6>), dim3(((NCx + 7) / 8) * 8), dim3(1024), 0, 0, hits, pstart, hT, lT, (int)P, NCx, chs, coff, out_p, out_b);
          e6.rec(); CK(hipEventSynchronize(e6.ev)); CK(hipGetLastError());
          uint32_t tt; CK(hipMemcpy(&tt, coff + NCx, 4, hipMemcpyDeviceToHost)); total = tt;
          if (rep) printf("  join5 chs=%d variant %d: join %.3f  transpose + totals + scan %.3f  restore(v0: 6/1024x6, v1: 6/512x8, v2: 5) %.3f  = %.3f ms\n", chs, variant, ms(e3, e4), ms(e4, e5), ms(e5, e6), ms(e3, e6));
        }
        CK(hipMemset(chk, 0, 32));
        hipLaunchKernelGGL(k_checksum2, dim3((total + 255) / 256), dim3(256), 0, 0, out_p, out_b, total, bk, pk, chk);
        unsigned long long hh[3]; CK(hipMemcpy(hh, chk, 24, hipMemcpyDeviceToHost));
        printf("    v5: %ld pairs, checksum %016llx, key mismatches %llu, order violations %llu\n", total, hh[0], hh[1], hh[2]);
      }
    }
  }
  return 0;
}
