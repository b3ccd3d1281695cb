// device_utils.h -- gfx950 device helpers: hashing, bitmaps, wave64 / workgroup scans, column cells.
#pragma once
#include "dfgpu_internal.h"

namespace dfgpu {

using i128 = __int128;
using u128 = unsigned __int128;

constexpr int WAVE = 64;           // CDNA4 wavefront
constexpr int BLOCK = 256;         // 4 waves, one per SIMD

// ---------------------------------------------------------------- hashing (DESIGN.md "hash function")
__host__ __device__ inline uint64_t mix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ULL;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
  return x ^ (x >> 31);
}
// combine_hashes, common/src/hash_utils.rs:38-41
__host__ __device__ inline uint64_t combine_hashes(uint64_t l, uint64_t r) { return ((uint64_t)(17 * 37) + l) * 37 + r; }

// ---------------------------------------------------------------- bitmaps (u64 words, LSB first)
__device__ inline bool bit_get(const uint64_t* w, int64_t i) { return (w[i >> 6] >> (i & 63)) & 1; }
__device__ inline bool valid_at(const uint64_t* w, int64_t i) { return w == nullptr || bit_get(w, i); }

// ---------------------------------------------------------------- wave64 primitives
__device__ inline int lane_id() { return threadIdx.x & 63; }
__device__ inline uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }
__device__ inline uint64_t ballot64(bool p) { return __ballot(p); }
// inclusive scan across the wave
template <typename T> __device__ inline T wave_inclusive_sum(T v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { T o = __shfl_up(v, d, 64); if (lane_id() >= d) v += o; }
  return v;
}
template <typename T> __device__ inline T wave_sum(T v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
// exclusive scan over a 256-thread workgroup; returns the exclusive prefix and the block total.
// `lds` must hold >= 4 T; callers must not reuse it before the trailing barrier.
template <typename T> __device__ inline T block_exclusive_sum(T v, T* lds, T* total) {
  T inc = wave_inclusive_sum(v);
  int w = threadIdx.x >> 6;
  __syncthreads();
  if (lane_id() == 63) lds[w] = inc;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < BLOCK / WAVE; i++) { T x = lds[i]; if (i < w) base += x; tot += x; }
  *total = tot;
  return base + inc - v;
}

// ---------------------------------------------------------------- column cells
__device__ inline int64_t key_at(const void* keys, int32_t key_type, int64_t i) {
  switch (key_type) {
    case DFGPU_INT8: return ((const int8_t*)keys)[i];
    case DFGPU_INT16: return ((const int16_t*)keys)[i];
    case DFGPU_INT32: case DFGPU_DATE32: return ((const int32_t*)keys)[i];      // Date32 is a 4-byte value: without its own case it fell to the 8-byte default and read past the column
    case DFGPU_INT64: return ((const int64_t*)keys)[i];
    case DFGPU_UINT8: return ((const uint8_t*)keys)[i];
    case DFGPU_UINT16: return ((const uint16_t*)keys)[i];
    case DFGPU_UINT32: return ((const uint32_t*)keys)[i];
    default: return (int64_t)((const uint64_t*)keys)[i];
  }
}
// Resolve dictionary indirection; returns false when the cell is NULL, else *row indexes c.values.
__device__ inline bool cell_resolve(const ColView& c, int64_t i, int64_t* row) {
  if (c.keys) {
    if (!valid_at(c.key_validity, i)) return false;
    int64_t k = key_at(c.keys, c.key_type, i);
    if (!valid_at(c.validity, k)) return false;
    *row = k; return true;
  }
  if (!valid_at(c.validity, i)) return false;
  *row = i; return true;
}
__device__ inline uint64_t load_bits(const ColView& c, int64_t i) {   // value as zero/sign-extended 64-bit pattern used by hash
  switch (c.type) {
    case DFGPU_BOOL: return bit_get((const uint64_t*)c.values, i);
    case DFGPU_INT8: return (uint64_t)(int64_t)((const int8_t*)c.values)[i];
    case DFGPU_INT16: return (uint64_t)(int64_t)((const int16_t*)c.values)[i];
    case DFGPU_INT32: case DFGPU_DATE32: return (uint64_t)(int64_t)((const int32_t*)c.values)[i];
    case DFGPU_UINT8: return ((const uint8_t*)c.values)[i];
    case DFGPU_UINT16: return ((const uint16_t*)c.values)[i];
    case DFGPU_UINT32: case DFGPU_FLOAT32: return ((const uint32_t*)c.values)[i];
    default: return ((const uint64_t*)c.values)[i];     // INT64 / UINT64 / FLOAT64
  }
}
__device__ inline uint64_t hash_utf8(const uint8_t* p, int64_t len, uint64_t seed) {
  uint64_t h = mix64((uint64_t)len ^ seed);
  for (int64_t o = 0; o < len; o += 8) {
    uint64_t w = 0; int m = len - o < 8 ? (int)(len - o) : 8;
    for (int b = 0; b < m; b++) w |= (uint64_t)p[o + b] << (8 * b);
    h = mix64(h ^ w);
  }
  return h;
}
__device__ inline uint64_t cell_hash(const ColView& c, int64_t row, uint64_t seed) {   // row already resolved, non-null
  if (c.type == DFGPU_DECIMAL128) { const uint64_t* p = (const uint64_t*)c.values + 2 * row; return mix64(mix64(p[1] ^ seed) ^ p[0]); }
  if (c.type == DFGPU_UTF8) { int32_t o = c.offsets[row]; return hash_utf8((const uint8_t*)c.values + o, c.offsets[row + 1] - o, seed); }
  return mix64(load_bits(c, row) ^ seed);
}
// equality of two resolved non-null cells of the same logical type (floats: bit pattern == totalOrder equality)
__device__ inline bool cell_equal(const ColView& a, int64_t i, const ColView& b, int64_t j) {
  if (a.type == DFGPU_DECIMAL128) { const uint64_t* p = (const uint64_t*)a.values + 2 * i; const uint64_t* q = (const uint64_t*)b.values + 2 * j; return p[0] == q[0] && p[1] == q[1]; }
  if (a.type == DFGPU_UTF8) {
    int32_t oa = a.offsets[i], ob = b.offsets[j]; int32_t la = a.offsets[i + 1] - oa, lb = b.offsets[j + 1] - ob;
    if (la != lb) return false;
    const uint8_t* p = (const uint8_t*)a.values + oa; const uint8_t* q = (const uint8_t*)b.values + ob;
    for (int32_t k = 0; k < la; k++) if (p[k] != q[k]) return false;
    return true;
  }
  return load_bits(a, i) == load_bits(b, j);
}
// row hash over a key set, create_hashes semantics (NULL leaves the running hash unchanged)
__device__ inline uint64_t keyset_hash(const KeySet& ks, int64_t i, uint64_t seed, bool* any_null) {
  uint64_t h = 0; bool an = false;
  for (int c = 0; c < ks.n; c++) {
    int64_t r;
    if (!cell_resolve(ks.c[c], i, &r)) { an = true; continue; }
    uint64_t x = cell_hash(ks.c[c], r, seed);
    h = c == 0 ? x : combine_hashes(x, h);
  }
  *any_null = an;
  return h;
}
// eq_dyn_null over all key columns (hash_join.rs:1067-1118): NULL never equals unless null_equals_null
__device__ inline bool keyset_equal(const KeySet& a, int64_t i, const KeySet& b, int64_t j, bool null_equals_null) {
  for (int c = 0; c < a.n; c++) {
    int64_t ri, rj; bool va = cell_resolve(a.c[c], i, &ri), vb = cell_resolve(b.c[c], j, &rj);
    if (!va || !vb) { if (null_equals_null && !va && !vb) continue; return false; }
    if (!cell_equal(a.c[c], ri, b.c[c], rj)) return false;
  }
  return true;
}

}  // namespace dfgpu
