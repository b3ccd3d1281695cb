// sort.hip -- a13 SortExec / sort_batch (physical-plan/src/sorts/sort.rs:584-609) and the stable
// multi-split used by the join CSR build and by RepartitionExec.
//
// lexsort_to_indices is done as: order-preserving byte encoding of the sort columns (one plane per
// key byte, [byte][row] so a pass streams one n-byte plane) + stable LSD radix sort (8-bit digits)
// of the UInt32 row ids.  Key bytes never move; passes whose byte is constant over all rows are
// skipped using one up-front histogram of every plane.  Ranking inside a wave uses 64-wide ballots
// (match-any over the 8 digit bits), across the 4 waves of a workgroup a small LDS table.
// Stable => ties keep input order (the reference leaves tie order unspecified: sort_unstable_by,
// sorts/sort.rs:641).
#include "device_utils.h"
#include "radix_partition.h"
#include <algorithm>

namespace dfgpu {

constexpr int RS_MAX_BLOCKS = 4096;

struct DigitKeys { const uint32_t* keys; int shift; };                 // digit from a u32 key that moves with the value
struct DigitPlane { const uint8_t* plane; };                           // digit = plane[row id], only row ids move
__device__ inline uint32_t digit_of(const DigitKeys& d, uint32_t key, uint32_t) { return (key >> d.shift) & 0xFFu; }
__device__ inline uint32_t digit_of(const DigitPlane& d, uint32_t, uint32_t val) { return d.plane[val]; }

template <typename D>
__global__ void __launch_bounds__(BLOCK) k_rs_hist(D dg, const uint32_t* keys, const uint32_t* vals, int64_t n, int64_t chunk, int nb, uint32_t* hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0; __syncthreads();
  int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  for (int64_t i = lo + threadIdx.x; i < hi; i += BLOCK) atomicAdd(&h[digit_of(dg, keys ? keys[i] : 0u, vals[i])], 1u);
  __syncthreads();
  hist[(int64_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

// Stable scatter of one pass.  A workgroup walks its chunk in tiles of RS_ITEMS x 256 rows: every lane loads its RS_ITEMS elements up
// front (element q of the tile's q-th 256-row slab, so slab order = input order), each wave ranks its elements slab by slab with
// 64-wide match-any ballots while keeping per-digit counts of the slabs already ranked in LDS (waves touch only their own rows of the
// count table between the barriers); the thread that owns a digit then turns the tile's counts into start positions in (slab, wave)
// order, and every element lands at start[slab][wave][digit] + its rank inside (slab, wave): 4 barriers per 1024 rows.
constexpr int RS_ITEMS = 4;
template <typename D>
__global__ void __launch_bounds__(BLOCK) k_rs_scatter(D dg, const uint32_t* keys, const uint32_t* vals, int64_t n, int64_t chunk, int nb,
                                                      const uint32_t* offsets, uint32_t* out_keys, uint32_t* out_vals) {
  constexpr int NW = BLOCK / WAVE;
  __shared__ uint32_t running[256];
  __shared__ uint32_t cnt[RS_ITEMS][NW][256];            // rows with digit d in (slab q, wave w) of the current tile
  running[threadIdx.x] = offsets[(int64_t)threadIdx.x * nb + blockIdx.x];
#pragma unroll
  for (int q = 0; q < RS_ITEMS; q++)
#pragma unroll
    for (int w = 0; w < NW; w++) cnt[q][w][threadIdx.x] = 0;
  __syncthreads();
  int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  const int wave = threadIdx.x >> 6;
  for (int64_t t0 = lo; t0 < hi; t0 += (int64_t)BLOCK * RS_ITEMS) {
    uint32_t key[RS_ITEMS], val[RS_ITEMS], d[RS_ITEMS], rank[RS_ITEMS]; bool active[RS_ITEMS];
#pragma unroll
    for (int q = 0; q < RS_ITEMS; q++) {
      int64_t i = t0 + (int64_t)q * BLOCK + threadIdx.x; active[q] = i < hi;
      int64_t ic = active[q] ? i : hi - 1;
      val[q] = vals[ic]; key[q] = keys ? keys[ic] : 0u;
    }
#pragma unroll
    for (int q = 0; q < RS_ITEMS; q++) d[q] = digit_of(dg, key[q], val[q]);
#pragma unroll
    for (int q = 0; q < RS_ITEMS; q++) {
      uint64_t peers = ballot64(active[q]);
#pragma unroll
      for (int b = 0; b < 8; b++) { uint64_t m = ballot64((d[q] >> b) & 1u); peers &= ((d[q] >> b) & 1u) ? m : ~m; }
      rank[q] = __popcll(peers & lanemask_lt());
      if (active[q] && rank[q] == 0) cnt[q][wave][d[q]] = __popcll(peers);
    }
    __syncthreads();
    {                                                     // thread t owns digit t: counts -> start positions, in (slab, wave) order
      uint32_t run = running[threadIdx.x];
#pragma unroll
      for (int q = 0; q < RS_ITEMS; q++)
#pragma unroll
        for (int w = 0; w < NW; w++) { uint32_t c = cnt[q][w][threadIdx.x]; cnt[q][w][threadIdx.x] = run; run += c; }
      running[threadIdx.x] = run;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RS_ITEMS; q++) if (active[q]) {
      uint32_t pos = cnt[q][wave][d[q]] + rank[q];
      out_vals[pos] = val[q]; if (out_keys) out_keys[pos] = key[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RS_ITEMS; q++)
#pragma unroll
      for (int w = 0; w < NW; w++) cnt[q][w][threadIdx.x] = 0;
    __syncthreads();
  }
}

// Small inputs (a query's result rows): a pass is launch latency, not bandwidth -- three launches (histogram, scan, scatter) of a few microseconds each, 18 for six varying
// key bytes.  Here the scan is folded into the scatter: every workgroup turns the raw [digit][workgroup] counts into its own start positions (thread t owns digit t: it adds up
// its digit's counts over the workgroups in front -- at most a few hundred words -- and a 256-wide scan over the digit totals gives the digit's base): two launches per pass.
// (Counting the NEXT pass's histogram inside the scatter as well -- the output position names the workgroup that reads the element next -- was measured and dropped: a key
// byte with few distinct values sends a workgroup's 4096 global atomics to a handful of counters; 141 K rows: 0.39 ms against 0.19 ms for the three-launch passes.)
// Same stable ranking as k_rs_scatter.
__global__ void __launch_bounds__(BLOCK) k_rs_plane_pass(const uint8_t* plane, const uint8_t* next_plane, const uint32_t* vals, int64_t n, int64_t chunk, int nb,
                                                         const uint32_t* counts /*[256][nb] raw*/, uint32_t* next_counts /*[256][nb] zeroed, or null*/, uint32_t* out_vals) {
  constexpr int NW = BLOCK / WAVE;
  __shared__ uint32_t running[256];
  __shared__ uint32_t cnt[RS_ITEMS][NW][256];
  __shared__ uint32_t scan_lds[NW];
  {
    uint32_t before = 0, tot = 0;
    for (int b = 0; b < nb; b++) { const uint32_t c = counts[(int64_t)threadIdx.x * nb + b]; before += b < (int)blockIdx.x ? c : 0u; tot += c; }
    uint32_t all; const uint32_t base = block_exclusive_sum<uint32_t>(tot, scan_lds, &all);
    running[threadIdx.x] = base + before;
  }
#pragma unroll
  for (int q = 0; q < RS_ITEMS; q++)
#pragma unroll
    for (int w = 0; w < NW; w++) cnt[q][w][threadIdx.x] = 0;
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  const int wave = threadIdx.x >> 6;
  for (int64_t t0 = lo; t0 < hi; t0 += (int64_t)BLOCK * RS_ITEMS) {
    uint32_t val[RS_ITEMS], d[RS_ITEMS], rank[RS_ITEMS]; bool active[RS_ITEMS];
#pragma unroll
    for (int q = 0; q < RS_ITEMS; q++) { const int64_t i = t0 + (int64_t)q * BLOCK + threadIdx.x; active[q] = i < hi; val[q] = vals[active[q] ? i : hi - 1]; }
#pragma unroll
    for (int q = 0; q < RS_ITEMS; q++) d[q] = plane[val[q]];
#pragma unroll
    for (int q = 0; q < RS_ITEMS; q++) {
      uint64_t peers = ballot64(active[q]);
#pragma unroll
      for (int b = 0; b < 8; b++) { const uint64_t m = ballot64((d[q] >> b) & 1u); peers &= ((d[q] >> b) & 1u) ? m : ~m; }
      rank[q] = __popcll(peers & lanemask_lt());
      if (active[q] && rank[q] == 0) cnt[q][wave][d[q]] = __popcll(peers);
    }
    __syncthreads();
    {
      uint32_t run = running[threadIdx.x];
#pragma unroll
      for (int q = 0; q < RS_ITEMS; q++)
#pragma unroll
        for (int w = 0; w < NW; w++) { const uint32_t c = cnt[q][w][threadIdx.x]; cnt[q][w][threadIdx.x] = run; run += c; }
      running[threadIdx.x] = run;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RS_ITEMS; q++) if (active[q]) {
      const uint32_t pos = cnt[q][wave][d[q]] + rank[q];
      out_vals[pos] = val[q];
      if (next_counts) atomicAdd(&next_counts[(int64_t)next_plane[val[q]] * nb + (int64_t)(pos / chunk)], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RS_ITEMS; q++)
#pragma unroll
      for (int w = 0; w < NW; w++) cnt[q][w][threadIdx.x] = 0;
    __syncthreads();
  }
}

struct RadixPlan { int nb; int64_t chunk; };
static RadixPlan plan_for(int64_t n) {
  // chunks of 4096 rows.  (Shorter chunks for small inputs -- 1024 rows, so that a 141 K-row result sort runs on 138 workgroups instead of 35 -- were measured: the six
  // passes of the SF12.5 Q3 step 0.127 -> 0.160 ms; every workgroup of the scatter adds up its digit's counts over the workgroups in front, and that loop grows with them;
  // 2048- and 3072-row chunks: 0.123 / 0.119 ms, inside the run-to-run spread.)
  int64_t nb = (n + 4095) / 4096; if (nb < 1) nb = 1; if (nb > RS_MAX_BLOCKS) nb = RS_MAX_BLOCKS;
  int64_t chunk = (n + nb - 1) / nb; chunk = (chunk + BLOCK - 1) / BLOCK * BLOCK;
  nb = (n + chunk - 1) / chunk; if (nb < 1) nb = 1;
  return { (int)nb, chunk };
}

template <typename D>
static void radix_pass(dfgpu_ctx* ctx, D dg, const uint32_t* keys, const uint32_t* vals, uint32_t* out_keys, uint32_t* out_vals, int64_t n, uint32_t* hist, RadixPlan p) {
  KernelTimer kt_(ctx, "radix_pass");
  hipLaunchKernelGGL((k_rs_hist<D>), dim3(p.nb), dim3(BLOCK), 0, ctx->stream, dg, keys, vals, n, p.chunk, p.nb, hist);
  exclusive_scan_u32_inplace32(ctx, hist, (int64_t)256 * p.nb, nullptr);
  hipLaunchKernelGGL((k_rs_scatter<D>), dim3(p.nb), dim3(BLOCK), 0, ctx->stream, dg, keys, vals, n, p.chunk, p.nb, (const uint32_t*)hist, out_keys, out_vals);
  KERNEL_CHECK();
}

// up to four key-byte planes of every row packed into one u32 (byte j = plane[j][row]), 4 rows per lane
struct PackPlanes { int n; const uint8_t* plane[4]; };
__global__ void __launch_bounds__(BLOCK) k_pack_planes(PackPlanes pp, int64_t n, uint32_t* out) {
  int64_t i0 = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * 4;
  if (i0 >= n) return;
  uint32_t k[4] = {0, 0, 0, 0};
  if (i0 + 4 <= n) {
#pragma unroll
    for (int j = 0; j < 4; j++) if (j < pp.n) {                     // byte loads: plane b starts at byte b * n, any alignment
#pragma unroll
      for (int r = 0; r < 4; r++) k[r] |= (uint32_t)pp.plane[j][i0 + r] << (8 * j); }
#pragma unroll
    for (int r = 0; r < 4; r++) out[i0 + r] = k[r];
  } else {
    for (int64_t i = i0; i < n; i++) { uint32_t v = 0; for (int j = 0; j < pp.n; j++) v |= (uint32_t)pp.plane[j][i] << (8 * j); out[i] = v; }
  }
}
__global__ void __launch_bounds__(BLOCK) k_gather_u32(const uint32_t* src, const uint32_t* idx, int64_t n, uint32_t* out) {
  const int64_t base = (int64_t)blockIdx.x * BLOCK * 4 + threadIdx.x;
  uint32_t j[4], v[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { int64_t i = base + (int64_t)q * BLOCK; j[q] = idx[i < n ? i : n - 1]; }
#pragma unroll
  for (int q = 0; q < 4; q++) v[q] = src[j[q]];
#pragma unroll
  for (int q = 0; q < 4; q++) { int64_t i = base + (int64_t)q * BLOCK; if (i < n) out[i] = v[q]; }
}
static void gather_u32(dfgpu_ctx* ctx, const uint32_t* src, const uint32_t* idx, int64_t n, uint32_t* out) {
  hipLaunchKernelGGL(k_gather_u32, dim3(grid_for(n, BLOCK * 4)), dim3(BLOCK), 0, ctx->stream, src, idx, n, out);
  KERNEL_CHECK();
}

void radix_sort_pairs_u32(dfgpu_ctx* ctx, uint32_t* keys, uint32_t* vals, int64_t n, int bits) {
  if (n <= 1) return;
  if (n > 0xFFFFFFF0ll) fail(DFGPU_NOT_IMPLEMENTED, "sort above 2^32-16 rows");
  RadixPlan p = plan_for(n);
  BufferPtr tk = alloc_buffer(ctx, (size_t)n * 4), tv = alloc_buffer(ctx, (size_t)n * 4), hist = alloc_buffer(ctx, (size_t)256 * p.nb * 4);
  uint32_t *k0 = keys, *v0 = vals, *k1 = (uint32_t*)tk->ptr, *v1 = (uint32_t*)tv->ptr;
  int passes = (bits + 7) / 8;
  for (int ps = 0; ps < passes; ps++) {
    radix_pass(ctx, DigitKeys{ k0, ps * 8 }, k0, v0, k1, v1, n, (uint32_t*)hist->ptr, p);
    std::swap(k0, k1); std::swap(v0, v1);
  }
  if (k0 != keys) {
    HIP_CHECK(hipMemcpyAsync(keys, k0, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(vals, v0, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
  }
}

// Inputs one workgroup holds (<= 8192 rows: a TopK's candidates, a small result): EVERY varying plane's pass inside one launch.  The row ids live in LDS (two buffers of
// 32 KB), a pass is the stable counting sort of radix_partition.h's STABLE ranking -- a wave owns 512 consecutive elements, ballots rank a row inside its slab, a running
// count per (wave, digit) across the wave's slabs, a prefix over the waves and a 256-wide scan over the digits -- and costs four barriers instead of two launches
// (a Utf8 tie-break key has ~45 varying planes: 90 launches of 3-4 us each for 4 K candidate rows).
constexpr int OB_NT = 1024, OB_R = 8, OB_NW = OB_NT / WAVE, OB_MAX = OB_NT * OB_R;
__global__ void __launch_bounds__(OB_NT) k_rs_one_block(const uint8_t* __restrict__ planes, int n, int W, const uint32_t* __restrict__ varies, const uint32_t* __restrict__ in_vals, uint32_t* __restrict__ out_vals) {
  __shared__ uint32_t va[OB_MAX], vb[OB_MAX];
  __shared__ uint16_t wcnt[OB_NW * 256];
  __shared__ uint32_t cnt[256], wsum[4];
  uint32_t* cur = va; uint32_t* nxt = vb;
  for (int e = threadIdx.x; e < n; e += OB_NT) cur[e] = in_vals[e];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  for (int b = W - 1; b >= 0; b--) {          // least significant plane first
    if (!varies[b]) continue;
    const uint8_t* plane = planes + (int64_t)b * n;
    __syncthreads();                          // the previous pass's scatter (or the load above) is complete; wcnt is free
    for (int x = threadIdx.x; x < OB_NW * 128; x += OB_NT) ((uint32_t*)wcnt)[x] = 0;
    __syncthreads();
    uint32_t v[OB_R], d[OB_R], rk[OB_R]; bool on[OB_R];
#pragma unroll
    for (int q = 0; q < OB_R; q++) { const int e = wave * (OB_R * WAVE) + q * WAVE + lane; on[q] = e < n; v[q] = on[q] ? cur[e] : 0u; }
#pragma unroll
    for (int q = 0; q < OB_R; q++) d[q] = on[q] ? (uint32_t)plane[v[q]] : 0u;
#pragma unroll
    for (int q = 0; q < OB_R; q++) {
      uint64_t peers = ballot64(on[q]);
#pragma unroll
      for (int bit = 0; bit < 8; bit++) { const uint64_t mb = ballot64((d[q] >> bit) & 1u); peers &= ((d[q] >> bit) & 1u) ? mb : ~mb; }
      const uint32_t below = (uint32_t)__popcll(peers & lanemask_lt());
      uint32_t seen = 0;
      if (on[q] && below == 0) { uint16_t* wc = wcnt + wave * 256 + d[q]; seen = *wc; *wc = (uint16_t)(seen + (uint32_t)__popcll(peers)); }
      seen = __shfl(seen, peers ? __ffsll((unsigned long long)peers) - 1 : 0, 64);
      rk[q] = seen + below;
    }
    __syncthreads();
    uint32_t c = 0, inc = 0;
    if (threadIdx.x < 256) {
      uint32_t run = 0;
#pragma unroll
      for (int w = 0; w < OB_NW; w++) { const uint32_t x = wcnt[w * 256 + threadIdx.x]; wcnt[w * 256 + threadIdx.x] = (uint16_t)run; run += x; }
      c = run; inc = wave_inclusive_sum(c);
      if (lane == 63) wsum[wave] = inc;
    }
    __syncthreads();
    if (threadIdx.x < 256) { uint32_t run = inc - c; for (int w = 0; w < wave; w++) run += wsum[w]; cnt[threadIdx.x] = run; }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < OB_R; q++) if (on[q]) nxt[cnt[d[q]] + (uint32_t)wcnt[wave * 256 + d[q]] + rk[q]] = v[q];
    uint32_t* t = cur; cur = nxt; nxt = t;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n; e += OB_NT) out_vals[e] = cur[e];
}

// ---------------------------------------------------------------- order-preserving key encoding
struct SortCol { ColView v; int32_t byte_off; int32_t has_null_byte; int32_t descending; int32_t nulls_first; int32_t max_len; };   // max_len: Utf8 only
struct SortCols { int32_t n; SortCol c[MAX_KEYS]; };

__device__ inline void put_be(uint8_t* planes, int64_t n, int64_t row, int off, uint64_t v, int width, uint8_t inv) {
  for (int b = 0; b < width; b++) planes[(int64_t)(off + b) * n + row] = (uint8_t)(v >> (8 * (width - 1 - b))) ^ inv;
}
__global__ void __launch_bounds__(BLOCK) k_encode_sort_keys(SortCols sc, int64_t n, uint8_t* planes) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  for (int c = 0; c < sc.n; c++) {
    const SortCol& s = sc.c[c]; const ColView& v = s.v;
    int64_t r; bool ok = cell_resolve(v, i, &r);
    int off = s.byte_off;
    if (s.has_null_byte) { planes[(int64_t)off * n + i] = ok ? (s.nulls_first ? 1 : 0) : (s.nulls_first ? 0 : 1); off++; }
    int w = v.type == DFGPU_BOOL ? 1 : (v.type == DFGPU_UTF8 ? s.max_len + 4 : v.width);
    if (!ok) { for (int b = 0; b < w; b++) planes[(int64_t)(off + b) * n + i] = 0; continue; }
    uint8_t inv = s.descending ? 0xFF : 0x00;
    if (v.type == DFGPU_UTF8) {
      // bytes zero-padded to the longest string, then the length (big endian): byte-wise lexicographic order with a
      // prefix sorting before its extensions -- arrow-ord's order for Utf8 (memcmp, then length)
      int32_t o = v.offsets[r], len = v.offsets[r + 1] - o; const uint8_t* p = (const uint8_t*)v.values + o;
      for (int b = 0; b < s.max_len; b++) planes[(int64_t)(off + b) * n + i] = (uint8_t)((b < len ? p[b] : 0) ^ inv);
      put_be(planes, n, i, off + s.max_len, (uint32_t)len, 4, inv);
      continue;
    }
    switch (v.type) {
      case DFGPU_BOOL: put_be(planes, n, i, off, bit_get((const uint64_t*)v.values, r), 1, inv); break;
      case DFGPU_INT8: put_be(planes, n, i, off, (uint8_t)(((const uint8_t*)v.values)[r] ^ 0x80u), 1, inv); break;
      case DFGPU_INT16: put_be(planes, n, i, off, (uint16_t)(((const uint16_t*)v.values)[r] ^ 0x8000u), 2, inv); break;
      case DFGPU_INT32: case DFGPU_DATE32: put_be(planes, n, i, off, ((const uint32_t*)v.values)[r] ^ 0x80000000u, 4, inv); break;
      case DFGPU_INT64: put_be(planes, n, i, off, ((const uint64_t*)v.values)[r] ^ 0x8000000000000000ull, 8, inv); break;
      case DFGPU_UINT8: put_be(planes, n, i, off, ((const uint8_t*)v.values)[r], 1, inv); break;
      case DFGPU_UINT16: put_be(planes, n, i, off, ((const uint16_t*)v.values)[r], 2, inv); break;
      case DFGPU_UINT32: put_be(planes, n, i, off, ((const uint32_t*)v.values)[r], 4, inv); break;
      case DFGPU_UINT64: put_be(planes, n, i, off, ((const uint64_t*)v.values)[r], 8, inv); break;
      case DFGPU_FLOAT32: { uint32_t b = ((const uint32_t*)v.values)[r]; b ^= (b >> 31) ? 0xFFFFFFFFu : 0x80000000u; put_be(planes, n, i, off, b, 4, inv); break; }   // IEEE totalOrder
      case DFGPU_FLOAT64: { uint64_t b = ((const uint64_t*)v.values)[r]; b ^= (b >> 63) ? ~0ull : 0x8000000000000000ull; put_be(planes, n, i, off, b, 8, inv); break; }
      case DFGPU_DECIMAL128: { const uint64_t* p = (const uint64_t*)v.values + 2 * r; put_be(planes, n, i, off, p[1] ^ 0x8000000000000000ull, 8, inv); put_be(planes, n, i, off + 8, p[0], 8, inv); break; }
      default: break;
    }
  }
}
__global__ void __launch_bounds__(BLOCK) k_max_utf8_len(ColView v, int64_t n, unsigned int* out) {
  unsigned int m = 0;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    int64_t r; if (cell_resolve(v, i, &r)) { unsigned int len = (unsigned int)(v.offsets[r + 1] - v.offsets[r]); m = len > m ? len : m; }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { unsigned int o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
  if (lane_id() == 0 && m) atomicMax(out, m);
}
// varies[plane] = 1 when the plane holds more than one byte value (a constant plane needs no radix pass)
__global__ void __launch_bounds__(BLOCK) k_plane_varies(const uint8_t* planes, int64_t n, uint32_t* varies) {
  const uint8_t* p = planes + (int64_t)blockIdx.y * n;
  uint8_t first = p[0]; bool diff = false;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) diff |= p[i] != first;
  if (ballot64(diff) && lane_id() == 0) varies[blockIdx.y] = 1u;
}

// ---- TopK (SortExec with fetch, ≙ physical-plan/src/topk/mod.rs): radix SELECT over the same order-preserving byte planes, most
// significant varying plane first.  One step applies the previous plane's decision (digit < chosen: accepted into the top set;
// == chosen: still a candidate; > chosen: out) and histograms the next plane over the surviving candidates.  The host picks the digit
// holding the k-th row.  When few candidates are left they all join the accepted rows and that small set is sorted by the full path --
// same rows and same order as sorting everything and slicing (ties keep row order).
__global__ void __launch_bounds__(BLOCK) k_topk_step(const uint8_t* planes, int64_t n, int prev_plane, int prev_digit, int cur_plane, uint64_t* cand, uint64_t* accept, uint32_t* hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  int lane = lane_id();
  int64_t nw = (n + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6); w < nw; w += (int64_t)gridDim.x * (BLOCK / WAVE)) {
    int64_t i = w * 64 + lane;
    bool c = i < n && ((cand[w] >> lane) & 1ull), acc = false;
    if (prev_plane >= 0 && c) { int dg = planes[(int64_t)prev_plane * n + i]; if (dg < prev_digit) { acc = true; c = false; } else if (dg > prev_digit) c = false; }
    uint64_t cm = ballot64(c), am = ballot64(acc);
    if (lane == 0 && prev_plane >= 0) { cand[w] = cm; if (am) accept[w] |= am; }
    if (cur_plane >= 0 && cm) {
      int d = c ? (int)planes[(int64_t)cur_plane * n + i] : -1;
      uint64_t left = cm;
      while (left) {                                            // one LDS add per distinct digit of the wave (a near-constant plane would serialise 64 ways)
        int src = __ffsll((long long)left) - 1; int d0 = __shfl(d, src, 64);
        uint64_t same = ballot64(d == d0);
        if (lane == 0) atomicAdd(&h[d0], (uint32_t)__popcll(same));
        left &= ~same;
      }
    }
  }
  __syncthreads();
  if (cur_plane >= 0 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void k_or_words(const uint64_t* a, const uint64_t* b, uint64_t* out, int64_t nw) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nw) out[i] = a[i] | b[i];
}


// ---------------------------------------------------------------- packed-key path (large inputs, fixed-width keys)
// Every sort column is reduced to an unsigned offset inside its value range (ascending: v - min, descending: max - v; NULLs take the
// slot below / above the range as nulls_first says), the offsets are concatenated into ONE u64 (first column in the top bits): comparing
// the packed keys == comparing the rows lexicographically.  Only the bits that vary are sorted: LSD passes over the packed key
// through the stable one-pass partition of radix_partition.h (digits of up to 8 bits), the (key, row id) pair moving with every pass.
struct PkCol { const void* v; const uint64_t* valid; int32_t type; int32_t desc; int32_t nulls_first; int32_t shift; uint64_t lo_bits, hi_bits; uint64_t span; int32_t bits; void* sorted_dst; };   // lo/hi: order-preserving 128-bit minimum as two words (hi only for Decimal128)
struct PkCols { int32_t n; PkCol c[MAX_KEYS]; };
__device__ inline void pk_order_bits(const void* v, int32_t type, int64_t i, uint64_t* hi, uint64_t* lo) {      // value -> unsigned 128-bit pattern whose order is the value order
  *hi = 0;
  switch (type) {
    case DFGPU_INT8: *lo = (uint64_t)((int64_t)((const int8_t*)v)[i]) ^ 0x8000000000000000ull; break;
    case DFGPU_INT16: *lo = (uint64_t)((int64_t)((const int16_t*)v)[i]) ^ 0x8000000000000000ull; break;
    case DFGPU_INT32: case DFGPU_DATE32: *lo = (uint64_t)((int64_t)((const int32_t*)v)[i]) ^ 0x8000000000000000ull; break;
    case DFGPU_INT64: *lo = ((const uint64_t*)v)[i] ^ 0x8000000000000000ull; break;
    case DFGPU_UINT8: *lo = ((const uint8_t*)v)[i]; break; case DFGPU_UINT16: *lo = ((const uint16_t*)v)[i]; break;
    case DFGPU_UINT32: *lo = ((const uint32_t*)v)[i]; break; case DFGPU_UINT64: *lo = ((const uint64_t*)v)[i]; break;
    case DFGPU_FLOAT32: { uint32_t b = ((const uint32_t*)v)[i]; b ^= (b >> 31) ? 0xFFFFFFFFu : 0x80000000u; *lo = b; break; }          // IEEE totalOrder (NaN above every number)
    case DFGPU_FLOAT64: { uint64_t b = ((const uint64_t*)v)[i]; b ^= (b >> 63) ? ~0ull : 0x8000000000000000ull; *lo = b; break; }
    default: { const uint64_t* p = (const uint64_t*)v + 2 * i; *lo = p[0]; *hi = p[1] ^ 0x8000000000000000ull; break; }                   // DECIMAL128
  }
}
__device__ inline bool pk_less(uint64_t ah, uint64_t al, uint64_t bh, uint64_t bl) { return ah < bh || (ah == bh && al < bl); }
// per column: minimum and maximum order pattern over the valid rows -> out[4 c .. 4 c + 3] = (min hi, min lo, max hi, max lo)
__global__ void __launch_bounds__(BLOCK) k_pk_minmax(PkCols pc, int64_t n, int64_t step, unsigned long long* out) {       // step > 1: every step-th row (a sample)
  __shared__ unsigned long long sh[BLOCK / WAVE][4];
  for (int c = 0; c < MAX_KEYS; c++) {
    if (c >= pc.n) break;
    uint64_t mnh = ~0ull, mnl = ~0ull, mxh = 0, mxl = 0;
    for (int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * step; i < n; i += (int64_t)gridDim.x * BLOCK * step) {
      if (!valid_at(pc.c[c].valid, i)) continue;
      uint64_t h, l; pk_order_bits(pc.c[c].v, pc.c[c].type, i, &h, &l);
      if (pk_less(h, l, mnh, mnl)) { mnh = h; mnl = l; }
      if (pk_less(mxh, mxl, h, l)) { mxh = h; mxl = l; }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      uint64_t oh = __shfl_xor(mnh, d, 64), ol = __shfl_xor(mnl, d, 64); if (pk_less(oh, ol, mnh, mnl)) { mnh = oh; mnl = ol; }
      oh = __shfl_xor(mxh, d, 64); ol = __shfl_xor(mxl, d, 64); if (pk_less(mxh, mxl, oh, ol)) { mxh = oh; mxl = ol; }
    }
    if (lane_id() == 0) { sh[threadIdx.x >> 6][0] = mnh; sh[threadIdx.x >> 6][1] = mnl; sh[threadIdx.x >> 6][2] = mxh; sh[threadIdx.x >> 6][3] = mxl; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < BLOCK / WAVE; w++) { if (pk_less(sh[w][0], sh[w][1], mnh, mnl)) { mnh = sh[w][0]; mnl = sh[w][1]; } if (pk_less(mxh, mxl, sh[w][2], sh[w][3])) { mxh = sh[w][2]; mxl = sh[w][3]; } }
      // 128-bit min / max through two 64-bit atomics is not atomic as a pair: one slot per workgroup, reduced on the host
      unsigned long long* o = out + ((size_t)blockIdx.x * MAX_KEYS + c) * 4; o[0] = mnh; o[1] = mnl; o[2] = mxh; o[3] = mxl;
    }
    __syncthreads();
  }
}
// the workgroups' slots of k_pk_minmax folded into one (out = MAX_KEYS x 4 words): the host reads 256 bytes through the mailbox instead of 32 KB per column set
__global__ void __launch_bounds__(BLOCK) k_pk_minmax_fold(const unsigned long long* part, int nb, int ncols, unsigned long long* out) {
  __shared__ unsigned long long sh[BLOCK / WAVE][4];
  for (int c = 0; c < ncols; c++) {
    uint64_t mnh = ~0ull, mnl = ~0ull, mxh = 0, mxl = 0;
    for (int b = threadIdx.x; b < nb; b += BLOCK) { const unsigned long long* o = part + ((size_t)b * MAX_KEYS + c) * 4;
      if (pk_less(o[0], o[1], mnh, mnl)) { mnh = o[0]; mnl = o[1]; } if (pk_less(mxh, mxl, o[2], o[3])) { mxh = o[2]; mxl = o[3]; } }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      uint64_t oh = __shfl_xor(mnh, d, 64), ol = __shfl_xor(mnl, d, 64); if (pk_less(oh, ol, mnh, mnl)) { mnh = oh; mnl = ol; }
      oh = __shfl_xor(mxh, d, 64); ol = __shfl_xor(mxl, d, 64); if (pk_less(mxh, mxl, oh, ol)) { mxh = oh; mxl = ol; }
    }
    if (lane_id() == 0) { sh[threadIdx.x >> 6][0] = mnh; sh[threadIdx.x >> 6][1] = mnl; sh[threadIdx.x >> 6][2] = mxh; sh[threadIdx.x >> 6][3] = mxl; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < BLOCK / WAVE; w++) { if (pk_less(sh[w][0], sh[w][1], mnh, mnl)) { mnh = sh[w][0]; mnl = sh[w][1]; } if (pk_less(mxh, mxl, sh[w][2], sh[w][3])) { mxh = sh[w][2]; mxl = sh[w][3]; } }
      out[4 * c] = mnh; out[4 * c + 1] = mnl; out[4 * c + 2] = mxh; out[4 * c + 3] = mxl;
    }
    __syncthreads();
  }
}
// ib = 0: keys[i] = packed key, idx[i] = i.  ib > 0 ("word mode": key bits + row-number bits fit 64): keys[i] = packed key << ib | i, one 8-byte record moves through the passes
// outside (optional): the ranges came from a sample -- a value outside its column's range sets the flag (its key is garbage; the caller encodes again with exact ranges)
__global__ void __launch_bounds__(BLOCK) k_pk_encode(PkCols pc, int64_t n, uint64_t* keys, uint32_t* idx, int ib, uint32_t* outside) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint64_t key = 0; bool out = false;
  for (int c = 0; c < MAX_KEYS; c++) {
    if (c >= pc.n) break;
    const PkCol& k = pc.c[c]; uint64_t off;
    if (!valid_at(k.valid, i)) off = k.nulls_first ? 0 : k.span + 1;        // slot 0 / span + 1 are the NULL slots, values occupy 1 .. span (span = max - min + 1)
    else { uint64_t h, l; pk_order_bits(k.v, k.type, i, &h, &l); uint64_t d = l - k.lo_bits;      // the range fits 63 bits: inside it the high words cancel
      if (outside) { const uint64_t dh = h - k.hi_bits - (l < k.lo_bits ? 1ull : 0ull); out |= dh != 0 || d >= k.span; }
      off = 1 + (k.desc ? k.span - 1 - d : d); }
    key |= off << k.shift;
  }
  if (outside && out) atomicOr(outside, 1u);
  if (ib) keys[i] = (key << ib) | (uint64_t)i; else { keys[i] = key; if (idx) idx[i] = (uint32_t)i; }
}
// word mode, after the last pass: row numbers out of the low bits; every key column that asked for it (sorted_dst) is rebuilt from its bits of the sorted word --
// offset -> order pattern (minimum + distance, 128-bit) -> value -- with sequential reads and writes instead of a gather through the permutation
__device__ inline void pk_finish_row(const PkCols& pc, uint64_t w, int64_t i, int ib, uint32_t* __restrict__ idx) {
  idx[i] = (uint32_t)(w & ((1ull << ib) - 1ull)); const uint64_t key = w >> ib;
  for (int c = 0; c < MAX_KEYS; c++) {
    if (c >= pc.n) break;
    const PkCol& k = pc.c[c]; if (!k.sorted_dst) continue;
    const uint64_t off = (key >> k.shift) & (k.bits >= 64 ? ~0ull : (1ull << k.bits) - 1ull);
    uint64_t d = off - 1; if (k.desc) d = k.span - 1 - d;
    const uint64_t lo = k.lo_bits + d, hi = k.hi_bits + (lo < k.lo_bits ? 1ull : 0ull);
    switch (k.type) {
      case DFGPU_INT8: ((int8_t*)k.sorted_dst)[i] = (int8_t)(int64_t)(lo ^ 0x8000000000000000ull); break;
      case DFGPU_INT16: ((int16_t*)k.sorted_dst)[i] = (int16_t)(int64_t)(lo ^ 0x8000000000000000ull); break;
      case DFGPU_INT32: case DFGPU_DATE32: ((int32_t*)k.sorted_dst)[i] = (int32_t)(int64_t)(lo ^ 0x8000000000000000ull); break;
      case DFGPU_INT64: ((uint64_t*)k.sorted_dst)[i] = lo ^ 0x8000000000000000ull; break;
      case DFGPU_UINT8: ((uint8_t*)k.sorted_dst)[i] = (uint8_t)lo; break; case DFGPU_UINT16: ((uint16_t*)k.sorted_dst)[i] = (uint16_t)lo; break;
      case DFGPU_UINT32: ((uint32_t*)k.sorted_dst)[i] = (uint32_t)lo; break; case DFGPU_UINT64: ((uint64_t*)k.sorted_dst)[i] = lo; break;
      case DFGPU_FLOAT32: { uint32_t b = (uint32_t)lo; b ^= (b >> 31) ? 0x80000000u : 0xFFFFFFFFu; ((uint32_t*)k.sorted_dst)[i] = b; break; }
      case DFGPU_FLOAT64: { uint64_t b = lo; b ^= (b >> 63) ? 0x8000000000000000ull : ~0ull; ((uint64_t*)k.sorted_dst)[i] = b; break; }
      default: { uint64_t* o = (uint64_t*)k.sorted_dst + 2 * i; o[0] = lo; o[1] = hi ^ 0x8000000000000000ull; break; }      // DECIMAL128
    }
  }
}
__global__ void __launch_bounds__(BLOCK) k_pk_finish(PkCols pc, const uint64_t* __restrict__ words, int64_t m, int ib, uint32_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i >= m) return;
  pk_finish_row(pc, words[i], i, ib, idx);
}
// ---- one-sweep LSD passes over the words of word mode
// The three launches of a pass (per-tile histogram, scan of the [digit][tile] matrix, scatter) read the words twice.  The DEVICE-WIDE digit histogram of every pass does not
// depend on the order of the words, so all of them are counted while the words are encoded (k_pk_encode_hist: one LDS table per pass, merged with one global atomic per
// used counter and workgroup), and a pass is ONE launch: a workgroup takes the next tile (a ticket: tiles start in order, so every tile a workgroup waits for is resident),
// ranks its rows as the LDS-free scatter of radix_partition.h does, publishes the tile's digit counts, and finds the rows of its digit in the tiles before it by walking
// back over their published words -- (count | AGG) until one carries (inclusive prefix | PREFIX) -- the chained scan with decoupled look-back of Merrill & Garland; thread d
// walks for digit d, OS_LOOK words in flight per step (a word read at agent scope comes from beyond the XCD's L2: ~1-2 us each).  A wait that makes no progress for
// ~2^21 polls raises DFGPU_FLAG_STALLED and ends (the host reports an error) instead of keeping the device busy: a variant that published and walked the rows 16 bytes
// per lane through one wave (inline global_load/store_dwordx4 sc0 sc1) ended exactly there on its first test and was removed (round 4, call q).
constexpr int OS_NT = 512, OS_NW = OS_NT / WAVE, OS_LOOK = 8;
constexpr uint32_t OS_AGG = 1u << 30, OS_PREFIX = 2u << 30, OS_VAL = (1u << 30) - 1u;
struct OsLayout { int32_t npass; int32_t shift[8]; uint32_t mask[8]; };
__global__ void __launch_bounds__(BLOCK) k_pk_encode_hist(PkCols pc, int64_t n, uint64_t* keys, int ib, uint32_t* outside, OsLayout L, uint32_t* hists /*[npass][256]*/) {
  __shared__ uint32_t h[8 * 256];
  for (int x = threadIdx.x; x < 8 * 256; x += BLOCK) h[x] = 0;
  __syncthreads();
  const int lane = lane_id(); bool out = false;
  constexpr int U = 2;          // rows per lane and round: both rows' column loads are in flight before the first histogram add
  const int64_t nwaves = (int64_t)gridDim.x * (BLOCK / WAVE);
  for (int64_t c0 = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6; c0 * (WAVE * U) < n; c0 += nwaves) {
    uint64_t word[U]; bool on[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = c0 * (WAVE * U) + u * WAVE + lane; on[u] = i < n;
      uint64_t key = 0;
      if (on[u]) for (int c = 0; c < MAX_KEYS; c++) {
        if (c >= pc.n) break;
        const PkCol& k = pc.c[c]; uint64_t off;
        if (!valid_at(k.valid, i)) off = k.nulls_first ? 0 : k.span + 1;
        else { uint64_t hh, l; pk_order_bits(k.v, k.type, i, &hh, &l); uint64_t d = l - k.lo_bits;
          if (outside) { const uint64_t dh = hh - k.hi_bits - (l < k.lo_bits ? 1ull : 0ull); out |= dh != 0 || d >= k.span; }
          off = 1 + (k.desc ? k.span - 1 - d : d); }
        key |= off << k.shift;
      }
      word[u] = (key << ib) | (uint64_t)i;
      if (on[u]) keys[i] = word[u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint64_t onm = ballot64(on[u]); if (!onm) continue;
      const int src = __ffsll((unsigned long long)onm) - 1;
      for (int p = 0; p < L.npass; p++) {
        const uint32_t d = (uint32_t)(word[u] >> L.shift[p]) & L.mask[p], d0 = __shfl(d, src, 64);
        if (ballot64(on[u] && d != d0) == 0) { if (lane == src) atomicAdd(&h[p * 256 + d0], (uint32_t)__popcll(onm)); }       // a constant digit (the top bits of a narrow range) is one add per wave
        else if (on[u]) atomicAdd(&h[p * 256 + d], 1u);
      }
    }
  }
  if (outside && out) atomicOr(outside, 1u);
  __syncthreads();
  for (int x = threadIdx.x; x < L.npass * 256; x += BLOCK) if (h[x]) atomicAdd(&hists[x], h[x]);
}
// hists[p][256] -> gbase[p][256] = first output slot of digit d in pass p (one workgroup per pass)
__global__ void __launch_bounds__(256) k_os_bases(const uint32_t* hists, uint32_t* gbase) {
  __shared__ uint32_t wsum[4];
  const uint32_t v = hists[blockIdx.x * 256 + threadIdx.x], inc = wave_inclusive_sum(v);
  if (lane_id() == 63) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint32_t run = inc - v; for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += wsum[w];
  gbase[blockIdx.x * 256 + threadIdx.x] = run;
}
struct OsPayload { int32_t n; const void* src[4]; void* dst[4]; int32_t width[4]; };          // columns that ride behind the sort (sort_batch's take()): gathered by the last pass
struct OsFinish { PkCols pc; int32_t ib; uint32_t* idx; int64_t m; OsPayload pay; };       // the last pass of a sort: instead of the word, its slot receives the row number and the key columns rebuilt from the word (k_pk_finish's work without writing and reading the words once more)
template <int R, bool FINISH>          // R rows per lane: a tile is R x 512 rows
__global__ void __launch_bounds__(OS_NT) k_os_pass(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int64_t n, int shift, uint32_t mask, const uint32_t* __restrict__ gbase,
                                                   uint32_t* status /*[ntiles][256]*/, uint32_t* ticket, OsFinish fin, uint32_t* flags) {
  __shared__ uint16_t wcnt[R * OS_NW * 256];          // rows of digit d in (slab q, wave w), then their exclusive prefix in (slab, wave) order
  __shared__ uint32_t tbase[256]; __shared__ uint32_t tile_sh;
  if (threadIdx.x == 0) tile_sh = atomicAdd(ticket, 1u);
  for (int x = threadIdx.x; x < R * OS_NW * 128; x += OS_NT) ((uint32_t*)wcnt)[x] = 0;
  __syncthreads();
  const int64_t t = tile_sh, base = t * (int64_t)(R * OS_NT); const int wave = threadIdx.x >> 6;
  uint64_t w[R]; uint32_t d[R], rk[R]; bool on[R];
#pragma unroll
  for (int q = 0; q < R; q++) { const int64_t i = base + (int64_t)q * OS_NT + threadIdx.x; on[q] = i < n; w[q] = on[q] ? in[i] : 0ull; }
#pragma unroll
  for (int q = 0; q < R; q++) {
    d[q] = (uint32_t)(w[q] >> shift) & mask;
    uint64_t peers = ballot64(on[q]);
    for (uint32_t b = 1; b <= mask; b <<= 1) { const uint64_t mb = ballot64((d[q] & b) != 0); peers &= (d[q] & b) ? mb : ~mb; }
    rk[q] = (uint32_t)__popcll(peers & lanemask_lt());
    if (on[q] && rk[q] == 0) wcnt[((size_t)q * OS_NW + wave) * 256 + d[q]] = (uint16_t)__popcll(peers);
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    uint32_t run = 0;
#pragma unroll 8
    for (int x = 0; x < R * OS_NW; x++) { const uint16_t c = wcnt[(size_t)x * 256 + threadIdx.x]; wcnt[(size_t)x * 256 + threadIdx.x] = (uint16_t)run; run += c; }
    uint32_t* const mine = status + t * 256 + threadIdx.x;
    uint32_t excl = 0;
    if (t == 0) __hip_atomic_store(mine, run | OS_PREFIX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else {
      __hip_atomic_store(mine, run | OS_AGG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int64_t tt = t - 1; bool done = false; uint32_t idle = 0;
      while (!done) {
        uint32_t sv[OS_LOOK];
#pragma unroll
        for (int k = 0; k < OS_LOOK; k++) sv[k] = tt - k >= 0 ? __hip_atomic_load(status + (tt - k) * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : OS_PREFIX;
        int used = 0;
#pragma unroll
        for (int k = 0; k < OS_LOOK; k++) if (!done && used == k) {
          if ((sv[k] >> 30) != 0) { excl += sv[k] & OS_VAL; used = k + 1; done = (sv[k] >> 30) == 2u; }          // a word not yet published: poll again from that tile on
        }
        tt -= used;
        // Every tile in front holds an earlier ticket, so its workgroup is running and publishes without waiting for anyone: the wait ends.  Should that ever fail
        // (2^21 polls without progress: seconds), the pass gives up with a flag the host turns into an error instead of keeping the device busy for good
        if (used == 0) { __builtin_amdgcn_s_sleep(2); if (++idle > (1u << 21)) { atomicOr(flags, DFGPU_FLAG_STALLED); done = true; } } else idle = 0;
      }
      __hip_atomic_store(mine, (excl + run) | OS_PREFIX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    tbase[threadIdx.x] = gbase[threadIdx.x] + excl;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < R; q++) if (on[q]) {
    const int64_t pos = (int64_t)(tbase[d[q]] + (uint32_t)wcnt[((size_t)q * OS_NW + wave) * 256 + d[q]] + rk[q]);
    if (FINISH) { if (pos < fin.m) { pk_finish_row(fin.pc, w[q], pos, fin.ib, fin.idx);
        // the row's other columns: one random sector each, R rows of them in flight per lane while the pass streams -- the gather a separate take() would make afterwards
        const uint64_t row = w[q] & ((1ull << fin.ib) - 1ull);
        for (int c = 0; c < 4; c++) { if (c >= fin.pay.n) break;
          if (fin.pay.width[c] == 8) ((uint64_t*)fin.pay.dst[c])[pos] = ((const uint64_t*)fin.pay.src[c])[row];
          else if (fin.pay.width[c] == 4) ((uint32_t*)fin.pay.dst[c])[pos] = ((const uint32_t*)fin.pay.src[c])[row];
          else ((ulonglong2*)fin.pay.dst[c])[pos] = ((const ulonglong2*)fin.pay.src[c])[row]; } } }
    else out[pos] = w[q];
  }
}
// ---- TopK over the words of word mode (SortExec with fetch over a large input whose keys pack; topk/mod.rs keeps a heap of k rows, here:) radix SELECT on the word --
// every word is distinct (its low bits are the row number), so the k smallest words are exactly the rows a stable sort puts first.  A pass counts one 8-bit digit over the
// words that share the digits chosen so far; the host picks the digit holding the k-th word; when few candidates are left, the words up to the chosen prefix are
// compacted (k + at most 4096 of them) and sorted on their own: by counting, for every word, the words below it (one launch) when they fit, by the LSD passes otherwise.
__global__ void __launch_bounds__(BLOCK) k_ws_hist(const uint64_t* __restrict__ words, int64_t n, int shift, uint32_t mask, int has_prefix, int pshift, uint64_t prefix, uint32_t* hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0; __syncthreads();
  const int lane = lane_id();
  for (int64_t b0 = ((int64_t)blockIdx.x * BLOCK + (threadIdx.x & ~63)) * 4; b0 < n; b0 += (int64_t)gridDim.x * BLOCK * 4) {
    uint64_t w[4]; bool on[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const int64_t i = b0 + q * 64 + lane; on[q] = i < n; w[q] = on[q] ? words[i] : 0ull; }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const bool c = on[q] && (!has_prefix || (w[q] >> pshift) == prefix);
      const uint64_t cm = ballot64(c); if (!cm) continue;
      const int src = __ffsll((unsigned long long)cm) - 1; const uint32_t d = (uint32_t)(w[q] >> shift) & mask, d0 = __shfl(d, src, 64);
      if (ballot64(c && d != d0) == 0) { if (lane == src) atomicAdd(&h[d0], (uint32_t)__popcll(cm)); }
      else if (c) atomicAdd(&h[d], 1u);
    }
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void __launch_bounds__(BLOCK) k_ws_collect(const uint64_t* __restrict__ words, int64_t n, int pshift, uint64_t prefix, uint64_t* __restrict__ out, unsigned long long* counter, uint64_t cap) {
  const int lane = lane_id();
  for (int64_t b0 = (int64_t)blockIdx.x * BLOCK + (threadIdx.x & ~63); b0 < n; b0 += (int64_t)gridDim.x * BLOCK) {
    const int64_t i = b0 + lane; const uint64_t w = i < n ? words[i] : ~0ull;
    const bool sel = i < n && (w >> pshift) <= prefix;
    const uint64_t sm = ballot64(sel); if (!sm) continue;
    const int src = __ffsll((unsigned long long)sm) - 1;
    unsigned long long pos = 0; if (lane == src) pos = atomicAdd(counter, (unsigned long long)__popcll(sm));
    pos = __shfl(pos, src, 64) + (unsigned long long)__popcll(sm & lanemask_lt());
    if (sel && pos < cap) out[pos] = w;
  }
}
// m distinct words (m <= 16384): out[#words below w] = w.  Every thread keeps one word and walks all of them through LDS (broadcast reads): m^2 compares, one launch
__global__ void __launch_bounds__(BLOCK) k_ws_rank_sort(const uint64_t* __restrict__ in, int m, uint64_t* __restrict__ out) {
  __shared__ uint64_t tile[1024];
  const int i = blockIdx.x * BLOCK + threadIdx.x; const uint64_t w = i < m ? in[i] : ~0ull;
  uint32_t below = 0;
  for (int t0 = 0; t0 < m; t0 += 1024) {
    __syncthreads();
    for (int x = threadIdx.x; x < 1024; x += BLOCK) tile[x] = t0 + x < m ? in[t0 + x] : ~0ull;
    __syncthreads();
#pragma unroll 8
    for (int x = 0; x < 1024; x++) below += tile[x] < w ? 1u : 0u;
  }
  if (i < m) out[below] = w;
}
// the same select when key and row number do not share a word (keys may repeat): the rows up to the chosen prefix are marked in a bitmap, listed in row order, their
// keys gathered; (key, position in the list) then orders them as the stable sort does
__global__ void __launch_bounds__(BLOCK) k_ws_mark(const uint64_t* __restrict__ keys, int64_t n, int pshift, uint64_t prefix, uint64_t* __restrict__ bits) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const uint64_t m = ballot64(i < n && (keys[i] >> pshift) <= prefix);
  if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) bits[i >> 6] = m;
}
__global__ void __launch_bounds__(BLOCK) k_ws_gather(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ rows, int64_t m, uint64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i < m) out[i] = keys[rows[i]];
}
__global__ void __launch_bounds__(BLOCK) k_ws_rank_sort_pairs(const uint64_t* __restrict__ in, const uint32_t* __restrict__ vals, int m, uint64_t* __restrict__ out, uint32_t* __restrict__ out_vals) {
  __shared__ uint64_t tile[1024];
  const int i = blockIdx.x * BLOCK + threadIdx.x; const uint64_t w = i < m ? in[i] : ~0ull;
  uint32_t below = 0;
  for (int t0 = 0; t0 < m; t0 += 1024) {
    __syncthreads();
    for (int x = threadIdx.x; x < 1024; x += BLOCK) tile[x] = t0 + x < m ? in[t0 + x] : ~0ull;
    __syncthreads();
#pragma unroll 8
    for (int x = 0; x < 1024; x++) below += (tile[x] < w || (tile[x] == w && t0 + x < i)) ? 1u : 0u;
  }
  if (i < m) { out[below] = w; out_vals[below] = vals[i]; }
}
}  // namespace dfgpu

using namespace dfgpu;
static void sort_impl(dfgpu_ctx* ctx, const dfgpu_array* const* cols, const uint8_t* descending, const uint8_t* nulls_first, int32_t k, int64_t fetch, dfgpu_array** out, dfgpu_array** out_sorted,
                      const dfgpu_array* const* payload = nullptr, int32_t n_payload = 0, dfgpu_array** out_payload = nullptr) {
  {
    if (!cols || k < 1 || !out) fail(DFGPU_INVALID_ARGUMENT, "Sort requires at least one column");
    if (k > MAX_KEYS) fail(DFGPU_NOT_IMPLEMENTED, "more than %d sort columns", MAX_KEYS);
    int64_t n = cols[0]->length;
    SortCols sc{}; sc.n = k; int W = 0;
    for (int c = 0; c < k; c++) {
      if (cols[c]->length != n) fail(DFGPU_INVALID_ARGUMENT, "sort columns differ in length");
      int32_t lt = logical_type(cols[c]);
      SortCol& s = sc.c[c]; s.v = make_view(cols[c]); s.byte_off = W; s.max_len = 0;
      if (lt == DFGPU_UTF8) {
        HIP_CHECK(hipMemsetAsync(ctx->d_scratch64 + 12, 0, 8, ctx->stream));
        if (n) hipLaunchKernelGGL(k_max_utf8_len, dim3(grid_for(n, BLOCK * 8, 512)), dim3(BLOCK), 0, ctx->stream, s.v, n, (unsigned int*)(ctx->d_scratch64 + 12));
        KERNEL_CHECK();
        uint64_t ml = read_scratch(ctx, 12) & 0xFFFFFFFFull;
        if (ml > 1024) fail(DFGPU_NOT_IMPLEMENTED, "Utf8 sort keys longer than 1024 bytes (%llu) are not supported on device", (unsigned long long)ml);
        s.max_len = (int32_t)ml;
      }
      s.has_null_byte = (s.v.validity || s.v.key_validity) ? 1 : 0; s.descending = descending && descending[c]; s.nulls_first = nulls_first ? nulls_first[c] : 1;
      W += s.has_null_byte + (lt == DFGPU_BOOL ? 1 : (lt == DFGPU_UTF8 ? s.max_len + 4 : type_width(lt)));
    }
    // ---- packed-key path: large input, fixed-width keys whose value ranges concatenate into 64 bits; with a fetch of at most n / 16 rows only from sort_topk_words_min_rows
    // rows on and only when key and row number share one word (the select over the words; below, the byte-plane select that follows is the cheaper one)
    const bool topk = fetch > 0 && fetch * 16 <= n;
    if (n >= ctx->sort_packed_min_rows && ctx->sort_packed_keys && (!topk || n >= ctx->sort_topk_words_min_rows)) {
      bool ok = true; PkCols pc{}; pc.n = k;
      for (int c = 0; c < k && ok; c++) {
        const dfgpu_array* a = cols[c];
        ok = a->type != DFGPU_DICTIONARY && a->type != DFGPU_UTF8 && a->type != DFGPU_BOOL && type_width(a->type) > 0;
        pc.c[c].v = ok ? a->values->ptr : nullptr; pc.c[c].valid = a->validity ? (const uint64_t*)a->validity->ptr : nullptr; pc.c[c].type = a->type;
        pc.c[c].desc = descending && descending[c]; pc.c[c].nulls_first = nulls_first ? nulls_first[c] : 1;
      }
      // The value ranges cost a pass over the key columns (0.64 ms for a Decimal128 + Date32 pair over 100 M rows).  A large input takes them from a sample first (every
      // n / 2^19-th row), widened by 1/32 of the span on either side (the extremes of a sample lie inside the extremes of the rows); the encode pass checks every value
      // against them and only a miss costs the exact pass.  The packed keys, and with them the indices, do not depend on which ranges were used as long as they hold every value.
      const bool estimate = ok && ctx->sort_estimate_ranges && n >= ((int64_t)1 << 22) && !topk;      // a TopK asks for the extremes, which a sample misses more often than not (SUM per group over 20 M groups: the sampled maximum plus 1/32 of the span was too low, the sample pass and one encode pass wasted)
      for (int attempt = estimate ? 0 : 1; ok && attempt < 2; attempt++) {
        const bool sampled = attempt == 0;
        const int nblk = (int)std::min<int64_t>(ctx->num_cus * 4, std::max<int64_t>(1, n / (sampled ? std::max<int64_t>(2, n >> 19) : 1) / BLOCK + 1)), nb = 1;
        BufferPtr mm = alloc_buffer(ctx, (size_t)(nblk + 1) * MAX_KEYS * 32);
        unsigned long long* folded = (unsigned long long*)mm->ptr + (size_t)nblk * MAX_KEYS * 4;
        { KernelTimer kt_(ctx, sampled ? "sort_key_sample" : "sort_key_ranges");
          hipLaunchKernelGGL(k_pk_minmax, dim3(nblk), dim3(BLOCK), 0, ctx->stream, pc, n, sampled ? std::max<int64_t>(2, n >> 19) : (int64_t)1, (unsigned long long*)mm->ptr);
          hipLaunchKernelGGL(k_pk_minmax_fold, dim3(1), dim3(BLOCK), 0, ctx->stream, (const unsigned long long*)mm->ptr, nblk, k, folded); KERNEL_CHECK(); }
        std::vector<uint64_t> h((size_t)MAX_KEYS * 4);
        ctx->count_sync("sync:sort_key_ranges"); fetch_to_host(ctx, h.data(), folded, (size_t)k * 32);
        int total_bits = 0; int bits_of[MAX_KEYS];
        for (int c = 0; c < k && ok; c++) {
          uint64_t mnh = ~0ull, mnl = ~0ull, mxh = 0, mxl = 0;
          for (int b = 0; b < nb; b++) { const uint64_t* o = &h[((size_t)b * MAX_KEYS + c) * 4];
            if (o[0] < mnh || (o[0] == mnh && o[1] < mnl)) { mnh = o[0]; mnl = o[1]; } if (mxh < o[2] || (mxh == o[2] && mxl < o[3])) { mxh = o[2]; mxl = o[3]; } }
          uint64_t span = 1;            // a column of NULLs only: one (unused) value slot
          if (!(mnh == ~0ull && mnl == ~0ull && mxh == 0 && mxl == 0)) {
            unsigned __int128 mn = ((unsigned __int128)mnh << 64) | mnl, mx = ((unsigned __int128)mxh << 64) | mxl, d = mx - mn;
            if (sampled) { const unsigned __int128 margin = (d >> 5) + 1; mn = mn > margin ? mn - margin : 0; mx = mx + margin < mx ? ~(unsigned __int128)0 : mx + margin; d = mx - mn; mnl = (uint64_t)mn; mnh = (uint64_t)(mn >> 64); }
            if (d >= ((unsigned __int128)1 << 62)) { ok = false; break; }
            span = (uint64_t)d + 1; pc.c[c].lo_bits = mnl; pc.c[c].hi_bits = mnh;
          }
          pc.c[c].span = span;
          int b = 1; while (((span + 2) >> b) != 0 && b < 64) b++;       // offsets 0 .. span + 1
          bits_of[c] = b; total_bits += b;
        }
        if (sampled && !(ok && total_bits <= 64)) { ok = true; continue; }          // the widened sample does not pack: the exact ranges may
        if (ok && total_bits <= 64) {
          int sh = total_bits; for (int c = 0; c < k; c++) { sh -= bits_of[c]; pc.c[c].shift = sh; }
          int ib = 1; while (((uint64_t)(n - 1) >> ib) != 0) ib++;                     // bits of the largest row number
          const bool word = total_bits + ib <= 64;
          for (int c = 0; c < k; c++) { pc.c[c].bits = bits_of[c]; pc.c[c].sorted_dst = nullptr; }
          BufferPtr k0 = alloc_buffer(ctx, (size_t)n * 8), k1 = alloc_buffer(ctx, (size_t)n * 8), v1 = word ? BufferPtr() : alloc_buffer(ctx, (size_t)n * 4);
          ArrayHolder idx(new_fixed(ctx, DFGPU_UINT32, (word || topk) && fetch >= 0 && fetch < n ? fetch : n));
          const int npass = (total_bits + 7) / 8, dbits = (total_bits + npass - 1) / npass;          // up to 8-bit digits, evened out over the passes (9-bit digits double the count table of the stable scatter: 4 passes of 9 measured 6.6 ms against 4.5 ms for 5 of 8 on 100 M rows)
          // one-sweep passes: word mode, inputs large enough that a pass is bandwidth and not launches, row counts a 30-bit status value holds
          const int os_r = ctx->sort_onesweep_rows == 16 && n >= ((int64_t)1 << 21) ? 16 : 8;          // 8192-row tiles from 2 M rows on (256 tiles: one per CU); below, the 4096-row tile fills the chip better
          const bool onesweep = word && !topk && ctx->sort_onesweep_rows > 0 && npass <= 8 && n >= ctx->sort_onesweep_min_rows && n < ((int64_t)1 << 30);
          const int64_t os_tiles = (n + (int64_t)os_r * OS_NT - 1) / ((int64_t)os_r * OS_NT);
          BufferPtr os_buf; OsLayout L{}; uint32_t *os_hist = nullptr, *os_base = nullptr, *os_ticket = nullptr, *os_status = nullptr;
          if (onesweep) {
            L.npass = npass; for (int p = 0; p < npass; p++) { const int sh = p * dbits, bits = total_bits - sh < dbits ? total_bits - sh : dbits; L.shift[p] = sh + ib; L.mask[p] = (1u << bits) - 1u; }
            const size_t head = (size_t)(8 * 256 * 2 + 64) * 4;
            os_buf = alloc_buffer(ctx, head + (size_t)npass * (size_t)os_tiles * 256 * 4);
            os_hist = (uint32_t*)os_buf->ptr; os_base = os_hist + 8 * 256; os_ticket = os_base + 8 * 256; os_status = os_ticket + 64;
            HIP_CHECK(hipMemsetAsync(os_buf->ptr, 0, head + (size_t)npass * (size_t)os_tiles * 256 * 4, ctx->stream));
          }
          { KernelTimer kt_(ctx, "sort_key_encode");
            if (sampled) HIP_CHECK(hipMemsetAsync(ctx->d_scratch64 + 14, 0, 8, ctx->stream));
            if (onesweep) hipLaunchKernelGGL(k_pk_encode_hist, dim3(grid_for(n, BLOCK, ctx->num_cus * 8)), dim3(BLOCK), 0, ctx->stream, pc, n, (uint64_t*)k0->ptr, ib, sampled ? (uint32_t*)(ctx->d_scratch64 + 14) : (uint32_t*)nullptr, L, os_hist);
            else hipLaunchKernelGGL(k_pk_encode, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, pc, n, (uint64_t*)k0->ptr, word || topk ? (uint32_t*)nullptr : (uint32_t*)idx.get()->values->ptr, word ? ib : 0,
                               sampled ? (uint32_t*)(ctx->d_scratch64 + 14) : (uint32_t*)nullptr);
            KERNEL_CHECK(); }
          if (sampled) { const uint64_t miss = read_scratch(ctx, 14); ctx->count_sync("sync:sort_key_outside"); if (miss) continue; }       // some value lies outside the sampled ranges: exact pass
          uint64_t* ka = (uint64_t*)k0->ptr; uint64_t* kb = (uint64_t*)k1->ptr; uint32_t* va = (uint32_t*)idx.get()->values->ptr; uint32_t* vb = word ? nullptr : (uint32_t*)v1->ptr;
          bool fuse_finish = false; std::vector<ArrayHolder> sk_fused((size_t)k), pay_out((size_t)(n_payload > 0 ? n_payload : 0));
          int64_t ns = n;                                // rows the passes sort
          bool sorted_already = false; ArrayHolder rows;
          if (topk) {
            KernelTimer kt_(ctx, "sort_topk_words");
            BufferPtr hist = alloc_buffer(ctx, 256 * 4 + 8);
            int sh = total_bits + (word ? ib : 0), pshift = 0, has_prefix = 0; uint64_t prefix = 0; int64_t remaining = fetch, ncand = n;
            const int grid = grid_for(n, BLOCK * 4, ctx->num_cus * 8);
            while (sh > 0) {
              const int bits = sh < 8 ? sh : 8; sh -= bits;
              HIP_CHECK(hipMemsetAsync(hist->ptr, 0, 256 * 4, ctx->stream));
              hipLaunchKernelGGL(k_ws_hist, dim3(grid), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)ka, n, sh, (1u << bits) - 1u, has_prefix, pshift, prefix, (uint32_t*)hist->ptr); KERNEL_CHECK();
              uint32_t hh[256]; ctx->count_sync("sync:topk_histogram"); fetch_to_host(ctx, hh, hist->ptr, sizeof hh);
              int64_t cum = 0; int d = 0;
              for (; d < (1 << bits) - 1; d++) { if (cum + (int64_t)hh[d] >= remaining) break; cum += hh[d]; }
              remaining -= cum; ncand = hh[d]; prefix = (prefix << bits) | (uint64_t)d; pshift = sh; has_prefix = 1;
              if (ncand <= remaining + 4096) break;
            }
            ns = fetch - remaining + ncand;              // rows up to the chosen prefix: the fetch first of the sorted order are among them
            if (word) {
              unsigned long long* counter = (unsigned long long*)((uint32_t*)hist->ptr + 256);
              HIP_CHECK(hipMemsetAsync(counter, 0, 8, ctx->stream));
              hipLaunchKernelGGL(k_ws_collect, dim3(grid_for(n, BLOCK, ctx->num_cus * 16)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)ka, n, pshift, prefix, kb, counter, (uint64_t)ns); KERNEL_CHECK();
              std::swap(ka, kb);
              if (ns <= 16384) { hipLaunchKernelGGL(k_ws_rank_sort, dim3(grid_for(ns, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)ka, (int)ns, kb); KERNEL_CHECK(); std::swap(ka, kb); sorted_already = true; }
            } else {
              BufferPtr bm = alloc_buffer(ctx, (size_t)((n + 63) / 64) * 8);
              hipLaunchKernelGGL(k_ws_mark, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)ka, n, pshift, prefix, (uint64_t*)bm->ptr); KERNEL_CHECK();
              rows.a = mask_to_indices_uncounted(ctx, (const uint64_t*)bm->ptr, n);          // ascending rows; the first ns entries are written (ns is known from the histograms)
              hipLaunchKernelGGL(k_ws_gather, dim3(grid_for(ns, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)ka, (const uint32_t*)rows.get()->values->ptr, ns, kb); KERNEL_CHECK();
              std::swap(ka, kb); va = (uint32_t*)rows.get()->values->ptr;
              if (ns <= 16384) { hipLaunchKernelGGL(k_ws_rank_sort_pairs, dim3(grid_for(ns, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)ka, (const uint32_t*)va, (int)ns, kb, vb); KERNEL_CHECK();
                std::swap(ka, kb); std::swap(va, vb); sorted_already = true; }
            }
          }
          if (sorted_already) {}
          else if (onesweep) {
            KernelTimer kt_(ctx, "sort_pass_onesweep");
            hipLaunchKernelGGL(k_os_bases, dim3(npass), dim3(256), 0, ctx->stream, (const uint32_t*)os_hist, os_base);
            // the last pass writes the result itself (row numbers + rebuilt key columns) when every key column can be rebuilt or none is asked for
            fuse_finish = ctx->sort_onesweep_fused_finish;
            const int64_t m = idx.get()->length;
            if (fuse_finish && out_sorted) for (int c = 0; c < k; c++) if (!pc.c[c].valid) {
              sk_fused[(size_t)c].a = new_fixed(ctx, cols[c]->type, m, cols[c]->precision, cols[c]->scale); pc.c[c].sorted_dst = sk_fused[(size_t)c].get()->values->ptr; }
            for (int p = 0; p < npass; p++) {
              const bool fin = fuse_finish && p == npass - 1;
              OsFinish of{}; if (fin) { of.pc = pc; of.ib = ib; of.idx = (uint32_t*)idx.get()->values->ptr; of.m = m;
                // payload columns: fixed-width (4 / 8 / 16 bytes), no NULLs, at most four; the others are left to the caller's take()
                for (int c = 0; c < n_payload && out_payload && ctx->sort_payload_in_last_pass && of.pay.n < 4; c++) {
                  const dfgpu_array* a = payload[c]; if (!a || a->validity || a->type == DFGPU_DICTIONARY || a->type == DFGPU_UTF8 || a->type == DFGPU_BOOL || a->length != n) continue;
                  const int wdt = type_width(a->type); if (wdt != 4 && wdt != 8 && wdt != 16) continue;
                  pay_out[(size_t)c].a = new_fixed(ctx, a->type, m, a->precision, a->scale);
                  of.pay.src[of.pay.n] = a->values->ptr; of.pay.dst[of.pay.n] = pay_out[(size_t)c].get()->values->ptr; of.pay.width[of.pay.n] = wdt; of.pay.n++;
                } }
              const uint32_t* gb = os_base + p * 256; uint32_t* stp = os_status + (size_t)p * os_tiles * 256;
#define OS_LAUNCH(R_, F_) hipLaunchKernelGGL((k_os_pass<R_, F_>), dim3((unsigned)os_tiles), dim3(OS_NT), 0, ctx->stream, (const uint64_t*)ka, kb, n, L.shift[p], L.mask[p], gb, stp, os_ticket + p, of, ctx->d_flags)
              if (os_r == 16) { if (fin) OS_LAUNCH(16, true); else OS_LAUNCH(16, false); } else { if (fin) OS_LAUNCH(8, true); else OS_LAUNCH(8, false); }
#undef OS_LAUNCH
              std::swap(ka, kb);
            }
            KERNEL_CHECK();
            { const int saved = ctx->defer_flag_checks; ctx->defer_flag_checks = 1; check_flags(ctx, "sort_onesweep"); ctx->defer_flag_checks = saved; }      // read with the next flag read-back (ctx_synchronize at the latest), no round trip of its own
          }
          // the words a select compacted arrive in no order: their passes also sort the row-number bits (ties = row order); everywhere else the input order breaks ties
          const int sort_bits = topk && word ? total_bits + ib : total_bits, base_shift = word && !topk ? ib : 0;
          const int npass_l = (sort_bits + 7) / 8, dbits_l = (sort_bits + npass_l - 1) / npass_l;
          if (sorted_already || onesweep) {}
          else for (int shift = 0; shift < sort_bits; shift += dbits_l) {
            const int bits = sort_bits - shift < dbits_l ? sort_bits - shift : dbits_l; const bool last = shift + dbits_l >= sort_bits;
            RpCols rc{};
            if (word) { rc.n = 1; rc.c[0] = RpCol{ nullptr, kb, 8, RP_HASHKEY, 0 }; }       // the whole record is the word the digit is read from
            else { rc.n = last ? 1 : 2; rc.c[0] = RpCol{ va, vb, 4, RP_RAW, 0 };
              if (!last) rc.c[1] = RpCol{ nullptr, kb, 8, RP_HASHKEY, 0 }; }                // the last pass only needs the row ids
            (void)rp_partition(ctx, RpHashDigit{ ka, shift + base_shift, (1u << bits) - 1u }, ns, 1u << bits, rc, true, ctx->d_scratch64 + 9, "sort_pass_hist", "sort_pass_scan", "sort_pass_scatter", false);
            std::swap(ka, kb); std::swap(va, vb);
          }
          if (word) {
            const int64_t m = idx.get()->length;
            if (fuse_finish) { if (out_sorted) for (int c = 0; c < k; c++) out_sorted[c] = sk_fused[(size_t)c].release();
              if (out_payload) for (int c = 0; c < n_payload; c++) out_payload[c] = pay_out[(size_t)c].release();
              *out = idx.release(); return; }
            std::vector<ArrayHolder> sk((size_t)k);
            if (out_sorted) for (int c = 0; c < k; c++) if (!pc.c[c].valid) {          // a key column without NULLs comes back in sorted order for the price of its sequential write
              sk[(size_t)c].a = new_fixed(ctx, cols[c]->type, m, cols[c]->precision, cols[c]->scale); pc.c[c].sorted_dst = sk[(size_t)c].get()->values->ptr; }
            if (m) { KernelTimer kt_(ctx, "sort_finish");
              hipLaunchKernelGGL(k_pk_finish, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, pc, (const uint64_t*)ka, m, ib, (uint32_t*)idx.get()->values->ptr); KERNEL_CHECK(); }
            if (out_sorted) for (int c = 0; c < k; c++) out_sorted[c] = sk[(size_t)c].release();
            *out = idx.release();
            return;
          }
          if (topk) { HIP_CHECK(hipMemcpyAsync(idx.get()->values->ptr, va, (size_t)fetch * 4, hipMemcpyDeviceToDevice, ctx->stream)); *out = idx.release(); return; }     // the first fetch of the ns sorted rows
          if (va != (uint32_t*)idx.get()->values->ptr) HIP_CHECK(hipMemcpyAsync(idx.get()->values->ptr, va, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
          if (fetch >= 0 && fetch < n) { dfgpu_array* s2 = nullptr; dfgpu_status st = dfgpu_array_slice(ctx, idx.get(), 0, fetch, &s2); if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str()); *out = s2; }
          else *out = idx.release();
          return;
        }
      }
    }
    ArrayHolder idx(new_fixed(ctx, DFGPU_UINT32, n));
    launch_iota_u32(ctx, (uint32_t*)idx.get()->values->ptr, n, 0);
    if (n > 1) {
      BufferPtr planes = alloc_buffer(ctx, (size_t)W * n), varies = alloc_buffer(ctx, (size_t)W * 4, true);
      hipLaunchKernelGGL(k_encode_sort_keys, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, sc, n, (uint8_t*)planes->ptr);
      hipLaunchKernelGGL(k_plane_varies, dim3(grid_for(n, BLOCK * 16, 256), W), dim3(BLOCK), 0, ctx->stream, (const uint8_t*)planes->ptr, n, (uint32_t*)varies->ptr);
      KERNEL_CHECK();
      std::vector<uint32_t> h((size_t)W);
      ctx->count_sync("sync:sort_planes"); fetch_to_host(ctx, h.data(), varies->ptr, h.size() * 4);
      if (fetch > 0 && fetch * 16 <= n && n >= (1 << 16)) {          // TopK: select, then sort the few selected rows
        std::vector<int> vp; for (int b = 0; b < W; b++) if (h[(size_t)b]) vp.push_back(b);
        int64_t nwords = (n + 63) / 64, remaining = fetch, ncand = n; bool small = false;
        BufferPtr cand = alloc_buffer(ctx, (size_t)nwords * 8), accept = alloc_buffer(ctx, (size_t)nwords * 8, true), hist = alloc_buffer(ctx, 256 * 4);
        HIP_CHECK(hipMemsetAsync(cand->ptr, 0xFF, (size_t)nwords * 8, ctx->stream));        // bits >= n are masked by i < n in the kernel
        KernelTimer kt_(ctx, "k_topk_select");
        int grid = grid_for(nwords, BLOCK / WAVE, ctx->num_cus * 16);
        int prev = -1, prev_digit = 0;
        for (size_t it = 0; it <= vp.size() && !small; it++) {
          int cur = it < vp.size() ? vp[it] : -1;
          if (cur >= 0) HIP_CHECK(hipMemsetAsync(hist->ptr, 0, 256 * 4, ctx->stream));
          hipLaunchKernelGGL(k_topk_step, dim3(grid), dim3(BLOCK), 0, ctx->stream, (const uint8_t*)planes->ptr, n, prev, prev_digit, cur, (uint64_t*)cand->ptr, (uint64_t*)accept->ptr, (uint32_t*)hist->ptr);
          KERNEL_CHECK();
          if (cur < 0) break;
          uint32_t hh[256];
          ctx->count_sync("sync:topk_histogram"); fetch_to_host(ctx, hh, hist->ptr, sizeof hh);
          int64_t cum = 0; int d = 0;
          for (; d < 255; d++) { if (cum + (int64_t)hh[d] >= remaining) break; cum += hh[d]; }
          remaining -= cum; ncand = hh[d]; prev = cur; prev_digit = d;
          if (ncand <= remaining + 4096) {                     // few enough: apply this decision and stop selecting
            hipLaunchKernelGGL(k_topk_step, dim3(grid), dim3(BLOCK), 0, ctx->stream, (const uint8_t*)planes->ptr, n, prev, prev_digit, -1, (uint64_t*)cand->ptr, (uint64_t*)accept->ptr, (uint32_t*)hist->ptr);
            KERNEL_CHECK(); small = true;
          }
        }
        if (small) {
          hipLaunchKernelGGL(k_or_words, dim3(grid_for(nwords, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)accept->ptr, (const uint64_t*)cand->ptr, (uint64_t*)accept->ptr, nwords);
          KERNEL_CHECK();
          ArrayHolder rows(mask_to_indices_impl(ctx, (const uint64_t*)accept->ptr, n));         // the top set plus boundary ties, ascending row order
          std::vector<ArrayHolder> keys((size_t)k); std::vector<const dfgpu_array*> kp;
          for (int c = 0; c < k; c++) { keys[(size_t)c].a = take_impl(ctx, cols[c], rows.get()->values->ptr, 4, nullptr, rows.get()->length); kp.push_back(keys[(size_t)c].get()); }
          dfgpu_array* local = nullptr;
          dfgpu_status st = dfgpu_sort_to_indices(ctx, kp.data(), descending, nulls_first, k, fetch, &local);      // small: the full path (stable)
          if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str());
          ArrayHolder lh(local);
          *out = take_impl(ctx, rows.get(), lh.get()->values->ptr, 4, nullptr, lh.get()->length);
          return;
        }
        // every varying plane is decided and the candidates (rows with identical keys) still outnumber what is needed: sort everything
      }
      RadixPlan p = plan_for(n);
      BufferPtr tmp = alloc_buffer(ctx, (size_t)n * 4), hist = alloc_buffer(ctx, (size_t)256 * p.nb * 4);
      uint32_t *v0 = (uint32_t*)idx.get()->values->ptr, *v1 = (uint32_t*)tmp->ptr;
      if (n >= (1 << 20)) {
        // Large inputs: a pass that looks its digit up by row id (plane[row]) is a random one-byte gather per row and pass -- 64-B sectors,
        // twice (histogram + scatter): 13 GB per pass at 100 M rows.  Instead up to four varying planes at a time are packed into a u32 that
        // MOVES with the row id (sequential 8 B per row and pass), fetched through the current order once per group of four.
        std::vector<int> vp; for (int b = 0; b < W; b++) if (h[(size_t)b]) vp.push_back(b);
        BufferPtr ka = alloc_buffer(ctx, (size_t)n * 4), kb = alloc_buffer(ctx, (size_t)n * 4), packed = alloc_buffer(ctx, (size_t)n * 4);
        uint32_t *k0 = (uint32_t*)ka->ptr, *k1 = (uint32_t*)kb->ptr;
        bool first = true;
        for (int end = (int)vp.size(); end > 0; end -= 4) {
          int beg = end - 4 < 0 ? 0 : end - 4, cnt = end - beg;
          PackPlanes pp{}; pp.n = cnt; for (int j = 0; j < cnt; j++) pp.plane[j] = (const uint8_t*)planes->ptr + (int64_t)vp[(size_t)(end - 1 - j)] * n;      // byte 0 = least significant plane
          hipLaunchKernelGGL(k_pack_planes, dim3(grid_for(n, BLOCK * 4)), dim3(BLOCK), 0, ctx->stream, pp, n, first ? k0 : (uint32_t*)packed->ptr);
          KERNEL_CHECK();
          if (!first) gather_u32(ctx, (const uint32_t*)packed->ptr, v0, n, k0);                 // the group's bytes in the order reached so far
          for (int j = 0; j < cnt; j++) {
            radix_pass(ctx, DigitKeys{ k0, 8 * j }, (const uint32_t*)k0, v0, k1, v1, n, (uint32_t*)hist->ptr, p);
            std::swap(k0, k1); std::swap(v0, v1);
          }
          first = false;
        }
      } else if (n <= ctx->sort_one_block_max_rows && n <= OB_MAX) {
        KernelTimer kt_(ctx, "radix_pass_one_block");
        hipLaunchKernelGGL(k_rs_one_block, dim3(1), dim3(OB_NT), 0, ctx->stream, (const uint8_t*)planes->ptr, (int)n, W, (const uint32_t*)varies->ptr, (const uint32_t*)v0, v1);
        KERNEL_CHECK(); std::swap(v0, v1);
      } else if (p.nb <= 512 && ctx->sort_fused_small_passes) {
        // two launches per varying plane: the histogram, then k_rs_plane_pass (scan folded into the scatter)
        KernelTimer kt_(ctx, "radix_pass");
        for (int b = W - 1; b >= 0; b--) {        // least significant plane first
          if (!h[(size_t)b]) continue;
          const uint8_t* plane = (const uint8_t*)planes->ptr + (int64_t)b * n;
          hipLaunchKernelGGL((k_rs_hist<DigitPlane>), dim3(p.nb), dim3(BLOCK), 0, ctx->stream, DigitPlane{ plane }, (const uint32_t*)nullptr, (const uint32_t*)v0, n, p.chunk, p.nb, (uint32_t*)hist->ptr);
          hipLaunchKernelGGL(k_rs_plane_pass, dim3(p.nb), dim3(BLOCK), 0, ctx->stream, plane, (const uint8_t*)nullptr, (const uint32_t*)v0, n, p.chunk, p.nb, (const uint32_t*)hist->ptr, (uint32_t*)nullptr, v1);
          std::swap(v0, v1);
        }
        KERNEL_CHECK();
      } else
      for (int b = W - 1; b >= 0; b--) {        // least significant plane first
        if (!h[(size_t)b]) continue;
        radix_pass(ctx, DigitPlane{ (const uint8_t*)planes->ptr + (int64_t)b * n }, (const uint32_t*)nullptr, v0, (uint32_t*)nullptr, v1, n, (uint32_t*)hist->ptr, p);
        std::swap(v0, v1);
      }
      if (v0 != (uint32_t*)idx.get()->values->ptr) HIP_CHECK(hipMemcpyAsync(idx.get()->values->ptr, v0, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (fetch >= 0 && fetch < n) { dfgpu_array* s = nullptr; dfgpu_status st = dfgpu_array_slice(ctx, idx.get(), 0, fetch, &s); if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str()); *out = s; }
    else *out = idx.release();
  }
}
extern "C" dfgpu_status dfgpu_sort_to_indices(dfgpu_ctx* ctx, const dfgpu_array* const* cols, const uint8_t* descending, const uint8_t* nulls_first, int32_t k, int64_t fetch, dfgpu_array** out) {
  return guard(ctx, [&] { sort_impl(ctx, cols, descending, nulls_first, k, fetch, out, nullptr); });
}
extern "C" dfgpu_status dfgpu_sort_take(dfgpu_ctx* ctx, const dfgpu_array* const* cols, const uint8_t* descending, const uint8_t* nulls_first, int32_t k, int64_t fetch, const dfgpu_array* const* payload, int32_t n_payload,
                                        dfgpu_array** out, dfgpu_array** out_sorted, dfgpu_array** out_payload) {
  return guard(ctx, [&] {
    if (!out_sorted || (n_payload > 0 && (!payload || !out_payload)) || n_payload < 0) fail(DFGPU_INVALID_ARGUMENT, "sort_take: null argument");
    for (int32_t c = 0; c < k; c++) out_sorted[c] = nullptr;
    for (int32_t c = 0; c < n_payload; c++) out_payload[c] = nullptr;
    sort_impl(ctx, cols, descending, nulls_first, k, fetch, out, out_sorted, payload, n_payload, out_payload);
  });
}
extern "C" dfgpu_status dfgpu_sort_to_indices_keys(dfgpu_ctx* ctx, const dfgpu_array* const* cols, const uint8_t* descending, const uint8_t* nulls_first, int32_t k, int64_t fetch, dfgpu_array** out,
                                                   dfgpu_array** out_sorted) {
  return guard(ctx, [&] {
    if (!out_sorted) fail(DFGPU_INVALID_ARGUMENT, "sort_to_indices_keys: null argument");
    for (int32_t c = 0; c < k; c++) out_sorted[c] = nullptr;
    sort_impl(ctx, cols, descending, nulls_first, k, fetch, out, out_sorted);
  });
}
