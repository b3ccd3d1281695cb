// Experiment (not product code): the whole radix-partitioned probe pipeline at the fan-out that puts a partition's table into LDS:
//   hist (G tiles per workgroup so that counts[p][t..t+G) is one burst) -> scan -> staged scatter (XCD-aware tile map) of (key, row)
//   -> memset found -> join (LDS table of u32 = tag:18 | local build row:14, verified against the partition's 16-B build records)
//   -> order-restoring compaction of found[] into (probe row, build row) pairs.
// Build: hipcc --offload-arch=gfx950 -O3 -o pjoin_pipeline_microbench.bin pjoin_pipeline_microbench.hip
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__host__ __device__ inline uint64_t mix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL; return x ^ (x >> 31);
}
__device__ inline uint32_t pid_of(uint64_t h, uint32_t P) { return (uint32_t)(((h >> 32) * (uint64_t)P) >> 32); }
struct BRec { uint64_t key; uint32_t row; uint32_t pad; };

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

template <int NT, int R, typename REC>
__global__ void __launch_bounds__(NT) k_staged(const uint64_t* keys, long n, uint32_t P, long ntiles, const uint32_t* goff, uint64_t* out_key, uint32_t* out_idx, BRec* out_rec) {
  extern __shared__ uint32_t lds[];
  constexpr int TILE = NT * R;
  uint32_t* cnt = lds; int32_t* delta = (int32_t*)(lds + P); uint32_t* sidx = lds + 2 * P; uint64_t* skey = (uint64_t*)(lds + 2 * P + TILE + ((2 * P + TILE) & 1));
  __shared__ uint32_t wsum[NT / 64];
  long per = (ntiles + 7) / 8; long t = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (t >= ntiles) return;
  const long base = t * (long)TILE;
  for (int p = threadIdx.x; p < (int)P; p += NT) cnt[p] = 0;
  __syncthreads();
  uint64_t k[R]; uint32_t pid[R], rk[R];
#pragma unroll
  for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; k[q] = i < n ? keys[i] : 0; }
#pragma unroll
  for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; pid[q] = pid_of(mix64(k[q]), P); rk[q] = i < n ? atomicAdd(&cnt[pid[q]], 1u) : 0; }
  __syncthreads();
  {
    const int per_t = ((int)P + NT - 1) / NT; uint32_t loc[8]; uint32_t s = 0;
    for (int j = 0; j < per_t; j++) { int p = threadIdx.x * per_t + j; loc[j] = p < (int)P ? cnt[p] : 0; s += loc[j]; }
    uint32_t inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if ((threadIdx.x & 63) >= d) inc += o; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t wbase = 0; for (int w = 0; w < (int)(threadIdx.x >> 6); w++) wbase += wsum[w];
    uint32_t run = wbase + inc - s;
    for (int j = 0; j < per_t; j++) { int p = threadIdx.x * per_t + j; if (p < (int)P) { cnt[p] = run; delta[p] = (int32_t)goff[(long)p * ntiles + t] - (int32_t)run; run += loc[j]; } }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; if (i < n) { uint32_t s = cnt[pid[q]] + rk[q]; skey[s] = k[q]; sidx[s] = (uint32_t)i; } }
  __syncthreads();
  long left = n - base; int m = left < TILE ? (int)left : TILE;
  for (int i = threadIdx.x; i < m; i += NT) {
    uint64_t kk = skey[i]; uint32_t p = pid_of(mix64(kk), P); long pos = (long)delta[p] + i;
    if (out_rec) { BRec r; r.key = kk; r.row = sidx[i]; r.pad = 0; out_rec[pos] = r; } else { out_key[pos] = kk; out_idx[pos] = sidx[i]; }
  }
}

constexpr uint32_t EMPTY = 0xFFFFFFFFu;
template <int NT, int U>
__global__ void __launch_bounds__(NT) k_join(const BRec* brec, const uint32_t* bstart, const uint64_t* pkey, const uint32_t* pidx, const uint32_t* pstart, int sbits,
                                            uint32_t* found, unsigned long long* total /*[0] matches, [1] duplicate build keys*/) {
  extern __shared__ uint32_t tab[];
  const uint32_t S = 1u << sbits, M = S - 1;
  const int p = blockIdx.x;
  for (uint32_t s = threadIdx.x; s < S; s += NT) tab[s] = EMPTY;
  __syncthreads();
  const uint32_t b0 = bstart[p], b1 = bstart[p + 1];
  const BRec* br = brec + b0;
  for (uint32_t j = threadIdx.x; j < b1 - b0; j += NT) {
    uint64_t k = br[j].key, h = mix64(k); uint32_t s = (uint32_t)h & M, tag = (uint32_t)(h >> sbits) & 0x3FFFFu, ent = (tag << 14) | j;
    for (;;) {
      uint32_t old = atomicCAS(&tab[s], EMPTY, ent);
      if (old == EMPTY) break;
      if ((old >> 14) == tag && br[old & 0x3FFFu].key == k) { total[1] = 1; break; }
      s = (s + 1) & M;
    }
  }
  __syncthreads();
  const uint32_t q0 = pstart[p], q1 = pstart[p + 1]; unsigned cntm = 0;
  for (uint32_t i0 = q0 + threadIdx.x; i0 < q1; i0 += NT * U) {
    uint64_t k[U]; uint32_t s[U], tag[U], c[U]; bool on[U];
#pragma unroll
    for (int u = 0; u < U; u++) { uint32_t i = i0 + u * NT; on[u] = i < q1; k[u] = pkey[on[u] ? i : q1 - 1]; }
#pragma unroll
    for (int u = 0; u < U; u++) { uint64_t h = mix64(k[u]); s[u] = (uint32_t)h & M; tag[u] = (uint32_t)(h >> sbits) & 0x3FFFFu; c[u] = tab[s[u]]; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (!on[u]) continue;
      uint32_t cc = c[u], ss = s[u];
      while (cc != EMPTY) {
        if ((cc >> 14) == tag[u]) { BRec r = br[cc & 0x3FFFu]; if (r.key == k[u]) { found[pidx[i0 + u * NT]] = r.row; cntm++; break; } }
        ss = (ss + 1) & M; cc = tab[ss];
      }
    }
  }
  for (int d = 32; d > 0; d >>= 1) cntm += __shfl_xor(cntm, d, 64);
  if ((threadIdx.x & 63) == 0 && cntm) atomicAdd(&total[0], (unsigned long long)cntm);
}

// compaction of found[]: 4096 rows per workgroup, 16 consecutive rows per lane
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
__global__ void __launch_bounds__(256) k_found_write(const uint32_t* found, long n, const uint32_t* offs, uint32_t* out_probe, uint64_t* out_build) {
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
  for (int q = 0; q < 16; q++) if (v[q] != EMPTY) { out_probe[ex] = (uint32_t)(base + q); out_build[ex] = v[q]; ex++; }
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

template <int NT, int R, int G>
static void partition(const uint64_t* keys, long n, uint32_t P, uint32_t* counts, uint32_t* goff, void* tmp, size_t tmp_bytes, uint64_t* ok, uint32_t* oi, BRec* orec, uint32_t* start, const char* what) {
  const int TILE = NT * R; long ntiles = (n + TILE - 1) / TILE; long nh = (ntiles + G - 1) / G;
  Ev e0, e1, e2, e3;
  for (int rep = 0; rep < 2; rep++) {
    e0.rec();
    hipLaunchKernelGGL((k_hist<NT, R, G>), dim3(nh), dim3(NT), P * G * 4, 0, keys, n, P, ntiles, counts);
    e1.rec();
    size_t tb = tmp_bytes; CK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, counts, goff, (int)((long)P * ntiles)));
    e2.rec();
    long grid = ((ntiles + 7) / 8) * 8;
    size_t lds = (2 * P + 3 * TILE + 2) * 4;
    hipLaunchKernelGGL((k_staged<NT, R, BRec>), dim3(grid), dim3(NT), lds, 0, keys, n, P, ntiles, goff, ok, oi, orec);
    e3.rec(); CK(hipEventSynchronize(e3.ev)); CK(hipGetLastError());
  }
  hipLaunchKernelGGL(k_starts, dim3((P + 256) / 256), dim3(256), 0, 0, goff, ntiles, (int)P, n, start);
  CK(hipDeviceSynchronize());
  printf("%-6s P=%u tile=%d G=%d rows=%ld: hist %.3f  scan %.3f  scatter %.3f  = %.3f ms\n", what, P, TILE, G, n, ms(e0, e1), ms(e1, e2), ms(e2, e3), ms(e0, e3));
}

int main(int argc, char** argv) {
  long nb = argc > 1 ? atol(argv[1]) : 15000000, np = argc > 2 ? atol(argv[2]) : 150000000;
  uint64_t *bk, *pk; CK(hipMalloc(&bk, nb * 8)); CK(hipMalloc(&pk, np * 8));
  hipLaunchKernelGGL(k_fill, dim3((nb + 255) / 256), dim3(256), 0, 0, bk, nb, 1ull);
  hipLaunchKernelGGL(k_fill, dim3((np + 255) / 256), dim3(256), 0, 0, pk, np, 0x1234567ull << 20);
  hipLaunchKernelGGL(k_pick, dim3((np + 255) / 256), dim3(256), 0, 0, pk, np, bk, nb);
  CK(hipDeviceSynchronize());
  const long maxP = 2048; long max_cells = maxP * ((np + 4095) / 4096);
  uint32_t *counts, *goff; CK(hipMalloc(&counts, max_cells * 4)); CK(hipMalloc(&goff, max_cells * 4));
  size_t tmp_bytes = 0; CK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, counts, goff, (int)max_cells)); void* tmp; CK(hipMalloc(&tmp, tmp_bytes));
  BRec* brec; uint32_t *bstart, *pstart; uint64_t* ppk; uint32_t* ppi; CK(hipMalloc(&brec, nb * 16)); CK(hipMalloc(&bstart, (maxP + 1) * 4)); CK(hipMalloc(&pstart, (maxP + 1) * 4));
  CK(hipMalloc(&ppk, np * 8)); CK(hipMalloc(&ppi, np * 4));
  uint32_t* found; CK(hipMalloc(&found, np * 4)); unsigned long long* total; CK(hipMalloc(&total, 16));
  uint32_t* fcnt; long nfb = (np + 4095) / 4096; CK(hipMalloc(&fcnt, nfb * 4)); uint32_t* foff; CK(hipMalloc(&foff, nfb * 4));
  uint32_t* out_p; uint64_t* out_b; CK(hipMalloc(&out_p, np * 4)); CK(hipMalloc(&out_b, np * 8));
  CK(hipFuncSetAttribute((const void*)k_staged<512, 8, BRec>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  CK(hipFuncSetAttribute((const void*)k_staged<1024, 8, BRec>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  CK(hipFuncSetAttribute((const void*)k_join<1024, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  CK(hipFuncSetAttribute((const void*)k_join<1024, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  CK(hipFuncSetAttribute((const void*)k_join<512, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  CK(hipFuncSetAttribute((const void*)k_hist<512, 8, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  CK(hipFuncSetAttribute((const void*)k_hist<1024, 8, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  for (uint32_t P : {1024u, 1150u, 2048u}) {
    int sbits = 10; while ((1u << sbits) < 2 * (nb / P) + 1024) sbits++;
    partition<512, 8, 16>(bk, nb, P, counts, goff, tmp, tmp_bytes, nullptr, nullptr, brec, bstart, "build");
    partition<512, 8, 16>(pk, np, P, counts, goff, tmp, tmp_bytes, ppk, ppi, nullptr, pstart, "probe");
    partition<1024, 8, 16>(pk, np, P, counts, goff, tmp, tmp_bytes, ppk, ppi, nullptr, pstart, "probe");
    Ev e0, e1, e2, e3, e4, e5;
    for (int variant = 0; variant < 3; variant++) for (int rep = 0; rep < 2; rep++) {
      CK(hipMemsetAsync(total, 0, 16, 0));
      e0.rec(); CK(hipMemsetAsync(found, 0xFF, np * 4, 0)); e1.rec();
      size_t lds = (size_t)(1u << sbits) * 4;
      if (variant == 0) hipLaunchKernelGGL((k_join<1024, 4>), dim3(P), dim3(1024), lds, 0, brec, bstart, ppk, ppi, pstart, sbits, found, total);
      else if (variant == 1) hipLaunchKernelGGL((k_join<1024, 8>), dim3(P), dim3(1024), lds, 0, brec, bstart, ppk, ppi, pstart, sbits, found, total);
      else hipLaunchKernelGGL((k_join<512, 4>), dim3(P), dim3(512), lds, 0, brec, bstart, ppk, ppi, pstart, sbits, found, total);
      e2.rec();
      hipLaunchKernelGGL(k_found_count, dim3(nfb), dim3(256), 0, 0, found, np, fcnt);
      e3.rec(); size_t tb = tmp_bytes; CK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, fcnt, foff, (int)nfb)); e4.rec();
      hipLaunchKernelGGL(k_found_write, dim3(nfb), dim3(256), 0, 0, found, np, foff, out_p, out_b);
      e5.rec(); CK(hipEventSynchronize(e5.ev)); CK(hipGetLastError());
      unsigned long long h[2]; CK(hipMemcpy(h, total, 16, hipMemcpyDeviceToHost));
      if (rep) printf("P=%u slots=%u (%zu B LDS) variant %d: memset %.3f  join %.3f  count %.3f  scan %.3f  write %.3f ms; matches %llu dup %llu\n", P, 1u << sbits, lds, variant,
                      ms(e0, e1), ms(e1, e2), ms(e2, e3), ms(e3, e4), ms(e4, e5), h[0], h[1]);
    }
    // check order + content on a sample
    std::vector<uint32_t> hp(1000); std::vector<uint64_t> hb(1000); CK(hipMemcpy(hp.data(), out_p, 4000, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), out_b, 8000, hipMemcpyDeviceToHost));
    std::vector<uint64_t> hbk(nb), hpk(2000000); CK(hipMemcpy(hbk.data(), bk, nb * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hpk.data(), pk, hpk.size() * 8, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1000; i++) { if (i && hp[i] <= hp[i - 1]) bad++; if (hp[i] < hpk.size() && hbk[hb[i]] != hpk[hp[i]]) bad++; }
    printf("sample check: %d bad of 1000 (first probe rows %u %u %u)\n", bad, hp[0], hp[1], hp[2]);
  }
  return 0;
}
