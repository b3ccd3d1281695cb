// Experiment (not product code): how fast can 150 M (key, row id) pairs be hash-partitioned into P partitions in ONE pass on gfx950,
// and how fast is the LDS-resident build + probe of the partitions afterwards?  Decides the fan-out / pass structure of the
// radix-partitioned hash join (csrc/pjoin.hip).
//   hist      per tile: LDS histogram of the partition ids -> counts[p][tile]
//   direct    per tile: rank by LDS atomic, every lane stores its (key, row) straight to goff[p][tile] + rank (scattered 8 B + 4 B stores;
//             full lines only form if the L2 merges the neighbouring tiles' stores)
//   staged    per tile: counting sort of the tile inside LDS, then a linear write-out (consecutive lanes -> consecutive slots of a run)
//   xcd       tile index remapped so that the tiles an XCD works on concurrently are neighbours (their runs share cache lines)
//   join      one workgroup per partition: LDS open-addressing table (u64 key + u32 row), probe the partition's rows, store found[row]
// Build: hipcc --offload-arch=gfx950 -O3 -o radix_partition_microbench.bin radix_partition_microbench.hip
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__host__ __device__ inline uint64_t mix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL; return x ^ (x >> 31);
}
__device__ inline uint32_t pid_of(uint64_t key, int pbits) { return (uint32_t)(mix64(key) >> (64 - pbits)); }

__device__ inline long map_tile(long b, long ntiles, int xcd) {
  if (!xcd) return b;
  // blocks b, b+8, b+16.. share an XCD (round robin): give XCD x the contiguous tile range [x * per, (x+1) * per)
  long per = (ntiles + 7) / 8; long x = b & 7, j = b >> 3; long t = x * per + j; return t;   // caller checks t < ntiles && j < per
}

template <int NT, int R>
__global__ void __launch_bounds__(NT) k_hist(const uint64_t* keys, long n, int pbits, long ntiles, uint32_t* counts /*[P][ntiles]*/) {
  extern __shared__ uint32_t lds[];
  const int P = 1 << pbits; const long t = blockIdx.x; const long base = t * (long)(NT * R);
  for (int p = threadIdx.x; p < P; p += NT) lds[p] = 0;
  __syncthreads();
  uint64_t k[R];
#pragma unroll
  for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; k[q] = i < n ? keys[i] : 0; }
#pragma unroll
  for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; if (i < n) atomicAdd(&lds[pid_of(k[q], pbits)], 1u); }
  __syncthreads();
  for (int p = threadIdx.x; p < P; p += NT) counts[(long)p * ntiles + t] = lds[p];
}

template <int NT, int R>
__global__ void __launch_bounds__(NT) k_direct(const uint64_t* keys, long n, int pbits, long ntiles, const uint32_t* goff, int xcd, uint64_t* out_key, uint32_t* out_idx) {
  extern __shared__ uint32_t lds[];
  const int P = 1 << pbits;
  long per = (ntiles + 7) / 8; long t = xcd ? ((long)(blockIdx.x & 7) * per + (blockIdx.x >> 3)) : (long)blockIdx.x;
  if (xcd && ((long)(blockIdx.x >> 3) >= per || t >= ntiles)) return;
  if (t >= ntiles) return;
  const long base = t * (long)(NT * R);
  for (int p = threadIdx.x; p < P; p += NT) lds[p] = goff[(long)p * ntiles + t];
  __syncthreads();
  uint64_t k[R];
#pragma unroll
  for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; k[q] = i < n ? keys[i] : 0; }
#pragma unroll
  for (int q = 0; q < R; q++) {
    long i = base + (long)q * NT + threadIdx.x;
    if (i < n) { uint32_t pos = atomicAdd(&lds[pid_of(k[q], pbits)], 1u); out_key[pos] = k[q]; out_idx[pos] = (uint32_t)i; }
  }
}

// staged: LDS = start[P] (after scan), delta[P], stage_key[TILE], stage_idx[TILE]
template <int NT, int R>
__global__ void __launch_bounds__(NT) k_staged(const uint64_t* keys, long n, int pbits, long ntiles, const uint32_t* goff, int xcd, uint64_t* out_key, uint32_t* out_idx) {
  extern __shared__ uint32_t lds[];
  constexpr int TILE = NT * R;
  const int P = 1 << pbits;
  uint32_t* cnt = lds; int32_t* delta = (int32_t*)(lds + P); uint32_t* sidx = lds + 2 * P; uint64_t* skey = (uint64_t*)(lds + 2 * P + TILE);
  __shared__ uint32_t wsum[NT / 64];
  long per = (ntiles + 7) / 8; long t = xcd ? ((long)(blockIdx.x & 7) * per + (blockIdx.x >> 3)) : (long)blockIdx.x;
  if (xcd && ((long)(blockIdx.x >> 3) >= per || t >= ntiles)) return;
  if (t >= ntiles) return;
  const long base = t * (long)TILE;
  for (int p = threadIdx.x; p < P; p += NT) cnt[p] = 0;
  __syncthreads();
  uint64_t k[R]; uint32_t pid[R], rk[R];
#pragma unroll
  for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; k[q] = i < n ? keys[i] : 0; }
#pragma unroll
  for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; pid[q] = pid_of(k[q], pbits); rk[q] = i < n ? atomicAdd(&cnt[pid[q]], 1u) : 0; }
  __syncthreads();
  // exclusive scan of cnt[P] (P <= NT * 8): each thread owns P/NT consecutive bins
  {
    const int per_t = (P + NT - 1) / NT; uint32_t loc[8]; uint32_t s = 0;
    for (int j = 0; j < per_t; j++) { int p = threadIdx.x * per_t + j; loc[j] = p < P ? cnt[p] : 0; s += loc[j]; }
    uint32_t inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if ((threadIdx.x & 63) >= d) inc += o; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t wbase = 0; for (int w = 0; w < (int)(threadIdx.x >> 6); w++) wbase += wsum[w];
    uint32_t run = wbase + inc - s;
    for (int j = 0; j < per_t; j++) { int p = threadIdx.x * per_t + j; if (p < P) { cnt[p] = run; delta[p] = (int32_t)goff[(long)p * ntiles + t] - (int32_t)run; run += loc[j]; } }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < R; q++) { long i = base + (long)q * NT + threadIdx.x; if (i < n) { uint32_t s = cnt[pid[q]] + rk[q]; skey[s] = k[q]; sidx[s] = (uint32_t)i; } }
  __syncthreads();
  long left = n - base; int m = left < TILE ? (int)left : TILE;
  for (int i = threadIdx.x; i < m; i += NT) { uint64_t kk = skey[i]; uint32_t p = pid_of(kk, pbits); long pos = (long)delta[p] + i; out_key[pos] = kk; out_idx[pos] = sidx[i]; }
}

// join: one workgroup per partition; LDS table S slots (keys u64, rows u32)
template <int NT>
__global__ void __launch_bounds__(NT) k_join(const uint64_t* bkey, const uint32_t* bidx, const uint32_t* bstart, const uint64_t* pkey, const uint32_t* pidx, const uint32_t* pstart,
                                            int sbits, uint32_t* found, unsigned long long* total) {
  extern __shared__ uint32_t lds[];
  const uint32_t S = 1u << sbits, M = S - 1;
  uint64_t* tk = (uint64_t*)lds; uint32_t* tv = lds + 2 * S;
  const int p = blockIdx.x;
  for (uint32_t s = threadIdx.x; s < S; s += NT) tk[s] = ~0ull;
  __syncthreads();
  uint32_t b0 = bstart[p], b1 = bstart[p + 1];
  for (uint32_t i = b0 + threadIdx.x; i < b1; i += NT) {
    uint64_t k = bkey[i]; uint32_t s = (uint32_t)mix64(k) & M;
    for (;;) {
      unsigned long long old = atomicCAS((unsigned long long*)&tk[s], ~0ull, (unsigned long long)k);
      if (old == ~0ull) { tv[s] = bidx[i]; break; }
      if (old == k) break;     // duplicate key (bench data has none)
      s = (s + 1) & M;
    }
  }
  __syncthreads();
  uint32_t q0 = pstart[p], q1 = pstart[p + 1]; unsigned cntm = 0;
  for (uint32_t i = q0 + threadIdx.x; i < q1; i += NT) {
    uint64_t k = pkey[i]; uint32_t s = (uint32_t)mix64(k) & M;
    for (;;) {
      uint64_t c = tk[s];
      if (c == k) { found[pidx[i]] = tv[s]; cntm++; break; }
      if (c == ~0ull) break;
      s = (s + 1) & M;
    }
  }
  for (int d = 32; d > 0; d >>= 1) cntm += __shfl_xor(cntm, d, 64);
  if ((threadIdx.x & 63) == 0 && cntm) atomicAdd(total, (unsigned long long)cntm);
}

__global__ void k_starts(const uint32_t* goff, long ntiles, int P, long n, uint32_t* start) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) start[p] = goff[(long)p * ntiles];
  if (p == P) start[P] = (uint32_t)n;
}
__global__ void k_fill(uint64_t* k, long n, uint64_t seed) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) k[i] = mix64(seed + (uint64_t)i * 0x9E3779B97F4A7C15ull) >> 2; }
__global__ void k_pick(uint64_t* pk, long n, const uint64_t* bk, long nb) {     // every 5th probe key takes a build key
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n && i % 5 == 0) pk[i] = bk[mix64((uint64_t)i) % (uint64_t)nb];
}

struct Part { uint64_t* key; uint32_t* idx; uint32_t* start; };

template <int NT, int R>
static float run_partition(const char* what, int variant, int xcd, const uint64_t* keys, long n, int pbits, Part out, uint32_t* counts, uint32_t* goff, void* tmp, size_t tmp_bytes, bool print) {
  const int TILE = NT * R; const int P = 1 << pbits; long ntiles = (n + TILE - 1) / TILE;
  hipEvent_t e0, e1, e2, e3; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
  long grid = xcd ? ((ntiles + 7) / 8) * 8 : ntiles;
  float best = 1e9, bh = 0, bs = 0, bp = 0;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_hist<NT, R>), dim3(ntiles), dim3(NT), P * 4, 0, keys, n, pbits, ntiles, counts);
    CK(hipEventRecord(e1));
    size_t tb = tmp_bytes; CK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, counts, goff, (int)((long)P * ntiles)));
    CK(hipEventRecord(e2));
    if (variant == 0) hipLaunchKernelGGL((k_direct<NT, R>), dim3(grid), dim3(NT), P * 4, 0, keys, n, pbits, ntiles, goff, xcd, out.key, out.idx);
    else hipLaunchKernelGGL((k_staged<NT, R>), dim3(grid), dim3(NT), (2 * P + 3 * TILE) * 4, 0, keys, n, pbits, ntiles, goff, xcd, out.key, out.idx);
    CK(hipEventRecord(e3)); CK(hipEventSynchronize(e3)); CK(hipGetLastError());
    float h, s, p; CK(hipEventElapsedTime(&h, e0, e1)); CK(hipEventElapsedTime(&s, e1, e2)); CK(hipEventElapsedTime(&p, e2, e3));
    if (h + s + p < best) { best = h + s + p; bh = h; bs = s; bp = p; }
  }
  hipLaunchKernelGGL(k_starts, dim3((P + 256) / 256), dim3(256), 0, 0, goff, ntiles, P, n, out.start);
  CK(hipDeviceSynchronize());
  if (print) printf("%-8s P=%5d tile=%5d xcd=%d rows=%ld : hist %.3f ms  scan %.3f ms  scatter %.3f ms (%.0f GB/s of 8 B in + 12 B out)  total %.3f ms\n",
                    what, P, TILE, xcd, n, bh, bs, bp, n * 20.0 / bp / 1e6, best);
  return best;
}

int main(int argc, char** argv) {
  long nb = argc > 1 ? atol(argv[1]) : 15000000, np = argc > 2 ? atol(argv[2]) : 150000000;
  uint64_t *bk, *pk; CK(hipMalloc(&bk, nb * 8)); CK(hipMalloc(&pk, np * 8));
  hipLaunchKernelGGL(k_fill, dim3((nb + 255) / 256), dim3(256), 0, 0, bk, nb, 1ull);
  hipLaunchKernelGGL(k_fill, dim3((np + 255) / 256), dim3(256), 0, 0, pk, np, 0x1234567ull << 20);
  hipLaunchKernelGGL(k_pick, dim3((np + 255) / 256), dim3(256), 0, 0, pk, np, bk, nb);
  CK(hipDeviceSynchronize());
  const int maxP = 8192; long min_tile = 2048; long max_cells = (long)maxP * ((np + min_tile - 1) / min_tile);
  uint32_t *counts, *goff; CK(hipMalloc(&counts, max_cells * 4)); CK(hipMalloc(&goff, max_cells * 4));
  size_t tmp_bytes = 0; CK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, counts, goff, (int)max_cells)); void* tmp; CK(hipMalloc(&tmp, tmp_bytes));
  Part pb, pp; CK(hipMalloc(&pb.key, nb * 8)); CK(hipMalloc(&pb.idx, nb * 4)); CK(hipMalloc(&pb.start, (maxP + 1) * 4));
  CK(hipMalloc(&pp.key, np * 8)); CK(hipMalloc(&pp.idx, np * 4)); CK(hipMalloc(&pp.start, (maxP + 1) * 4));
  uint32_t* found; CK(hipMalloc(&found, np * 4)); unsigned long long* total; CK(hipMalloc(&total, 8));
  CK(hipFuncSetAttribute((const void*)k_staged<512, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  CK(hipFuncSetAttribute((const void*)k_staged<1024, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  CK(hipFuncSetAttribute((const void*)k_join<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  CK(hipFuncSetAttribute((const void*)k_join<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  // copy ceiling for reference
  { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
    for (int r = 0; r < 2; r++) { CK(hipEventRecord(e0)); CK(hipMemcpyAsync(pp.key, pk, np * 8, hipMemcpyDeviceToDevice, 0)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    printf("d2d copy of %.1f GB: %.3f ms = %.0f GB/s (read + write)\n", np * 8 / 1e9, ms, np * 16.0 / ms / 1e6); }
  for (int pbits : {6, 8, 9, 10, 11, 12}) {
    for (int xcd = 0; xcd < 2; xcd++) {
      run_partition<512, 8>("direct", 0, xcd, pk, np, pbits, pp, counts, goff, tmp, tmp_bytes, true);
      if (pbits <= 12) run_partition<512, 8>("staged", 1, xcd, pk, np, pbits, pp, counts, goff, tmp, tmp_bytes, true);
      if (pbits <= 12) run_partition<1024, 8>("staged", 1, xcd, pk, np, pbits, pp, counts, goff, tmp, tmp_bytes, true);
    }
  }
  // join at P = 4096 (15 M build rows -> ~3.7 K rows per partition -> 8192 slots = 96 KB LDS)
  for (int pbits : {12, 13}) {
    int P = 1 << pbits; int sbits = 1; while ((1l << sbits) < 2 * (nb / P) + 512) sbits++;
    run_partition<512, 8>("build", 0, 1, bk, nb, pbits, pb, counts, goff, tmp, tmp_bytes, true);
    run_partition<512, 8>("probe", 0, 1, pk, np, pbits, pp, counts, goff, tmp, tmp_bytes, true);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipMemsetAsync(found, 0xFF, np * 4, 0)); CK(hipMemsetAsync(total, 0, 8, 0));
      CK(hipEventRecord(e0));
      size_t lds = (size_t)(1u << sbits) * 12;
      if (lds <= 64 * 1024) hipLaunchKernelGGL((k_join<512>), dim3(P), dim3(512), lds, 0, pb.key, pb.idx, pb.start, pp.key, pp.idx, pp.start, sbits, found, total);
      else hipLaunchKernelGGL((k_join<1024>), dim3(P), dim3(1024), lds, 0, pb.key, pb.idx, pb.start, pp.key, pp.idx, pp.start, sbits, found, total);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError()); CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long h; CK(hipMemcpy(&h, total, 8, hipMemcpyDeviceToHost));
      printf("join P=%d slots=%d (%zu B LDS): %.3f ms, %llu matches\n", P, 1 << sbits, lds, ms, h);
    }
  }
  return 0;
}
