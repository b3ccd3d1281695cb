// Experiment (not product code): are scatter-adds into a large group-state array faster when every state word is only ever updated
// from ONE XCD -- workgroup-scope atomics that execute in that XCD's L2 -- than as device-scope atomics (executed at the memory side)?
// Each workgroup reads HW_REG_XCC_ID, pulls 4096-row tiles from its XCD's queue and applies only rows with (gid & 7) == xcc, so all
// 8 XCDs scan all rows (8x read amplification of the id column) but each state word has a single home L2.
// Build: hipcc --offload-arch=gfx950 -O3 -o xcd_atomics_microbench.bin xcd_atomics_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_direct(const unsigned* g, const double* v, long n, double* acc) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) atomicAdd(&acc[g[i]], v[i]);
}
__device__ inline unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }      // HW_REG_XCC_ID, bits [3:0]
__global__ void k_xcd(const unsigned* g, const double* v, long n, double* acc, unsigned* next /*[8]*/, unsigned* seen /*[8]*/) {
  const unsigned x = xcc_id();
  __shared__ unsigned tile;
  const long ntiles = (n + 4095) / 4096;
  if (threadIdx.x == 0) atomicAdd(&seen[x], 1u);
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) tile = atomicAdd(&next[x], 1u);
    __syncthreads();
    const long t = tile;
    if (t >= ntiles) break;                               // every wave reaches this once its XCD's queue is drained
    const long base = t * 4096;
#pragma unroll 4
    for (int q = 0; q < 16; q++) {
      long i = base + q * 256 + threadIdx.x;
      if (i < n) { unsigned gg = g[i]; if ((gg & 7u) == x) __hip_atomic_fetch_add(&acc[gg], v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    }
  }
}
// Variant 3: no row partitioning -- every XCD owns a full REPLICA of the state array; a workgroup adds its rows into the replica of the
// XCD it runs on (workgroup-scope atomics, that XCD's L2), a second kernel sums the 8 replicas.
__global__ void k_replica(const unsigned* g, const double* v, long n, double* rep /*[8][G]*/, long G) {
  double* mine = rep + (long)xcc_id() * G;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    __hip_atomic_fetch_add(&mine[g[i]], v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ void k_reduce8(const double* rep, long G, double* acc) {
  long k = (long)blockIdx.x * blockDim.x + threadIdx.x; if (k >= G) return;
  double s = 0; for (int x = 0; x < 8; x++) s += rep[(long)x * G + k]; acc[k] = s;
}
int main() {
  const long n = 100000000; const unsigned G = 1000000;
  std::vector<unsigned> hg(n); std::vector<double> hv(n);
  unsigned long long s = 88172645463325252ull;
  for (long i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hg[i] = (unsigned)(s % G); hv[i] = (double)((s >> 40) & 1023); }
  unsigned *g, *next, *seen; double *v, *a0, *a1;
  CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&v, n * 8)); CK(hipMalloc(&a0, G * 8)); CK(hipMalloc(&a1, G * 8)); CK(hipMalloc(&next, 32)); CK(hipMalloc(&seen, 32));
  CK(hipMemcpy(g, hg.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v, hv.data(), n * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipMemset(a0, 0, G * 8)); CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_direct, dim3(2048), dim3(256), 0, 0, g, v, n, a0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("device-scope atomics: %.3f ms\n", ms);
    CK(hipMemset(a1, 0, G * 8)); CK(hipMemset(next, 0, 32)); CK(hipMemset(seen, 0, 32));
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_xcd, dim3(2048), dim3(256), 0, 0, g, v, n, a1, next, seen); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("XCD-partitioned workgroup-scope atomics: %.3f ms\n", ms);
  }
  double *rep, *a2; CK(hipMalloc(&rep, 8l * G * 8)); CK(hipMalloc(&a2, G * 8));
  for (int r = 0; r < 3; r++) {
    CK(hipMemset(rep, 0, 8l * G * 8)); CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_replica, dim3(2048), dim3(256), 0, 0, g, v, n, rep, (long)G);
    hipLaunchKernelGGL(k_reduce8, dim3((G + 255) / 256), dim3(256), 0, 0, rep, (long)G, a2);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); printf("per-XCD replicas, workgroup-scope atomics + reduce: %.3f ms\n", ms);
  }
  { std::vector<double> r2(G), rr(G); CK(hipMemcpy(r2.data(), a2, G * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(rr.data(), a0, G * 8, hipMemcpyDeviceToHost));
    long bad2 = 0; for (unsigned k = 0; k < G; k++) if (r2[k] != rr[k]) bad2++; printf("replica variant mismatching groups: %ld\n", bad2); }
  std::vector<double> r0(G), r1(G); unsigned hs[8], hn[8];
  CK(hipMemcpy(r0.data(), a0, G * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), a1, G * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hs, seen, 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(hn, next, 32, hipMemcpyDeviceToHost));
  long bad = 0; for (unsigned k = 0; k < G; k++) if (r0[k] != r1[k]) bad++;            // integer-valued doubles: exact in any order
  printf("workgroups per XCD:"); for (int x = 0; x < 8; x++) printf(" %u", hs[x]); printf("\nmismatching groups: %ld of %u\n", bad, G);
  return 0;
}
