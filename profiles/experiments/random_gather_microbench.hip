// Experiment (not product code): rate of independent random 4-byte / 8-byte reads from a table of T bytes (L2 -> Infinity Cache -> HBM),
// driven by a streamed key column (8 B per probe, hashed on the fly), ILP rows per lane in flight.  Decides up to which build size a
// NON-partitioned, order-preserving hash probe over a compact table beats radix partitioning of the probe side.
// Build: hipcc --offload-arch=gfx950 -O3 -o random_gather_microbench.bin random_gather_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__host__ __device__ inline uint64_t mix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL; return x ^ (x >> 31);
}
template <typename T, int ILP>
__global__ void __launch_bounds__(256) k_gather(const uint64_t* keys, long n, const T* table, uint64_t mask, uint64_t* match_bits) {
  const long base = (long)blockIdx.x * 256 * ILP + threadIdx.x;
  uint64_t k[ILP]; T v[ILP];
#pragma unroll
  for (int q = 0; q < ILP; q++) { long i = base + (long)q * 256; k[q] = keys[i < n ? i : n - 1]; }
#pragma unroll
  for (int q = 0; q < ILP; q++) v[q] = table[mix64(k[q]) & mask];
#pragma unroll
  for (int q = 0; q < ILP; q++) {
    long i = base + (long)q * 256;
    uint64_t m = __ballot((uint64_t)v[q] == (k[q] & 0xFF));
    if ((threadIdx.x & 63) == 0 && i < n) match_bits[i >> 6] = m;
  }
}
__global__ void k_fill(uint64_t* k, long n, uint64_t seed) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) k[i] = mix64(seed + (uint64_t)i); }
template <typename T, int ILP>
static void run(const uint64_t* keys, long n, void* table, long slots, uint64_t* bits) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float best = 1e9;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather<T, ILP>), dim3((n + 256 * ILP - 1) / (256 * ILP)), dim3(256), 0, 0, keys, n, (const T*)table, (uint64_t)slots - 1, bits);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("table %7.1f MB (%ld slots x %zu B) ILP %d: %.3f ms per %ld probes = %.1f G probes/s\n", slots * sizeof(T) / 1e6, slots, sizeof(T), ILP, best, n, n / best / 1e6);
}
int main() {
  const long n = 150000000;
  uint64_t *keys, *bits; void* table; CK(hipMalloc(&keys, n * 8)); CK(hipMalloc(&bits, n / 8 + 64)); CK(hipMalloc(&table, 1l << 30)); CK(hipMemset(table, 1, 1l << 30));
  hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, 0, keys, n, 77ull); CK(hipDeviceSynchronize());
  for (long mb : {2l, 8l, 16l, 32l, 64l, 128l, 256l, 512l, 1024l}) {
    long bytes = mb << 20;
    run<uint32_t, 4>(keys, n, table, bytes / 4, bits);
    run<uint32_t, 8>(keys, n, table, bytes / 4, bits);
    run<uint64_t, 4>(keys, n, table, bytes / 8, bits);
    run<uint64_t, 8>(keys, n, table, bytes / 8, bits);
  }
  return 0;
}
