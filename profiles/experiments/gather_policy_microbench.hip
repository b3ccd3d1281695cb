// Experiment (not product code): random 8-byte reads from a 75 MB bitmap (TPC-H Q3's orders membership bitmap at SF100, probed by lineitem keys in no order) under different
// cache policies of the load (gfx950 global_load ... sc0 / sc1 / nt) and widths: does any of them move less than a 128-byte line per probe from the Infinity Cache?
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_policy_microbench.bin gather_policy_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__host__ __device__ inline uint64_t mix64(uint64_t x) { x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL; return x ^ (x >> 31); }
constexpr int ILP = 8;
#define LOADS(SUFFIX) \
  _Pragma("unroll") for (int q = 0; q < ILP; q++) asm volatile("global_load_dwordx2 %0, %1, off " SUFFIX : "=v"(v[q]) : "v"(p[q]) : "memory"); \
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
#define LOADS4(SUFFIX) \
  _Pragma("unroll") for (int q = 0; q < ILP; q++) { uint32_t t; asm volatile("global_load_dword %0, %1, off " SUFFIX : "=v"(t) : "v"((const uint32_t*)p[q] + 0) : "memory"); v4[q] = t; } \
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v4[0]), "+v"(v4[1]), "+v"(v4[2]), "+v"(v4[3]), "+v"(v4[4]), "+v"(v4[5]), "+v"(v4[6]), "+v"(v4[7]) : : "memory");
template <int POLICY>
__global__ void __launch_bounds__(256) k_gather(const uint64_t* keys, long n, const uint64_t* table, uint64_t nbits, uint64_t* match_bits) {
  const long base = (long)blockIdx.x * 256 * ILP + threadIdx.x;
  uint64_t d[ILP]; const uint64_t* p[ILP]; uint64_t v[ILP]; uint32_t v4[ILP];
#pragma unroll
  for (int q = 0; q < ILP; q++) { long i = base + (long)q * 256; d[q] = mix64(keys[i < n ? i : n - 1]) % nbits; p[q] = table + (d[q] >> 6); }
  if (POLICY == 0) {
#pragma unroll
    for (int q = 0; q < ILP; q++) v[q] = *p[q];
  } else if (POLICY == 1) {
#pragma unroll
    for (int q = 0; q < ILP; q++) v[q] = __builtin_nontemporal_load(p[q]);
  } else if (POLICY == 2) { LOADS("sc0") } else if (POLICY == 3) { LOADS("sc1") } else if (POLICY == 4) { LOADS("sc0 sc1") } else if (POLICY == 5) { LOADS("nt") }
  else if (POLICY == 6) { LOADS("sc0 nt") } else if (POLICY == 7) { LOADS("sc1 nt") } else if (POLICY == 8) { LOADS("sc0 sc1 nt") }
  else if (POLICY == 9) {            // 4-byte loads, plain
#pragma unroll
    for (int q = 0; q < ILP; q++) v[q] = (uint64_t)((const uint32_t*)table)[d[q] >> 5] << (d[q] & 32);
  } else if (POLICY == 10) { LOADS4("sc1") _Pragma("unroll") for (int q = 0; q < ILP; q++) v[q] = v4[q]; }
  else if (POLICY == 11) { LOADS4("nt") _Pragma("unroll") for (int q = 0; q < ILP; q++) v[q] = v4[q]; }
#pragma unroll
  for (int q = 0; q < ILP; q++) {
    long i = base + (long)q * 256;
    uint64_t m = __ballot((v[q] >> (d[q] & 63)) & 1ull);
    if ((threadIdx.x & 63) == 0 && i < n) match_bits[i >> 6] = m;
  }
}
__global__ void k_fill(uint64_t* k, long n, uint64_t seed) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) k[i] = mix64(seed + (uint64_t)i); }
template <int POLICY>
static void run(const char* what, const uint64_t* keys, long n, const uint64_t* table, uint64_t nbits, uint64_t* bits) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float best = 1e9;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather<POLICY>), dim3((n + 256 * ILP - 1) / (256 * ILP)), dim3(256), 0, 0, keys, n, table, nbits, bits);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("bitmap %6.1f MB  %-22s %.3f ms per %ld probes = %5.1f G probes/s\n", nbits / 8e6, what, best, n, n / best / 1e6);
}
int main() {
  const long n = 280000000;
  uint64_t *keys, *bits, *table; CK(hipMalloc(&keys, n * 8)); CK(hipMalloc(&bits, n / 8 + 64)); CK(hipMalloc(&table, 1l << 28)); CK(hipMemset(table, 0x5A, 1l << 28));
  hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, 0, keys, n, 77ull); CK(hipDeviceSynchronize());
  for (uint64_t nbits : {600000000ull, 75000000ull, 1200000000ull}) {
    run<0>("plain 8 B", keys, n, table, nbits, bits); run<1>("nontemporal builtin", keys, n, table, nbits, bits); run<2>("sc0", keys, n, table, nbits, bits); run<3>("sc1", keys, n, table, nbits, bits);
    run<4>("sc0 sc1", keys, n, table, nbits, bits); run<5>("nt", keys, n, table, nbits, bits); run<6>("sc0 nt", keys, n, table, nbits, bits); run<7>("sc1 nt", keys, n, table, nbits, bits);
    run<8>("sc0 sc1 nt", keys, n, table, nbits, bits); run<9>("plain 4 B", keys, n, table, nbits, bits); run<10>("4 B sc1", keys, n, table, nbits, bits); run<11>("4 B nt", keys, n, table, nbits, bits);
  }
  return 0;
}
