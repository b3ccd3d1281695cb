// Diagnostic micro-benchmark (not part of the product): Date32/Int32 column vs scalar -> 1 bit per row, 600M rows.
// Variants of k_compare_scalar_fast: rows per lane, persistent vs 1:1 grid, runtime vs compile-time operator, 16-B loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int BLOCK = 256, WAVE = 64;
template <int ROWS, bool PERSIST, bool CTOP, bool BARRIER = false>
__global__ void __launch_bounds__(BLOCK) k_a(int op, const int32_t* v, int32_t s, int64_t n, uint64_t* out) {
  int lane = threadIdx.x & 63;
  int64_t base = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * (WAVE * ROWS);
  int64_t stride = PERSIST ? (int64_t)gridDim.x * (BLOCK / WAVE) * (WAVE * ROWS) : n;
  for (; base < n; base += stride) {
    int32_t x[ROWS];
    bool full = base + WAVE * ROWS <= n;
#pragma unroll
    for (int r = 0; r < ROWS; r++) { int64_t j = base + r * WAVE + lane; x[r] = (full || j < n) ? v[j] : s; }
    if (BARRIER) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      int64_t j = base + r * WAVE + lane; bool b;
      if (CTOP) b = x[r] > s;
      else switch (op) { case 0: b = x[r] == s; break; case 1: b = x[r] != s; break; case 2: b = x[r] < s; break; case 3: b = x[r] <= s; break; case 4: b = x[r] > s; break; default: b = x[r] >= s; }
      uint64_t m = __ballot(b && j < n);
      if (lane == 0 && base + r * WAVE < n) out[(base >> 6) + r] = m;
    }
  }
}
// full chunks take unconditional loads (a bounds-checked load becomes a branch + s_waitcnt vmcnt(0) per load: no memory parallelism)
template <int ROWS, int OP, bool FULL>
__device__ inline void chunk_b(const int32_t* v, int32_t s, int64_t n, uint64_t* out, int64_t base, int lane) {
  int32_t x[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; r++) { int64_t j = base + r * WAVE + lane; x[r] = (FULL || j < n) ? v[j] : s; }
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    int64_t j = base + r * WAVE + lane; bool b;
    switch (OP) { case 0: b = x[r] == s; break; case 1: b = x[r] != s; break; case 2: b = x[r] < s; break; case 3: b = x[r] <= s; break; case 4: b = x[r] > s; break; default: b = x[r] >= s; }
    uint64_t m = __ballot(b && (FULL || j < n));
    if (lane == 0 && (FULL || base + r * WAVE < n)) out[(base >> 6) + r] = m;
  }
}
template <int ROWS, int OP>
__global__ void __launch_bounds__(BLOCK) k_b(const int32_t* v, int32_t s, int64_t n, uint64_t* out) {
  int lane = threadIdx.x & 63;
  int64_t base = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * (WAVE * ROWS);
  if (base + WAVE * ROWS <= n) chunk_b<ROWS, OP, true>(v, s, n, out, base, lane);
  else if (base < n) chunk_b<ROWS, OP, false>(v, s, n, out, base, lane);
}
__device__ inline uint64_t spread4(uint64_t x) {      // bit i (i < 16) -> bit 4i
  x &= 0xFFFFull; x = (x | (x << 24)) & 0x000000FF000000FFull; x = (x | (x << 12)) & 0x000F000F000F000Full;
  x = (x | (x << 6)) & 0x0303030303030303ull; x = (x | (x << 3)) & 0x1111111111111111ull; return x;
}
// 16-B loads: lane l owns rows base + 256 q + 4 l .. + 3
template <int LOADS>
__global__ void __launch_bounds__(BLOCK) k_v(const int32_t* v, int32_t s, int64_t n, uint64_t* out) {
  int lane = threadIdx.x & 63;
  int64_t base = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * (WAVE * 4 * LOADS);
  if (base + WAVE * 4 * LOADS > n) return;        // microbench: n is a multiple
  int4 x[LOADS];
#pragma unroll
  for (int q = 0; q < LOADS; q++) x[q] = *(const int4*)(v + base + q * 256 + 4 * lane);
#pragma unroll
  for (int q = 0; q < LOADS; q++) {
    uint64_t b0 = __ballot(x[q].x > s), b1 = __ballot(x[q].y > s), b2 = __ballot(x[q].z > s), b3 = __ballot(x[q].w > s);
    if (lane == 0) {
#pragma unroll
      for (int w = 0; w < 4; w++) out[(base >> 6) + q * 4 + w] = spread4(b0 >> (16 * w)) | (spread4(b1 >> (16 * w)) << 1) | (spread4(b2 >> (16 * w)) << 2) | (spread4(b3 >> (16 * w)) << 3);
    }
  }
}
__global__ void init(int32_t* v, int64_t n) { int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; if (i < n) { uint64_t x = i * 0x9E3779B97F4A7C15ull; v[i] = 8035 + (int32_t)((x >> 40) % 2405); } }
template <typename F> float timeit(F f) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); for (int i = 0; i < 2; i++) f(); hipEventRecord(a); for (int i = 0; i < 5; i++) f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms / 5; }
int main() {
  int64_t n = 600000000 / 2048 * 2048; int32_t* v; uint64_t* out;
  CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&out, n / 8 + 64));
  hipLaunchKernelGGL(init, dim3((n + 255) / 256), dim3(256), 0, 0, v, n); CK(hipDeviceSynchronize());
  int pg = 256 * 8;
#define RUN(NAME, K, GRID, ...) { float ms = timeit([&] { hipLaunchKernelGGL(K, dim3(GRID), dim3(BLOCK), 0, 0, __VA_ARGS__); }); printf("%-46s %.3f ms = %.2f TB/s\n", NAME, ms, n * 4.125 / ms / 1e9); }
  RUN("current: 4 rows, persistent, runtime op", (k_a<4, true, false>), pg, 4, v, 9204, n, out);
  RUN("4 rows, persistent, compile-time op", (k_a<4, true, true>), pg, 4, v, 9204, n, out);
  RUN("8 rows, persistent, compile-time op", (k_a<8, true, true>), pg, 4, v, 9204, n, out);
  RUN("16 rows, persistent, compile-time op", (k_a<16, true, true>), pg, 4, v, 9204, n, out);
  RUN("4 rows, 1:1 grid, compile-time op", (k_a<4, false, true>), (unsigned)(n / (BLOCK * 4)), 4, v, 9204, n, out);
  RUN("8 rows, 1:1 grid, compile-time op", (k_a<8, false, true>), (unsigned)(n / (BLOCK * 8)), 4, v, 9204, n, out);
  RUN("16 rows, 1:1 grid, compile-time op", (k_a<16, false, true>), (unsigned)(n / (BLOCK * 16)), 4, v, 9204, n, out);
  RUN("8 rows, 1:1 grid, runtime op", (k_a<8, false, false>), (unsigned)(n / (BLOCK * 8)), 4, v, 9204, n, out);
  RUN("8 rows, persistent, runtime op", (k_a<8, true, false>), pg, 4, v, 9204, n, out);
  RUN("16 rows, 1:1 grid, runtime op", (k_a<16, false, false>), (unsigned)(n / (BLOCK * 16)), 4, v, 9204, n, out);
  RUN("4 rows, 1:1 grid, runtime op", (k_a<4, false, false>), (unsigned)(n / (BLOCK * 4)), 4, v, 9204, n, out);
  RUN("2 rows, 1:1 grid, runtime op", (k_a<2, false, false>), (unsigned)(n / (BLOCK * 2)), 4, v, 9204, n, out);
  RUN("16 rows, 1:1, compile-time op + sched_barrier", (k_a<16, false, true, true>), (unsigned)(n / (BLOCK * 16)), 4, v, 9204, n, out);
  RUN("8 rows, 1:1, compile-time op + sched_barrier", (k_a<8, false, true, true>), (unsigned)(n / (BLOCK * 8)), 4, v, 9204, n, out);
  RUN("16 rows, 1:1, runtime op + sched_barrier", (k_a<16, false, false, true>), (unsigned)(n / (BLOCK * 16)), 4, v, 9204, n, out);
  RUN("32 rows, 1:1, runtime op", (k_a<32, false, false>), (unsigned)(n / (BLOCK * 32)), 4, v, 9204, n, out);
  RUN("FULL-chunk template, 8 rows, 1:1", (k_b<8, 4>), (unsigned)(n / (BLOCK * 8)), v, 9204, n, out);
  RUN("FULL-chunk template, 16 rows, 1:1", (k_b<16, 4>), (unsigned)(n / (BLOCK * 16)), v, 9204, n, out);
  RUN("FULL-chunk template, 32 rows, 1:1", (k_b<32, 4>), (unsigned)(n / (BLOCK * 32)), v, 9204, n, out);
  RUN("16-B loads x1 (4 rows), 1:1", (k_v<1>), (unsigned)(n / (BLOCK * 4)), v, 9204, n, out);
  RUN("16-B loads x2 (8 rows), 1:1", (k_v<2>), (unsigned)(n / (BLOCK * 8)), v, 9204, n, out);
  RUN("16-B loads x4 (16 rows), 1:1", (k_v<4>), (unsigned)(n / (BLOCK * 16)), v, 9204, n, out);
  return 0;
}
