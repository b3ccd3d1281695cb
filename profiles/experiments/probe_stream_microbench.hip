// Diagnostic micro-benchmark (not part of the product): where does k_probe_match_bitmap's time go?
// A: stream Int64 keys -> ballot; B: + selection bitmap; C: + membership bitmap gather (sorted keys); D: C with random keys.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ inline uint64_t spread32(uint64_t x) { x &= 0xFFFFFFFFull; x = (x | (x << 16)) & 0x0000FFFF0000FFFFull; x = (x | (x << 8)) & 0x00FF00FF00FF00FFull; x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full; x = (x | (x << 2)) & 0x3333333333333333ull; x = (x | (x << 1)) & 0x5555555555555555ull; return x; }
template <int MODE, int ROWS, int NT = 0>
__global__ void __launch_bounds__(256) k(const int64_t* keys, const uint64_t* mask, const uint64_t* bitmap, int64_t n, uint64_t range, uint64_t* out) {
  int lane = threadIdx.x & 63;
  int64_t base = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 * ROWS);
  // MODE 3/4: the wave's 64*ROWS clustered keys fall into one 64-word bitmap window: one coalesced load, then LDS / bpermute lookups
  __shared__ uint64_t win_lds[4][64];
  uint64_t win = 0; int64_t w0i = 0;
  if (MODE >= 3 && base < n) { uint64_t d0 = (uint64_t)__builtin_nontemporal_load(keys + base); d0 = __builtin_amdgcn_readfirstlane((uint32_t)d0) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(d0 >> 32)) << 32);
    w0i = (int64_t)(d0 >> 6); int64_t wi = w0i + lane; win = (uint64_t)wi * 64 < range ? bitmap[wi] : 0;
    if (MODE == 3) win_lds[threadIdx.x >> 6][lane] = win; }
  int64_t kk[ROWS / 2][2];
#pragma unroll
  for (int r = 0; r < ROWS / 2; r++) { int64_t j = base + r * 128 + 2 * lane; if (j + 1 < n) { if (NT) { kk[r][0] = __builtin_nontemporal_load(keys + j); kk[r][1] = __builtin_nontemporal_load(keys + j + 1); } else { longlong2 p = *(const longlong2*)(keys + j); kk[r][0] = p.x; kk[r][1] = p.y; } } else { kk[r][0] = kk[r][1] = 0; } }
#pragma unroll
  for (int r = 0; r < ROWS / 2; r++) {
    int64_t j = base + r * 128 + 2 * lane;
    uint64_t mw = 3; if (MODE >= 1 && j < n) mw = (NT ? __builtin_nontemporal_load(mask + (j >> 6)) : mask[j >> 6]) >> (j & 63);
    bool h[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
      uint64_t d = (uint64_t)kk[r][e];
      bool go = j + e < n && ((mw >> e) & 1) && d < range;
      if (MODE >= 3) {
        int64_t rel = (int64_t)(d >> 6) - w0i; bool in = rel >= 0 && rel < 64;
        uint64_t word;
        if (MODE == 3) word = win_lds[threadIdx.x >> 6][rel & 63]; else word = __shfl(win, (int)(rel & 63), 64);
        if (go && !in) word = bitmap[d >> 6];
        h[e] = go && ((word >> (d & 63)) & 1);
      } else
      if (MODE >= 2) h[e] = go ? ((NT == 2 ? __builtin_nontemporal_load(bitmap + (d >> 6)) : (NT == 3 ? (uint64_t)__builtin_nontemporal_load((const uint32_t*)bitmap + (d >> 5)) >> (d & 32) << (d & 32) : bitmap[d >> 6])) >> (d & 63)) & 1 : false; else h[e] = go && (d & 4);
    }
    uint64_t be = __ballot(h[0]), bo = __ballot(h[1]);
    uint64_t w0 = spread32(be) | (spread32(bo) << 1), w1 = spread32(be >> 32) | (spread32(bo >> 32) << 1);
    if (lane == 0 && base + r * 128 < n) { out[(base >> 6) + 2 * r] = w0; out[(base >> 6) + 2 * r + 1] = w1; }
  }
}
// ---- PRODUCT KERNEL (copy of join.hip k_probe_match_bitmap, for A/B against the variants above)
constexpr int BLOCK = 256, WAVE = 64, PM_ROWS = 8;
__device__ inline int lane_id() { return threadIdx.x & 63; }
__device__ inline uint64_t ballot64(bool p) { return __ballot(p); }
__device__ inline bool valid_at(const uint64_t* v, int64_t i) { return v == nullptr || ((v[i >> 6] >> (i & 63)) & 1ull); }
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_probe_match_bitmap(const T* keys, const uint64_t* key_valid, const uint64_t* mask, int64_t n, int64_t kmin, uint64_t range,
                                                              const uint64_t* bitmap, uint64_t* match_bits) {
  int lane = lane_id();
  // one 512-row chunk per wave when launched 1:1 (measured faster than a persistent grid: consecutive workgroups keep
  // the key stream and the bitmap window local); the loop only matters if a caller caps the grid
  for (int64_t base = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * (WAVE * PM_ROWS); base < n;
       base += (int64_t)gridDim.x * (BLOCK / WAVE) * (WAVE * PM_ROWS)) {
  // bitmap window: clustered probe keys (a fact table stored in key order) put the wave's 512 keys inside one run of 64
  // bitmap words, so the wave loads that run once, coalesced, from the first key of its chunk (a scalar load that is in
  // flight together with the vector key loads) and looks bits up with ds_bpermute; keys outside the window take the
  // per-lane gather.  Measured (profiles/experiments/probe_stream_microbench.hip): 1.29 -> 1.05 ms per 600M sorted keys,
  // unchanged for random keys.
  uint64_t d0 = (uint64_t)((int64_t)keys[base] - kmin);
  d0 = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)d0) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(d0 >> 32)) << 32);
  int64_t w0i = d0 < range ? (int64_t)(d0 >> 6) : 0;
  uint64_t win = (uint64_t)(w0i + lane) * 64 < range ? bitmap[w0i + lane] : 0ull;
  T k[PM_ROWS / 2][2];
#pragma unroll
  for (int r = 0; r < PM_ROWS / 2; r++) {                      // lane l owns rows base + 128 r + 2l, +1
    int64_t j = base + r * 2 * WAVE + 2 * lane;
    if (j + 1 < n) { struct alignas(2 * sizeof(T)) P { T a, b; }; P p = *(const P*)(keys + j); k[r][0] = p.a; k[r][1] = p.b; }
    else { k[r][0] = j < n ? keys[j] : (T)0; k[r][1] = 0; }
  }
  uint64_t mw[PM_ROWS / 2];
#pragma unroll
  for (int r = 0; r < PM_ROWS / 2; r++) { int64_t j = base + r * 2 * WAVE + 2 * lane; mw[r] = (mask && j < n) ? mask[j >> 6] >> (j & 63) : 3ull; }
  uint64_t bw[PM_ROWS / 2][2];
#pragma unroll
  for (int r = 0; r < PM_ROWS / 2; r++) {
    int64_t j = base + r * 2 * WAVE + 2 * lane;
#pragma unroll
    for (int e = 0; e < 2; e++) {
      uint64_t d = (uint64_t)((int64_t)k[r][e] - kmin);
      bool go = j + e < n && ((mw[r] >> e) & 1) && valid_at(key_valid, j + e) && d < range;
      int64_t rel = (int64_t)(d >> 6) - w0i;
      uint64_t word = __shfl(win, (int)(rel & 63), 64);
      if (go && (rel < 0 || rel >= WAVE)) word = bitmap[d >> 6];
      bw[r][e] = go ? (word >> (d & 63)) & 1ull : 0ull;
    }
  }
#pragma unroll
  for (int r = 0; r < PM_ROWS / 2; r++) {
    uint64_t be = ballot64(bw[r][0] != 0), bo = ballot64(bw[r][1] != 0);   // wave-uniform: the interleave below runs on the scalar unit
    uint64_t w0 = spread32(be) | (spread32(bo) << 1), w1 = spread32(be >> 32) | (spread32(bo >> 32) << 1);
    int64_t wbase = (base >> 6) + 2 * r;
    if (lane == 0) { if (base + r * 2 * WAVE < n) match_bits[wbase] = w0; if (base + r * 2 * WAVE + WAVE < n) match_bits[wbase + 1] = w1; }
  }
  }
}
// ---- candidate v2 of the product kernel: no chunk loop, one pass over r with the ballots inside (shape of MODE 4 above)
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_probe_match_bitmap_v2(const T* keys, const uint64_t* key_valid, const uint64_t* mask, int64_t n, int64_t kmin, uint64_t range,
                                                                 const uint64_t* bitmap, uint64_t* match_bits) {
  int lane = lane_id();
  int64_t base = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * (WAVE * PM_ROWS);
  if (base >= n) return;
  uint64_t d0 = (uint64_t)((int64_t)keys[base] - kmin);
  d0 = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)d0) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(d0 >> 32)) << 32);
  int64_t w0i = d0 < range ? (int64_t)(d0 >> 6) : 0;
  uint64_t win = (uint64_t)(w0i + lane) * 64 < range ? bitmap[w0i + lane] : 0ull;
  T k[PM_ROWS / 2][2];
#pragma unroll
  for (int r = 0; r < PM_ROWS / 2; r++) {
    int64_t j = base + r * 2 * WAVE + 2 * lane;
    if (j + 1 < n) { struct alignas(2 * sizeof(T)) P { T a, b; }; P p = *(const P*)(keys + j); k[r][0] = p.a; k[r][1] = p.b; }
    else { k[r][0] = j < n ? keys[j] : (T)0; k[r][1] = 0; }
  }
#pragma unroll
  for (int r = 0; r < PM_ROWS / 2; r++) {
    int64_t j = base + r * 2 * WAVE + 2 * lane;
    uint64_t mw = (mask && j < n) ? mask[j >> 6] >> (j & 63) : 3ull;
    bool h[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
      uint64_t d = (uint64_t)((int64_t)k[r][e] - kmin);
      bool go = j + e < n && ((mw >> e) & 1) && valid_at(key_valid, j + e) && d < range;
      int64_t rel = (int64_t)(d >> 6) - w0i;
      uint64_t word = __shfl(win, (int)(rel & 63), 64);
      if (go && (rel < 0 || rel >= WAVE)) word = bitmap[d >> 6];
      h[e] = go && ((word >> (d & 63)) & 1ull);
    }
    uint64_t be = ballot64(h[0]), bo = ballot64(h[1]);
    uint64_t w0 = spread32(be) | (spread32(bo) << 1), w1 = spread32(be >> 32) | (spread32(bo >> 32) << 1);
    int64_t wbase = (base >> 6) + 2 * r;
    if (lane == 0) { if (base + r * 2 * WAVE < n) match_bits[wbase] = w0; if (base + r * 2 * WAVE + WAVE < n) match_bits[wbase + 1] = w1; }
  }
}
// ---- candidate v3 (v2 + nontemporal first-key load) of the product kernel: no chunk loop, one pass over r with the ballots inside (shape of MODE 4 above)
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_probe_match_bitmap_v3(const T* keys, const uint64_t* key_valid, const uint64_t* mask, int64_t n, int64_t kmin, uint64_t range,
                                                                 const uint64_t* bitmap, uint64_t* match_bits) {
  int lane = lane_id();
  int64_t base = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * (WAVE * PM_ROWS);
  if (base >= n) return;
  uint64_t d0 = (uint64_t)((int64_t)__builtin_nontemporal_load(keys + base) - kmin);
  d0 = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)d0) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(d0 >> 32)) << 32);
  int64_t w0i = d0 < range ? (int64_t)(d0 >> 6) : 0;
  uint64_t win = (uint64_t)(w0i + lane) * 64 < range ? bitmap[w0i + lane] : 0ull;
  T k[PM_ROWS / 2][2];
#pragma unroll
  for (int r = 0; r < PM_ROWS / 2; r++) {
    int64_t j = base + r * 2 * WAVE + 2 * lane;
    if (j + 1 < n) { struct alignas(2 * sizeof(T)) P { T a, b; }; P p = *(const P*)(keys + j); k[r][0] = p.a; k[r][1] = p.b; }
    else { k[r][0] = j < n ? keys[j] : (T)0; k[r][1] = 0; }
  }
#pragma unroll
  for (int r = 0; r < PM_ROWS / 2; r++) {
    int64_t j = base + r * 2 * WAVE + 2 * lane;
    uint64_t mw = (mask && j < n) ? mask[j >> 6] >> (j & 63) : 3ull;
    bool h[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
      uint64_t d = (uint64_t)((int64_t)k[r][e] - kmin);
      bool go = j + e < n && ((mw >> e) & 1) && valid_at(key_valid, j + e) && d < range;
      int64_t rel = (int64_t)(d >> 6) - w0i;
      uint64_t word = __shfl(win, (int)(rel & 63), 64);
      if (go && (rel < 0 || rel >= WAVE)) word = bitmap[d >> 6];
      h[e] = go && ((word >> (d & 63)) & 1ull);
    }
    uint64_t be = ballot64(h[0]), bo = ballot64(h[1]);
    uint64_t w0 = spread32(be) | (spread32(bo) << 1), w1 = spread32(be >> 32) | (spread32(bo >> 32) << 1);
    int64_t wbase = (base >> 6) + 2 * r;
    if (lane == 0) { if (base + r * 2 * WAVE < n) match_bits[wbase] = w0; if (base + r * 2 * WAVE + WAVE < n) match_bits[wbase + 1] = w1; }
  }
}
// ---- candidate v4 (v2 + compile-time mask / validity presence) of the product kernel: no chunk loop, one pass over r with the ballots inside (shape of MODE 4 above)
template <typename T, bool HAS_MASK, bool HAS_VALID>
__global__ void __launch_bounds__(BLOCK) k_probe_match_bitmap_v4(const T* keys, const uint64_t* key_valid, const uint64_t* mask, int64_t n, int64_t kmin, uint64_t range,
                                                                 const uint64_t* bitmap, uint64_t* match_bits) {
  int lane = lane_id();
  int64_t base = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * (WAVE * PM_ROWS);
  if (base >= n) return;
  uint64_t d0 = (uint64_t)((int64_t)keys[base] - kmin);
  d0 = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)d0) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(d0 >> 32)) << 32);
  int64_t w0i = d0 < range ? (int64_t)(d0 >> 6) : 0;
  uint64_t win = (uint64_t)(w0i + lane) * 64 < range ? bitmap[w0i + lane] : 0ull;
  T k[PM_ROWS / 2][2];
#pragma unroll
  for (int r = 0; r < PM_ROWS / 2; r++) {
    int64_t j = base + r * 2 * WAVE + 2 * lane;
    if (j + 1 < n) { struct alignas(2 * sizeof(T)) P { T a, b; }; P p = *(const P*)(keys + j); k[r][0] = p.a; k[r][1] = p.b; }
    else { k[r][0] = j < n ? keys[j] : (T)0; k[r][1] = 0; }
  }
#pragma unroll
  for (int r = 0; r < PM_ROWS / 2; r++) {
    int64_t j = base + r * 2 * WAVE + 2 * lane;
    uint64_t mw = (HAS_MASK && j < n) ? mask[j >> 6] >> (j & 63) : 3ull;
    bool h[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
      uint64_t d = (uint64_t)((int64_t)k[r][e] - kmin);
      bool go = j + e < n && ((mw >> e) & 1) && (!HAS_VALID || valid_at(key_valid, j + e)) && d < range;
      int64_t rel = (int64_t)(d >> 6) - w0i;
      uint64_t word = __shfl(win, (int)(rel & 63), 64);
      if (go && (rel < 0 || rel >= WAVE)) word = bitmap[d >> 6];
      h[e] = go && ((word >> (d & 63)) & 1ull);
    }
    uint64_t be = ballot64(h[0]), bo = ballot64(h[1]);
    uint64_t w0 = spread32(be) | (spread32(bo) << 1), w1 = spread32(be >> 32) | (spread32(bo >> 32) << 1);
    int64_t wbase = (base >> 6) + 2 * r;
    if (lane == 0) { if (base + r * 2 * WAVE < n) match_bits[wbase] = w0; if (base + r * 2 * WAVE + WAVE < n) match_bits[wbase + 1] = w1; }
  }
}
__global__ void init_keys(int64_t* k, int64_t n, int rnd) { int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; if (i < n) { uint64_t x = i; if (rnd) { x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; x %= 600000000ull; } else x = i; k[i] = (int64_t)x; } }
__global__ void mod_keys(int64_t* k, int64_t n, uint64_t range) { int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; if (i < n) k[i] = (int64_t)((uint64_t)k[i] % range); }
__global__ void init_words(uint64_t* w, int64_t nw, uint64_t mul) { int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; if (i < nw) { uint64_t x = (i + 1) * mul; x ^= x >> 31; w[i] = x; } }
template <int MODE, int ROWS> float run(const int64_t* keys, const uint64_t* mask, const uint64_t* bm, int64_t n, uint64_t* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  int64_t grid = (n + 256 * ROWS - 1) / (256 * ROWS);
  for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k<MODE, ROWS>), dim3(grid), dim3(256), 0, 0, keys, mask, bm, n, 600000000ull, out);
  hipEventRecord(a); for (int it = 0; it < 5; it++) hipLaunchKernelGGL((k<MODE, ROWS>), dim3(grid), dim3(256), 0, 0, keys, mask, bm, n, 600000000ull, out); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
  int64_t n = 600000000; int64_t nw = (n + 63) / 64;
  int64_t* keys; uint64_t *mask, *bm, *out;
  CK(hipMalloc(&keys, n * 8)); CK(hipMalloc(&mask, nw * 8)); CK(hipMalloc(&bm, nw * 8)); CK(hipMalloc(&out, nw * 8 + 64));
  hipLaunchKernelGGL(init_words, dim3((nw + 255) / 256), dim3(256), 0, 0, mask, nw, 0x9E3779B97F4A7C15ull);
  hipLaunchKernelGGL(init_words, dim3((nw + 255) / 256), dim3(256), 0, 0, bm, nw, 0xD1B54A32D192ED03ull);
  for (int rnd = 0; rnd < 2; rnd++) {
    hipLaunchKernelGGL(init_keys, dim3((n + 255) / 256), dim3(256), 0, 0, keys, n, rnd); CK(hipDeviceSynchronize());
    float a8 = run<0, 8>(keys, mask, bm, n, out), b8 = run<1, 8>(keys, mask, bm, n, out), c8 = run<2, 8>(keys, mask, bm, n, out);
    float w3 = run<3, 8>(keys, mask, bm, n, out), w4 = run<4, 8>(keys, mask, bm, n, out), w3_16 = run<3, 16>(keys, mask, bm, n, out), w4_16 = run<4, 16>(keys, mask, bm, n, out);
    printf("window variants: LDS %.3f ms, bpermute %.3f ms | rows/lane 16: LDS %.3f bpermute %.3f\n", w3, w4, w3_16, w4_16);
    float a16 = run<0, 16>(keys, mask, bm, n, out), c16 = run<2, 16>(keys, mask, bm, n, out), c4 = run<2, 4>(keys, mask, bm, n, out);
    printf("%s keys: A(keys only) %.3f ms = %.2f TB/s | B(+mask) %.3f | C(+bitmap) %.3f ms = %.2f TB/s | rows/lane 16: A %.3f C %.3f | rows/lane 4: C %.3f\n", rnd ? "random" : "sorted",
           a8, n * 8.0 / a8 / 1e9, b8, c8, n * 8.25 / c8 / 1e9, a16, c16, c4);
  }
  {
    hipLaunchKernelGGL(init_keys, dim3((n + 255) / 256), dim3(256), 0, 0, keys, n, 0); CK(hipDeviceSynchronize());
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float ms;
    int64_t grid = (n + 2047) / 2048;
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k_probe_match_bitmap<int64_t>), dim3(grid), dim3(256), 0, 0, keys, (const uint64_t*)nullptr, mask, n, (int64_t)0, (uint64_t)600000000ull, bm, out);
    hipEventRecord(a); for (int it = 0; it < 5; it++) hipLaunchKernelGGL((k_probe_match_bitmap<int64_t>), dim3(grid), dim3(256), 0, 0, keys, (const uint64_t*)nullptr, mask, n, (int64_t)0, (uint64_t)600000000ull, bm, out); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); printf("product kernel, sorted keys + mask: %.3f ms\n", ms / 5);
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k_probe_match_bitmap_v2<int64_t>), dim3(grid), dim3(256), 0, 0, keys, (const uint64_t*)nullptr, mask, n, (int64_t)0, (uint64_t)600000000ull, bm, out);
    hipEventRecord(a); for (int it = 0; it < 5; it++) hipLaunchKernelGGL((k_probe_match_bitmap_v2<int64_t>), dim3(grid), dim3(256), 0, 0, keys, (const uint64_t*)nullptr, mask, n, (int64_t)0, (uint64_t)600000000ull, bm, out); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); printf("candidate v2: %.3f ms\n", ms / 5);
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k_probe_match_bitmap_v4<int64_t, true, false>), dim3(grid), dim3(256), 0, 0, keys, (const uint64_t*)nullptr, mask, n, (int64_t)0, (uint64_t)600000000ull, bm, out);
    hipEventRecord(a); for (int it = 0; it < 5; it++) hipLaunchKernelGGL((k_probe_match_bitmap_v4<int64_t, true, false>), dim3(grid), dim3(256), 0, 0, keys, (const uint64_t*)nullptr, mask, n, (int64_t)0, (uint64_t)600000000ull, bm, out); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); printf("candidate v4: %.3f ms\n", ms / 5);
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k_probe_match_bitmap_v3<int64_t>), dim3(grid), dim3(256), 0, 0, keys, (const uint64_t*)nullptr, mask, n, (int64_t)0, (uint64_t)600000000ull, bm, out);
    hipEventRecord(a); for (int it = 0; it < 5; it++) hipLaunchKernelGGL((k_probe_match_bitmap_v3<int64_t>), dim3(grid), dim3(256), 0, 0, keys, (const uint64_t*)nullptr, mask, n, (int64_t)0, (uint64_t)600000000ull, bm, out); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); printf("candidate v3: %.3f ms\n", ms / 5);
  }
  // orders -> customer shape: 150M random keys over a 15M-bit (1.9 MB) bitmap
  {
    int64_t n2 = 150000000; uint64_t range2 = 15000000;
    hipLaunchKernelGGL(init_keys, dim3((n2 + 255) / 256), dim3(256), 0, 0, keys, n2, 1); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(mod_keys, dim3((n2 + 255) / 256), dim3(256), 0, 0, keys, n2, range2); CK(hipDeviceSynchronize());
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float ms;
    int64_t grid = (n2 + 2047) / 2048;
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k<2, 8, 0>), dim3(grid), dim3(256), 0, 0, keys, mask, bm, n2, range2, out);
    hipEventRecord(a); for (int it = 0; it < 5; it++) hipLaunchKernelGGL((k<2, 8, 0>), dim3(grid), dim3(256), 0, 0, keys, mask, bm, n2, range2, out); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); printf("small-domain random gather: plain loads %.3f ms", ms / 5);
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k<2, 8, 1>), dim3(grid), dim3(256), 0, 0, keys, mask, bm, n2, range2, out);
    hipEventRecord(a); for (int it = 0; it < 5; it++) hipLaunchKernelGGL((k<2, 8, 1>), dim3(grid), dim3(256), 0, 0, keys, mask, bm, n2, range2, out); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); printf(" | nontemporal key+mask loads %.3f ms", ms / 5);
    hipEventRecord(a); for (int it = 0; it < 5; it++) hipLaunchKernelGGL((k<1, 8, 0>), dim3(grid), dim3(256), 0, 0, keys, mask, bm, n2, range2, out); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); printf(" | no gather %.3f ms\n", ms / 5);
    hipEventRecord(a); for (int it = 0; it < 5; it++) hipLaunchKernelGGL((k<2, 8, 2>), dim3(grid), dim3(256), 0, 0, keys, mask, bm, n2, range2, out); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); printf("small-domain random gather, nontemporal 8-B bitmap loads %.3f ms\n", ms / 5);
  }
  return 0;
}
