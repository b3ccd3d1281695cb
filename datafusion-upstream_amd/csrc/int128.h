// int128.h -- 128-bit helpers for Decimal128 on gfx950.  clang lowers i128 add/sub/mul inline on AMDGPU but
// division would call compiler-rt (__divti3), which does not exist on device, so division is done here.
#pragma once
#include "device_utils.h"

namespace dfgpu {

__host__ __device__ inline i128 pow10_i128(int k) { i128 r = 1; for (int i = 0; i < k; i++) r *= 10; return r; }

__host__ __device__ inline double pow10_f64(int k) { double r = 1.0; for (int i = 0; i < k; i++) r *= 10.0; return r; }

__device__ inline int clz128(u128 x) { uint64_t hi = (uint64_t)(x >> 64), lo = (uint64_t)x; return hi ? __clzll(hi) : 64 + (lo ? __clzll(lo) : 64); }
// unsigned 128 / 128 -> quotient, remainder (d != 0)
__device__ inline u128 udivmod128(u128 n, u128 d, u128* rem) {
  if (d > n) { *rem = n; return 0; }
  if ((d >> 64) == 0 && (n >> 64) == 0) { uint64_t q = (uint64_t)n / (uint64_t)d; *rem = (uint64_t)n % (uint64_t)d; return q; }
  int shift = clz128(d) - clz128(n);
  u128 q = 0; d <<= shift;
  for (int i = 0; i <= shift; i++) { q <<= 1; if (n >= d) { n -= d; q |= 1; } d >>= 1; }
  *rem = n; return q;
}
// truncating signed division (Rust div_wrapping / C semantics); d != 0
__device__ inline i128 sdiv128(i128 a, i128 b, i128* rem) {
  bool na = a < 0, nb = b < 0;
  u128 ua = na ? (u128)0 - (u128)a : (u128)a, ub = nb ? (u128)0 - (u128)b : (u128)b, ur;
  u128 uq = udivmod128(ua, ub, &ur);
  if (rem) *rem = na ? (i128)((u128)0 - ur) : (i128)ur;
  return (na != nb) ? (i128)((u128)0 - uq) : (i128)uq;
}
__device__ inline bool add128_checked(i128 a, i128 b, i128* out) {
  i128 r = (i128)((u128)a + (u128)b);
  if ((a >= 0) == (b >= 0) && (r >= 0) != (a >= 0)) return false;
  *out = r; return true;
}
__device__ inline bool sub128_checked(i128 a, i128 b, i128* out) {
  i128 r = (i128)((u128)a - (u128)b);
  if ((a >= 0) != (b >= 0) && (r >= 0) != (a >= 0)) return false;
  *out = r; return true;
}
__device__ inline bool mul128_checked(i128 a, i128 b, i128* out) {
  const long long al = (long long)a, bl = (long long)b;
  if ((i128)al == a && (i128)bl == b) { *out = (i128)al * (i128)bl; return true; }      // both operands fit 64 bits: the product fits 128
  bool neg = (a < 0) != (b < 0);
  u128 ua = a < 0 ? (u128)0 - (u128)a : (u128)a, ub = b < 0 ? (u128)0 - (u128)b : (u128)b;
  uint64_t a0 = (uint64_t)ua, a1 = (uint64_t)(ua >> 64), b0 = (uint64_t)ub, b1 = (uint64_t)(ub >> 64);
  if (a1 && b1) return false;
  u128 lo = (u128)a0 * (u128)b0;
  u128 cross = a1 ? (u128)a1 * (u128)b0 : (u128)b1 * (u128)a0;
  if (cross >> 64) return false;
  u128 r = lo + (cross << 64);
  if (r < lo) return false;
  if (neg) { if (r > ((u128)1 << 127)) return false; *out = (i128)((u128)0 - r); }
  else { if (r >> 127) return false; *out = (i128)r; }
  return true;
}
__device__ inline bool decimal_fits(i128 v, int precision) { i128 lim = pow10_i128(precision); return v > -lim && v < lim; }

__device__ inline i128 load_i128(const void* p, int64_t i) { const uint64_t* q = (const uint64_t*)p + 2 * i; return (i128)(((u128)q[1] << 64) | q[0]); }
__device__ inline void store_i128(void* p, int64_t i, i128 v) { uint64_t* q = (uint64_t*)p + 2 * i; q[0] = (uint64_t)(u128)v; q[1] = (uint64_t)((u128)v >> 64); }

}  // namespace dfgpu
