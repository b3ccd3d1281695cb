// hash.hip -- a1: create_hashes on device (datafusion/common/src/hash_utils.rs:357-417).
// One lane per row, coalesced loads of each key column, no intermediate per-column buffers:
// the k columns are combined in registers with combine_hashes (:38-41).
#include "device_utils.h"

namespace dfgpu {

__global__ void __launch_bounds__(BLOCK) k_hash_rows(KeySet ks, int64_t n, uint64_t seed, int force_zero, uint64_t* out) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  bool an; out[i] = force_zero ? 0 : keyset_hash(ks, i, seed, &an);
}
// single fixed-width, non-null, non-dictionary 8-byte key: the Q3/Q5/Q18 join key shape
__global__ void __launch_bounds__(BLOCK) k_hash_i64(const uint64_t* v, int64_t n, uint64_t seed, uint64_t* out) {
  int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * 2;
  if (i + 1 < n) { ulonglong2 x = *(const ulonglong2*)(v + i); ulonglong2 h; h.x = mix64(x.x ^ seed); h.y = mix64(x.y ^ seed); *(ulonglong2*)(out + i) = h; }
  else if (i < n) out[i] = mix64(v[i] ^ seed);
}

void hash_keys_device(dfgpu_ctx* ctx, const dfgpu_array* const* cols, int32_t k, uint64_t seed, uint64_t* out) {
  KeySet ks = make_keyset(cols, k);
  int64_t n = cols[0]->length;
  if (n == 0) return;
  const ColView& c = ks.c[0];
  if (k == 1 && !ctx->force_hash_collisions && !c.keys && !c.validity && (c.type == DFGPU_INT64 || c.type == DFGPU_UINT64 || c.type == DFGPU_FLOAT64) && ((uintptr_t)c.values & 15) == 0)
    hipLaunchKernelGGL(k_hash_i64, dim3(grid_for((n + 1) / 2, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)c.values, n, seed, out);
  else
    hipLaunchKernelGGL(k_hash_rows, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, ks, n, seed, ctx->force_hash_collisions ? 1 : 0, out);
  KERNEL_CHECK();
}

}  // namespace dfgpu

using namespace dfgpu;
extern "C" dfgpu_status dfgpu_hash_columns(dfgpu_ctx* ctx, const dfgpu_array* const* cols, int32_t k, uint64_t seed, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!cols || k < 1 || !out) fail(DFGPU_INVALID_ARGUMENT, "hash_columns: bad arguments");
    ArrayHolder h(new_fixed(ctx, DFGPU_UINT64, cols[0]->length));
    hash_keys_device(ctx, cols, k, seed, (uint64_t*)h.get()->values->ptr);
    *out = h.release();
  });
}
