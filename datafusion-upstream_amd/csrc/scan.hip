// scan.hip -- device-wide exclusive prefix sums (three-phase, deterministic, no inter-workgroup hand-off).
#include "device_utils.h"

namespace dfgpu {

constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;   // 2048 elements per workgroup

__global__ void __launch_bounds__(BLOCK) k_scan_block_sums(const uint32_t* in, int64_t n, uint64_t* block_sums) {
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) s += in[base + k];
  __shared__ uint64_t lds[4];
  uint64_t tot; (void)block_exclusive_sum<uint64_t>(s, lds, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}
// single workgroup: exclusive scan of block sums in place, total -> *d_total
__global__ void __launch_bounds__(BLOCK) k_scan_sums(uint64_t* sums, int64_t nb, uint64_t* d_total) {
  __shared__ uint64_t lds[4];
  uint64_t carry = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += BLOCK) {
    int64_t i = b0 + threadIdx.x;
    uint64_t v = i < nb ? sums[i] : 0, tot;
    uint64_t ex = block_exclusive_sum<uint64_t>(v, lds, &tot);
    if (i < nb) sums[i] = carry + ex;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && d_total) *d_total = carry;
}
template <typename OUT>
__global__ void __launch_bounds__(BLOCK) k_scan_write(const uint32_t* in, int64_t n, const uint64_t* block_sums, OUT* out) {
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS]; uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = base + k < n ? in[base + k] : 0; s += v[k]; }
  __shared__ uint64_t lds[4];
  uint64_t tot; uint64_t ex = block_exclusive_sum<uint64_t>(s, lds, &tot) + block_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { if (base + k < n) out[base + k] = (OUT)ex; ex += v[k]; }
}

// small inputs (block counts, radix histograms): one workgroup, one launch instead of three
template <typename OUT>
__global__ void __launch_bounds__(BLOCK) k_scan_small(const uint32_t* in, int64_t n, OUT* out, uint64_t* d_total) {
  __shared__ uint64_t lds[4];
  uint64_t carry = 0;
  for (int64_t t0 = 0; t0 < n; t0 += SCAN_TILE) {
    int64_t base = t0 + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS]; uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = base + k < n ? in[base + k] : 0; s += v[k]; }
    uint64_t tot; uint64_t ex = block_exclusive_sum<uint64_t>(s, lds, &tot) + carry;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { if (base + k < n) out[base + k] = (OUT)ex; ex += v[k]; }
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && d_total) *d_total = carry;
}
constexpr int64_t SCAN_SMALL_MAX = 8 * SCAN_TILE;

template <typename OUT>
static void scan_impl(dfgpu_ctx* ctx, const uint32_t* in, OUT* out, int64_t n, uint64_t* d_total) {
  if (n <= 0) { if (d_total) HIP_CHECK(hipMemsetAsync(d_total, 0, 8, ctx->stream)); return; }
  if (n <= SCAN_SMALL_MAX) { hipLaunchKernelGGL((k_scan_small<OUT>), dim3(1), dim3(BLOCK), 0, ctx->stream, in, n, out, d_total); KERNEL_CHECK(); return; }
  int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  BufferPtr sums = alloc_buffer(ctx, (size_t)nb * 8);
  hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(BLOCK), 0, ctx->stream, in, n, (uint64_t*)sums->ptr);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(BLOCK), 0, ctx->stream, (uint64_t*)sums->ptr, nb, d_total);
  hipLaunchKernelGGL((k_scan_write<OUT>), dim3((unsigned)nb), dim3(BLOCK), 0, ctx->stream, in, n, (const uint64_t*)sums->ptr, out);
  KERNEL_CHECK();
}
void exclusive_scan_u32(dfgpu_ctx* ctx, const uint32_t* in, uint64_t* out, int64_t n, uint64_t* d_total) { scan_impl<uint64_t>(ctx, in, out, n, d_total); }
// in-place is safe: every thread reads its 8 inputs into registers before any write of the same tile,
// and tiles are disjoint between workgroups.
void exclusive_scan_u32_inplace32(dfgpu_ctx* ctx, uint32_t* data, int64_t n, uint64_t* d_total) { scan_impl<uint32_t>(ctx, data, data, n, d_total); }

}  // namespace dfgpu
