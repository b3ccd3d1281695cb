// partition.hip -- a14: BatchPartitioner::partition_iter, Partitioning::Hash(exprs, n)
// (physical-plan/src/repartition/mod.rs:148-221).  destination = create_hashes(keys) % n (:185); the
// per-destination index lists keep input order (:196-214), which is exactly a stable multi-split:
// one (n <= 256) or two LSD radix passes over the destination id.  The caller gathers each column once
// with the grouped index list (`take`, :202) and slices it by the returned counts -- on N GPUs those
// slices are the send buffers of the RCCL all-to-all that replaces the in-process channels (:442-580).
#include "device_utils.h"

namespace dfgpu {

__global__ void __launch_bounds__(BLOCK) k_dest_of_rows(KeySet ks, int64_t n, uint32_t nparts, int force_zero, uint32_t* dest, unsigned long long* counts) {
  extern __shared__ unsigned long long lc[];
  for (uint32_t p = threadIdx.x; p < nparts; p += BLOCK) lc[p] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    bool an; uint64_t h = force_zero ? 0 : keyset_hash(ks, i, 0, &an);
    uint32_t d = (uint32_t)(h % nparts);
    dest[i] = d;
    atomicAdd(&lc[d], 1ull);
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < nparts; p += BLOCK) if (lc[p]) atomicAdd(&counts[p], lc[p]);
}

}  // namespace dfgpu

using namespace dfgpu;
extern "C" dfgpu_status dfgpu_hash_partition(dfgpu_ctx* ctx, const dfgpu_array* const* keys, int32_t nkeys, int32_t num_partitions,
                                             dfgpu_array** out_indices, int64_t* counts_host) {
  return guard(ctx, [&] {
    if (!keys || !out_indices || !counts_host) fail(DFGPU_INVALID_ARGUMENT, "hash_partition: null argument");
    if (num_partitions < 1 || num_partitions > 4096) fail(DFGPU_INVALID_ARGUMENT, "hash_partition: 1..4096 partitions supported, got %d", num_partitions);
    KeySet ks = make_keyset(keys, nkeys);
    int64_t n = keys[0]->length;
    ArrayHolder idx(new_fixed(ctx, DFGPU_UINT32, n));
    launch_iota_u32(ctx, (uint32_t*)idx.get()->values->ptr, n, 0);
    BufferPtr dest = alloc_buffer(ctx, (size_t)(n + 1) * 4), counts = alloc_buffer(ctx, (size_t)num_partitions * 8, true);
    if (n) {
      hipLaunchKernelGGL(k_dest_of_rows, dim3(grid_for(n, BLOCK * 8, ctx->num_cus * 8)), dim3(BLOCK), (size_t)num_partitions * 8, ctx->stream, ks, n, (uint32_t)num_partitions,
                         ctx->force_hash_collisions ? 1 : 0, (uint32_t*)dest->ptr, (unsigned long long*)counts->ptr);
      KERNEL_CHECK();
      int bits = 1; while ((1 << bits) < num_partitions) bits++;
      if (num_partitions > 1) radix_sort_pairs_u32(ctx, (uint32_t*)dest->ptr, (uint32_t*)idx.get()->values->ptr, n, bits);
    }
    HIP_CHECK(hipMemcpyAsync(counts_host, counts->ptr, (size_t)num_partitions * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *out_indices = idx.release();
  });
}
