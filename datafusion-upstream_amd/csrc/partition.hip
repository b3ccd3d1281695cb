// partition.hip -- a14: BatchPartitioner::partition_iter, Partitioning::Hash(exprs, n)
// (physical-plan/src/repartition/mod.rs:148-221).  destination = create_hashes(keys) % n (:185); the
// per-destination index lists keep input order (:196-214), which is exactly a stable multi-split:
// one (n <= 256) or two LSD radix passes over the destination id.  The caller gathers each column once
// with the grouped index list (`take`, :202) and slices it by the returned counts -- on N GPUs those
// slices are the send buffers of the RCCL all-to-all that replaces the in-process channels (:442-580).
#include "device_utils.h"
#include "radix_partition.h"

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
    if (num_partitions <= 256 && n > 0 && n <= 0xFFFFFFF0ll) {        // one stable LDS-staged pass (radix_partition.h): the row numbers only
      ArrayHolder idx(new_fixed(ctx, DFGPU_UINT32, n));
      RpCols cols{}; cols.n = 0; cols.rowid_dst = (uint32_t*)idx.get()->values->ptr;
      RpResult r = rp_partition(ctx, RpHashKeySet{ ks, nullptr, ctx->force_hash_collisions ? 1 : 0 }, n, (uint32_t)num_partitions, cols, true, ctx->d_scratch64 + 9, "rp_hist", "rp_scan", "rp_scatter");
      std::vector<uint32_t> st((size_t)num_partitions + 1);
      ctx->count_sync("sync:partition_counts"); fetch_to_host(ctx, st.data(), r.starts->ptr, st.size() * 4);
      for (int32_t p2 = 0; p2 < num_partitions; p2++) counts_host[p2] = (int64_t)st[(size_t)p2 + 1] - (int64_t)st[(size_t)p2];
      *out_indices = idx.release();
      return;
    }
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
    fetch_to_host(ctx, counts_host, counts->ptr, (size_t)num_partitions * 8);
    *out_indices = idx.release();
  });
}

// ≙ BatchPartitioner::partition_iter with the `take` of every column folded in (repartition/mod.rs:196-214): ONE pass reads the key and
// payload columns and writes them grouped by destination (input order kept inside a destination).  out_cols[c] receives column c of all
// destinations back to back (the caller slices it by counts_host) for fixed-width columns without NULLs; for any other column (Utf8,
// dictionary, nullable) out_cols[c] stays NULL and the caller gathers it through out_indices, which is always produced.
// opt_mask: rows not selected are dropped (a fused FilterExec).
extern "C" dfgpu_status dfgpu_partition_columns(dfgpu_ctx* ctx, const dfgpu_array* const* keys, int32_t nkeys, int32_t num_partitions, const dfgpu_array* const* cols, int32_t ncols,
                                                const dfgpu_array* opt_mask, dfgpu_array** out_cols, dfgpu_array** out_indices, int64_t* counts_host) {
  return guard(ctx, [&] {
    if (!keys || !out_indices || !counts_host || (ncols && (!cols || !out_cols))) fail(DFGPU_INVALID_ARGUMENT, "partition_columns: null argument");
    if (num_partitions < 1 || num_partitions > 256) fail(DFGPU_NOT_IMPLEMENTED, "partition_columns: 1..256 partitions, got %d", num_partitions);
    KeySet ks = make_keyset(keys, nkeys);
    const int64_t n = keys[0]->length;
    if (n > 0xFFFFFFF0ll) fail(DFGPU_NOT_IMPLEMENTED, "partition_columns above 2^32-16 rows");
    BufferPtr mask = effective_mask(ctx, opt_mask, n);
    for (int32_t c = 0; c < ncols; c++) { out_cols[c] = nullptr; if (cols[c] && cols[c]->length != n) fail(DFGPU_INVALID_ARGUMENT, "partition_columns: column %d has %lld rows, keys have %lld", c, (long long)cols[c]->length, (long long)n); }
    ArrayHolder idx(new_fixed(ctx, DFGPU_UINT32, n));
    std::vector<ArrayHolder> outs((size_t)ncols);
    std::vector<int> direct;
    for (int32_t c = 0; c < ncols; c++) {
      const dfgpu_array* a = cols[c];
      if (!a || a->type == DFGPU_DICTIONARY || a->type == DFGPU_UTF8 || a->type == DFGPU_BOOL || a->validity || !type_width(a->type)) continue;
      direct.push_back(c);
      outs[(size_t)c].a = new_fixed(ctx, a->type, n, a->precision, a->scale);
    }
    RpResult r; std::vector<uint32_t> st((size_t)num_partitions + 1, 0);
    if (n) {
      // RP_MAX_COLS columns per pass; the row numbers ride with the first pass
      for (size_t c0 = 0; c0 < direct.size() || c0 == 0; c0 += RP_MAX_COLS) {
        RpCols rc{}; rc.rowid_dst = c0 == 0 ? (uint32_t*)idx.get()->values->ptr : nullptr;
        for (size_t j = c0; j < direct.size() && j < c0 + RP_MAX_COLS; j++) { const dfgpu_array* a = cols[direct[j]]; rc.c[rc.n++] = RpCol{ a->values->ptr, outs[(size_t)direct[j]].get()->values->ptr, type_width(a->type), RP_RAW, a->type }; }
        r = rp_partition(ctx, RpHashKeySet{ ks, mask ? (const uint64_t*)mask->ptr : nullptr, ctx->force_hash_collisions ? 1 : 0 }, n, (uint32_t)num_partitions, rc, true, ctx->d_scratch64 + 9, "rp_hist", "rp_scan", "rp_scatter");
        if (direct.size() <= c0 + RP_MAX_COLS) break;
      }
      ctx->count_sync("sync:partition_counts"); fetch_to_host(ctx, st.data(), r.starts->ptr, st.size() * 4);
    }
    const int64_t moved = st[(size_t)num_partitions];
    for (int32_t p2 = 0; p2 < num_partitions; p2++) counts_host[p2] = (int64_t)st[(size_t)p2 + 1] - (int64_t)st[(size_t)p2];
    auto trim = [&](ArrayHolder& h) {        // rows dropped by the selection leave the tail unused
      if (moved == n) return h.release();
      dfgpu_array* s2 = nullptr; dfgpu_status rc2 = dfgpu_array_slice(ctx, h.get(), 0, moved, &s2); if (rc2 != DFGPU_OK) fail(rc2, "%s", ctx->err.c_str());
      return s2;
    };
    for (int c : direct) out_cols[c] = trim(outs[(size_t)c]);
    *out_indices = trim(idx);
  });
}
