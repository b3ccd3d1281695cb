// join.hip -- a2-a6: HashJoinExec build / probe / join-type index algebra on gfx950.
//
// Reference: datafusion/physical-plan/src/joins/hash_join.rs + joins/utils.rs (see include/dfgpu.h).
//
// Table layout (HBM): open addressing, linear probing, one 8-byte slot = (hash tag : 32 | representative
// build row : 32), capacity = next_pow2(2 * rows) so a probe touches one 64-B line in the common case.
// Instead of the reference's serial `next[]` chains (joins/utils.rs:203-229) the build is a parallel
// find-or-insert (64-bit CAS) that groups equal KEYS per slot, followed -- only when some key repeats --
// by a stable radix sort of (slot, row) into a CSR row list.  That makes the probe deterministic and
// reproduces the reference's emission order (probe order, then build input order; hash_join.rs:161-197)
// without pointer chasing; unique-key builds (every TPC-H PK join) skip the CSR entirely.
// The probe is count -> exclusive scan -> fill, so output order never depends on scheduling.
#include "device_utils.h"

namespace dfgpu {
constexpr uint64_t SLOT_EMPTY = ~0ull;
constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;
}
using namespace dfgpu;

struct dfgpu_join_table {
  dfgpu_ctx* ctx = nullptr;
  int64_t n_build = 0; int32_t nkeys = 0; bool null_equals_null = false;
  std::vector<dfgpu_array*> keys; KeySet ks{};
  uint64_t capacity = 0; int cap_bits = 0;
  BufferPtr slots;        // u64[capacity]
  BufferPtr slot_count;   // u32[capacity]   rows per key group
  BufferPtr slot_start;   // u32[capacity]   CSR start (non-unique only)
  BufferPtr csr_rows;     // u32[n_inserted] build rows ordered by (slot, row) (non-unique only)
  BufferPtr build_mask;   // effective opt_mask words or null
  BufferPtr visited;      // u64 words over n_build
  bool unique = true; int64_t n_inserted = 0, n_groups = 0;
  int64_t mem = 0;
  ~dfgpu_join_table() { for (auto* a : keys) dfgpu_array_release(a); }
};

namespace dfgpu {

__device__ inline bool row_selected(const uint64_t* mask, int64_t i) { return mask == nullptr || bit_get(mask, i); }

__global__ void __launch_bounds__(BLOCK) k_join_build(KeySet ks, int64_t n, const uint64_t* mask, int null_eq, int force_zero,
                                                      uint64_t* slots, uint32_t* slot_count, uint64_t cap_mask, uint32_t* row_slot,
                                                      unsigned long long* counters /* [0]=1 when some key repeats */) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t my_slot = NO_SLOT;
  if (row_selected(mask, i)) {
    bool any_null; uint64_t h = keyset_hash(ks, i, 0, &any_null);
    if (force_zero) h = 0;
    if (!any_null || null_eq) {     // a NULL key never matches (eq -> NULL, hash_join.rs:1067-1076) unless null_equals_null
      uint64_t tag = h >> 32, s = h & cap_mask, mine = (tag << 32) | (uint64_t)i;
      for (uint64_t step = 0; step <= cap_mask; step++) {
        uint64_t cur = slots[s];
        if (cur == SLOT_EMPTY) {
          cur = atomicCAS((unsigned long long*)&slots[s], (unsigned long long)SLOT_EMPTY, (unsigned long long)mine);
          if (cur == SLOT_EMPTY) cur = mine;
        }
        if ((cur >> 32) == tag) {
          int64_t rep = (int64_t)(cur & 0xFFFFFFFFull);
          if (rep == i || keyset_equal(ks, i, ks, rep, true)) { my_slot = (uint32_t)s; break; }
        }
        s = (s + 1) & cap_mask;
      }
      // no single-address counters here: 1e7 atomics on one word serialise (measured 10 ms per build at SF100)
      if (my_slot != NO_SLOT && atomicAdd(&slot_count[my_slot], 1u) != 0u) counters[0] = 1ull;
    }
  }
  row_slot[i] = my_slot;
}

__global__ void __launch_bounds__(BLOCK) k_fix_unslotted(uint32_t* row_slot, int64_t n, uint32_t sentinel) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n && row_slot[i] == NO_SLOT) row_slot[i] = sentinel;
}

// pass A: locate the key group of every probe row; per-workgroup match totals
__global__ void __launch_bounds__(BLOCK) k_join_probe_find(KeySet bks, KeySet pks, int64_t n, const uint64_t* mask, int null_eq, int force_zero,
                                                           const uint64_t* slots, const uint32_t* slot_count, uint64_t cap_mask, int unique,
                                                           uint32_t* match_slot, uint32_t* block_counts) {
  int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  uint32_t found = NO_SLOT, cnt = 0;
  if (j < n && row_selected(mask, j)) {
    bool any_null; uint64_t h = keyset_hash(pks, j, 0, &any_null);
    if (force_zero) h = 0;
    if (!any_null || null_eq) {
      uint64_t tag = h >> 32, s = h & cap_mask;
      for (uint64_t step = 0; step <= cap_mask; step++) {
        uint64_t cur = slots[s];
        if (cur == SLOT_EMPTY) break;
        if ((cur >> 32) == tag && keyset_equal(bks, (int64_t)(cur & 0xFFFFFFFFull), pks, j, null_eq != 0)) { found = (uint32_t)s; break; }
        s = (s + 1) & cap_mask;
      }
      if (found != NO_SLOT) cnt = unique ? 1u : slot_count[found];
    }
  }
  if (j < n) match_slot[j] = found;
  __shared__ uint32_t lds[4];
  uint32_t tot; (void)block_exclusive_sum<uint32_t>(cnt, lds, &tot);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = tot;
}
// pass B: emit (build row, probe row) pairs at scanned offsets; order = probe row, then build row ascending
__global__ void __launch_bounds__(BLOCK) k_join_probe_fill(int64_t n, const uint32_t* match_slot, const uint64_t* slots, const uint32_t* slot_count,
                                                           const uint32_t* slot_start, const uint32_t* csr_rows, int unique,
                                                           const uint64_t* block_offsets, uint64_t* out_build, uint32_t* out_probe) {
  int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  uint32_t s = j < n ? match_slot[j] : NO_SLOT;
  uint32_t cnt = s == NO_SLOT ? 0u : (unique ? 1u : slot_count[s]);
  __shared__ uint32_t lds[4];
  uint32_t tot; uint32_t ex = block_exclusive_sum<uint32_t>(cnt, lds, &tot);
  if (!cnt) return;
  uint64_t o = block_offsets[blockIdx.x] + ex;
  if (unique) { out_build[o] = slots[s] & 0xFFFFFFFFull; out_probe[o] = (uint32_t)j; }
  else { uint32_t st = slot_start[s]; for (uint32_t k = 0; k < cnt; k++) { out_build[o + k] = csr_rows[st + k]; out_probe[o + k] = (uint32_t)j; } }
}

__global__ void k_mark_bits_u64idx(const uint64_t* idx, const uint64_t* idx_valid, int64_t n, uint64_t* bits, int64_t nbits, uint32_t* flags) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !valid_at(idx_valid, i)) return;
  uint64_t r = idx[i];
  if (r >= (uint64_t)nbits) { atomicOr(flags, DFGPU_FLAG_OOB); return; }
  atomicOr((unsigned long long*)&bits[r >> 6], 1ull << (r & 63));
}
__global__ void k_mark_bits_range(const uint32_t* idx, int64_t n, int64_t lo, int64_t hi, uint64_t* bits) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t r = idx[i];
  if (r >= lo && r < hi) atomicOr((unsigned long long*)&bits[(r - lo) >> 6], 1ull << ((r - lo) & 63));
}
// out = (a ^ flip) & (b or all ones)
__global__ void k_combine_words(const uint64_t* a, uint64_t flip, const uint64_t* b, uint64_t* out, int64_t nw) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nw) out[i] = (a[i] ^ flip) & (b ? b[i] : ~0ull);
}
__global__ void k_u32_to_u64(const uint32_t* in, uint64_t* out, int64_t n, uint64_t add) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint64_t)in[i] + add;
}
__global__ void k_add_u32(uint32_t* v, int64_t n, uint32_t add) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] += add;
}

static void check_key_types(const dfgpu_join_table* t, const dfgpu_array* const* pk, int32_t nkeys) {
  if (nkeys != t->nkeys) fail(DFGPU_INVALID_ARGUMENT, "probe has %d key columns, build has %d", nkeys, t->nkeys);
  for (int c = 0; c < nkeys; c++)
    if (logical_type(pk[c]) != logical_type(t->keys[c])) fail(DFGPU_INVALID_ARGUMENT, "join key %d: build type %d vs probe type %d (the planner coerces first)", c, logical_type(t->keys[c]), logical_type(pk[c]));
}

}  // namespace dfgpu

extern "C" {

dfgpu_status dfgpu_join_build(dfgpu_ctx* ctx, const dfgpu_array* const* keys, int32_t nkeys, const dfgpu_array* opt_mask,
                              int32_t null_equals_null, dfgpu_join_table** out) {
  return guard(ctx, [&] {
    if (!keys || !out) fail(DFGPU_INVALID_ARGUMENT, "join_build: null argument");
    std::unique_ptr<dfgpu_join_table> t(new dfgpu_join_table());
    t->ctx = ctx; t->nkeys = nkeys; t->null_equals_null = null_equals_null != 0;
    t->ks = make_keyset(keys, nkeys);
    for (int c = 0; c < nkeys; c++) { t->keys.push_back(const_cast<dfgpu_array*>(keys[c])); dfgpu_array_retain(t->keys.back()); }
    int64_t n = keys[0]->length; t->n_build = n;
    t->build_mask = effective_mask(ctx, opt_mask, n);
    uint64_t cap = 64; int bits = 6; while (cap < (uint64_t)n * 2) { cap <<= 1; bits++; }
    if (cap > (1ull << 31)) fail(DFGPU_RESOURCES_EXHAUSTED, "build side of %lld rows exceeds the 2^30-row hash table limit", (long long)n);
    t->capacity = cap; t->cap_bits = bits;
    t->slots = alloc_buffer(ctx, cap * 8); HIP_CHECK(hipMemsetAsync(t->slots->ptr, 0xFF, cap * 8, ctx->stream));
    t->slot_count = alloc_buffer(ctx, cap * 4, true);
    t->visited = alloc_buffer(ctx, bitmap_bytes(n), true);
    BufferPtr row_slot = alloc_buffer(ctx, (size_t)(n + 1) * 4);
    zero_scratch(ctx);
    if (n) { KernelTimer kt_(ctx, "k_join_build"); hipLaunchKernelGGL(k_join_build, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, t->ks, n,
                              t->build_mask ? (const uint64_t*)t->build_mask->ptr : nullptr, null_equals_null ? 1 : 0, ctx->force_hash_collisions ? 1 : 0,
                              (uint64_t*)t->slots->ptr, (uint32_t*)t->slot_count->ptr, cap - 1, (uint32_t*)row_slot->ptr, (unsigned long long*)ctx->d_scratch64); }
    KERNEL_CHECK();
    t->unique = read_scratch(ctx, 0) == 0;
    t->mem = (int64_t)(cap * 12 + bitmap_bytes(n));
    if (!t->unique) {
      // CSR of build rows per key group: stable radix sort of (slot, row) then exclusive scan of group sizes
      hipLaunchKernelGGL(k_fix_unslotted, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (uint32_t*)row_slot->ptr, n, (uint32_t)cap);
      BufferPtr rows = alloc_buffer(ctx, (size_t)n * 4);
      launch_iota_u32(ctx, (uint32_t*)rows->ptr, n, 0);
      radix_sort_pairs_u32(ctx, (uint32_t*)row_slot->ptr, (uint32_t*)rows->ptr, n, bits + 1);
      t->csr_rows = rows;
      t->slot_start = alloc_buffer(ctx, cap * 4);
      HIP_CHECK(hipMemcpyAsync(t->slot_start->ptr, t->slot_count->ptr, cap * 4, hipMemcpyDeviceToDevice, ctx->stream));
      exclusive_scan_u32_inplace32(ctx, (uint32_t*)t->slot_start->ptr, (int64_t)cap, nullptr);
      t->mem += (int64_t)(cap * 4 + (size_t)n * 4);
    }
    *out = t.release();
  });
}
void dfgpu_join_table_free(dfgpu_join_table* t) { delete t; }
int64_t dfgpu_join_table_num_rows(const dfgpu_join_table* t) { return t ? t->n_build : 0; }
int64_t dfgpu_join_table_memory(const dfgpu_join_table* t) { return t ? t->mem : 0; }

dfgpu_status dfgpu_join_probe(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* const* probe_keys, int32_t nkeys,
                              const dfgpu_array* opt_mask, dfgpu_array** out_build_idx, dfgpu_array** out_probe_idx) {
  return guard(ctx, [&] {
    if (!t || !probe_keys || !out_build_idx || !out_probe_idx) fail(DFGPU_INVALID_ARGUMENT, "join_probe: null argument");
    check_key_types(t, probe_keys, nkeys);
    KeySet pks = make_keyset(probe_keys, nkeys);
    int64_t n = probe_keys[0]->length;
    BufferPtr mask = effective_mask(ctx, opt_mask, n);
    int64_t nb = (n + BLOCK - 1) / BLOCK;
    int64_t total = 0;
    BufferPtr match_slot = alloc_buffer(ctx, (size_t)(n + 1) * 4), bcounts = alloc_buffer(ctx, (size_t)(nb + 1) * 4), boffs = alloc_buffer(ctx, (size_t)(nb + 1) * 8);
    if (n) {
      { KernelTimer kt_(ctx, "k_join_probe_find");
      hipLaunchKernelGGL(k_join_probe_find, dim3((unsigned)nb), dim3(BLOCK), 0, ctx->stream, t->ks, pks, n, mask ? (const uint64_t*)mask->ptr : nullptr,
                         t->null_equals_null ? 1 : 0, ctx->force_hash_collisions ? 1 : 0, (const uint64_t*)t->slots->ptr, (const uint32_t*)t->slot_count->ptr,
                         t->capacity - 1, t->unique ? 1 : 0, (uint32_t*)match_slot->ptr, (uint32_t*)bcounts->ptr); }
      KERNEL_CHECK();
      exclusive_scan_u32(ctx, (const uint32_t*)bcounts->ptr, (uint64_t*)boffs->ptr, nb, ctx->d_scratch64 + 8);
      total = (int64_t)read_scratch(ctx, 8);
    }
    if (total > 0xFFFFFFF0ll) fail(DFGPU_RESOURCES_EXHAUSTED, "join output of %lld rows for one probe batch; split the probe batch", (long long)total);
    ArrayHolder ob(new_fixed(ctx, DFGPU_UINT64, total)), op(new_fixed(ctx, DFGPU_UINT32, total));
    if (total) { KernelTimer kt_(ctx, "k_join_probe_fill");
      hipLaunchKernelGGL(k_join_probe_fill, dim3((unsigned)nb), dim3(BLOCK), 0, ctx->stream, n, (const uint32_t*)match_slot->ptr, (const uint64_t*)t->slots->ptr,
                         (const uint32_t*)t->slot_count->ptr, t->slot_start ? (const uint32_t*)t->slot_start->ptr : nullptr,
                         t->csr_rows ? (const uint32_t*)t->csr_rows->ptr : nullptr, t->unique ? 1 : 0, (const uint64_t*)boffs->ptr,
                         (uint64_t*)ob.get()->values->ptr, (uint32_t*)op.get()->values->ptr); }
    KERNEL_CHECK();
    *out_build_idx = ob.release(); *out_probe_idx = op.release();
  });
}

dfgpu_status dfgpu_join_mark_visited(dfgpu_ctx* ctx, dfgpu_join_table* t, const dfgpu_array* build_idx) {
  return guard(ctx, [&] {
    if (!t || !build_idx || build_idx->type != DFGPU_UINT64) fail(DFGPU_INVALID_ARGUMENT, "mark_visited: UINT64 build indices expected");
    int64_t n = build_idx->length; if (!n) return;
    hipLaunchKernelGGL(k_mark_bits_u64idx, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)build_idx->values->ptr,
                       build_idx->validity ? (const uint64_t*)build_idx->validity->ptr : nullptr, n, (uint64_t*)t->visited->ptr, t->n_build, ctx->d_flags);
    KERNEL_CHECK();
    check_flags(ctx, "join_mark_visited");
  });
}

dfgpu_status dfgpu_join_final_indices(dfgpu_ctx* ctx, const dfgpu_join_table* t, int32_t join_type, dfgpu_array** out_build_idx) {
  return guard(ctx, [&] {
    if (!t || !out_build_idx) fail(DFGPU_INVALID_ARGUMENT, "final_indices: null argument");
    bool semi = join_type == DFGPU_JOIN_LEFT_SEMI;
    if (!semi && join_type != DFGPU_JOIN_LEFT && join_type != DFGPU_JOIN_FULL && join_type != DFGPU_JOIN_LEFT_ANTI)
      fail(DFGPU_INVALID_ARGUMENT, "join type %d produces no final build-side batch (need_produce_result_in_final)", join_type);
    int64_t n = t->n_build, nw = (n + 63) / 64;
    BufferPtr sel = alloc_buffer(ctx, bitmap_bytes(n), true);
    if (nw) hipLaunchKernelGGL(k_combine_words, dim3(grid_for(nw, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)t->visited->ptr, semi ? 0ull : ~0ull,
                               t->build_mask ? (const uint64_t*)t->build_mask->ptr : nullptr, (uint64_t*)sel->ptr, nw);
    KERNEL_CHECK();
    ArrayHolder idx32(mask_to_indices_impl(ctx, (const uint64_t*)sel->ptr, n));
    int64_t m = idx32.get()->length;
    ArrayHolder o(new_fixed(ctx, DFGPU_UINT64, m));
    if (m) hipLaunchKernelGGL(k_u32_to_u64, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)idx32.get()->values->ptr, (uint64_t*)o.get()->values->ptr, m, 0ull);
    KERNEL_CHECK();
    *out_build_idx = o.release();
  });
}

dfgpu_status dfgpu_join_adjust_indices(dfgpu_ctx* ctx, const dfgpu_array* build_idx, const dfgpu_array* probe_idx, int64_t range_start,
                                       int64_t range_end, int32_t join_type, dfgpu_array** out_build_idx, dfgpu_array** out_probe_idx) {
  return guard(ctx, [&] {
    if (!build_idx || !probe_idx || !out_build_idx || !out_probe_idx) fail(DFGPU_INVALID_ARGUMENT, "adjust_indices: null argument");
    if (build_idx->type != DFGPU_UINT64 || probe_idx->type != DFGPU_UINT32 || build_idx->length != probe_idx->length) fail(DFGPU_INVALID_ARGUMENT, "adjust_indices: (UINT64, UINT32) index arrays of equal length expected");
    int64_t m = probe_idx->length;
    switch (join_type) {
      case DFGPU_JOIN_INNER: case DFGPU_JOIN_LEFT:
        dfgpu_array_retain(const_cast<dfgpu_array*>(build_idx)); dfgpu_array_retain(const_cast<dfgpu_array*>(probe_idx));
        *out_build_idx = const_cast<dfgpu_array*>(build_idx); *out_probe_idx = const_cast<dfgpu_array*>(probe_idx); return;
      case DFGPU_JOIN_LEFT_SEMI: case DFGPU_JOIN_LEFT_ANTI:
        *out_build_idx = new_fixed(ctx, DFGPU_UINT64, 0); *out_probe_idx = new_fixed(ctx, DFGPU_UINT32, 0); return;
      default: break;
    }
    if (range_end < range_start) range_end = range_start;
    int64_t rl = range_end - range_start, nw = (rl + 63) / 64;
    BufferPtr bm = alloc_buffer(ctx, bitmap_bytes(rl), true);
    if (m && rl) hipLaunchKernelGGL(k_mark_bits_range, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)probe_idx->values->ptr, m, range_start, range_end, (uint64_t*)bm->ptr);
    bool want_set = join_type == DFGPU_JOIN_RIGHT_SEMI;
    if (!want_set && nw) hipLaunchKernelGGL(k_combine_words, dim3(grid_for(nw, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)bm->ptr, ~0ull, (const uint64_t*)nullptr, (uint64_t*)bm->ptr, nw);
    KERNEL_CHECK();
    ArrayHolder extra(mask_to_indices_impl(ctx, (const uint64_t*)bm->ptr, rl));   // tail bits beyond rl are masked by mask_word
    int64_t e = extra.get()->length;
    if (e && range_start) hipLaunchKernelGGL(k_add_u32, dim3(grid_for(e, BLOCK)), dim3(BLOCK), 0, ctx->stream, (uint32_t*)extra.get()->values->ptr, e, (uint32_t)range_start);
    KERNEL_CHECK();
    if (join_type == DFGPU_JOIN_RIGHT_SEMI || join_type == DFGPU_JOIN_RIGHT_ANTI) {
      // left indices are unused for right semi/anti (joins/utils.rs:1257-1269): emit an all-NULL build column of matching length
      dfgpu_array* nb = nullptr;
      dfgpu_status st = dfgpu_array_new_null(ctx, DFGPU_UINT64, 0, 0, e, &nb); if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str());
      *out_build_idx = nb; *out_probe_idx = extra.release(); return;
    }
    // Right / Full: matched pairs followed by the unmatched probe rows with NULL build index (append_right_indices :1284-1306)
    ArrayHolder ob(new_fixed(ctx, DFGPU_UINT64, m + e, 0, 0, e > 0)), op(new_fixed(ctx, DFGPU_UINT32, m + e));
    if (m) { HIP_CHECK(hipMemcpyAsync(ob.get()->values->ptr, build_idx->values->ptr, (size_t)m * 8, hipMemcpyDeviceToDevice, ctx->stream));
             HIP_CHECK(hipMemcpyAsync(op.get()->values->ptr, probe_idx->values->ptr, (size_t)m * 4, hipMemcpyDeviceToDevice, ctx->stream)); }
    if (e) {
      HIP_CHECK(hipMemsetAsync((uint64_t*)ob.get()->values->ptr + m, 0, (size_t)e * 8, ctx->stream));
      HIP_CHECK(hipMemcpyAsync((uint32_t*)op.get()->values->ptr + m, extra.get()->values->ptr, (size_t)e * 4, hipMemcpyDeviceToDevice, ctx->stream));
      // validity: first m bits set
      launch_set_bits_prefix(ctx, (uint64_t*)ob.get()->validity->ptr, m);
      ob.get()->null_count = e;
    }
    *out_build_idx = ob.release(); *out_probe_idx = op.release();
  });
}

}  // extern "C"

namespace dfgpu {
__global__ void k_set_bits_prefix(uint64_t* bits, int64_t m) {
  int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nw = (m + 63) >> 6;
  if (w >= nw) return;
  bits[w] = (w == nw - 1 && (m & 63)) ? ((1ull << (m & 63)) - 1ull) : ~0ull;
}
void launch_set_bits_prefix(dfgpu_ctx* ctx, uint64_t* bits, int64_t m) {
  if (m <= 0) return;
  hipLaunchKernelGGL(k_set_bits_prefix, dim3(grid_for((m + 63) / 64, BLOCK)), dim3(BLOCK), 0, ctx->stream, bits, m);
  KERNEL_CHECK();
}
}  // namespace dfgpu
