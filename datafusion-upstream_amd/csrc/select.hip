// select.hip -- arrow-select on device: take (gather), filter (ordered stream compaction), selection vectors.
//
// take:   ≙ arrow::compute::take as used by build_batch_from_indices (joins/utils.rs:1180-1230),
//         sort_batch (sorts/sort.rs:605) and BatchPartitioner (repartition/mod.rs:202).
// filter: ≙ filter_record_batch (filter.rs:315-327): wave64 ballot + popcount prefix, order preserving.
#include "device_utils.h"
#include <algorithm>

namespace dfgpu {

struct U128 { uint64_t lo, hi; };
template <int W> struct WT;
template <> struct WT<1> { using T = uint8_t; };
template <> struct WT<2> { using T = uint16_t; };
template <> struct WT<4> { using T = uint32_t; };
template <> struct WT<8> { using T = uint64_t; };
template <> struct WT<16> { using T = U128; };

__global__ void k_iota_u32(uint32_t* out, int64_t n, uint32_t start) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = start + (uint32_t)i;
}
void launch_iota_u32(dfgpu_ctx* ctx, uint32_t* out, int64_t n, uint32_t start) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_iota_u32, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, out, n, start);
  KERNEL_CHECK();
}

__device__ inline int64_t load_index(const void* idx, int w, int64_t i) { return w == 4 ? (int64_t)((const uint32_t*)idx)[i] : (int64_t)((const uint64_t*)idx)[i]; }

// One lane per output row; each wave owns 64 consecutive rows so the validity word is one ballot.
template <int W>
__global__ void __launch_bounds__(BLOCK) k_take_fixed(const typename WT<W>::T* src, const uint64_t* src_valid, int64_t src_len,
                                                      const void* idx, int idx_w, const uint64_t* idx_valid, int64_t n,
                                                      typename WT<W>::T* out, uint64_t* out_valid, uint32_t* flags) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  bool ok = false; typename WT<W>::T v{};
  if (i < n && valid_at(idx_valid, i)) {
    int64_t j = load_index(idx, idx_w, i);
    if (j < 0 || j >= src_len) atomicOr(flags, DFGPU_FLAG_OOB);
    else { ok = valid_at(src_valid, j); v = src[j]; }
  }
  if (i < n) out[i] = v;
  if (out_valid) { uint64_t m = ballot64(ok); if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) out_valid[i >> 6] = m; }
}
// No validity on either side (the shape of every gather behind a join of non-nullable columns): TAKE_ROWS rows per lane, their indices
// loaded together and unconditionally (row index clamped), then their values (source index clamped after the bounds check) -- the
// general kernel's load-behind-branch chain serialises the two dependent levels per row.
constexpr int TAKE_ROWS = 4;
template <int W, typename IT>
__global__ void __launch_bounds__(BLOCK) k_take_fixed_plain(const typename WT<W>::T* src, int64_t src_len, const IT* idx, int64_t n, typename WT<W>::T* out, uint32_t* flags) {
  const int64_t base = (int64_t)blockIdx.x * BLOCK * TAKE_ROWS + threadIdx.x;
  int64_t j[TAKE_ROWS]; typename WT<W>::T v[TAKE_ROWS]; bool oob = false;
#pragma unroll
  for (int q = 0; q < TAKE_ROWS; q++) { int64_t i = base + (int64_t)q * BLOCK; j[q] = (int64_t)idx[i < n ? i : n - 1]; }
#pragma unroll
  for (int q = 0; q < TAKE_ROWS; q++) { bool bad = j[q] < 0 || j[q] >= src_len; oob |= bad && (base + (int64_t)q * BLOCK < n); if (bad) j[q] = 0; }
#pragma unroll
  for (int q = 0; q < TAKE_ROWS; q++) v[q] = src[j[q]];
#pragma unroll
  for (int q = 0; q < TAKE_ROWS; q++) { int64_t i = base + (int64_t)q * BLOCK; if (i < n) out[i] = v[q]; }
  if (oob) atomicOr(flags, DFGPU_FLAG_OOB);
}
__global__ void __launch_bounds__(BLOCK) k_take_bool(const uint64_t* src, const uint64_t* src_valid, int64_t src_len,
                                                     const void* idx, int idx_w, const uint64_t* idx_valid, int64_t n,
                                                     uint64_t* out, uint64_t* out_valid, uint32_t* flags) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  bool ok = false, v = false;
  if (i < n && valid_at(idx_valid, i)) {
    int64_t j = load_index(idx, idx_w, i);
    if (j < 0 || j >= src_len) atomicOr(flags, DFGPU_FLAG_OOB);
    else { ok = valid_at(src_valid, j); v = bit_get(src, j); }
  }
  uint64_t mv = ballot64(v), mo = ballot64(ok);
  if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) { out[i >> 6] = mv; if (out_valid) out_valid[i >> 6] = mo; }
}
__global__ void __launch_bounds__(BLOCK) k_take_utf8_len(const int32_t* src_off, const uint64_t* src_valid, int64_t src_len,
                                                         const void* idx, int idx_w, const uint64_t* idx_valid, int64_t n,
                                                         uint32_t* out_len, uint64_t* out_valid, uint32_t* flags) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  bool ok = false; uint32_t len = 0;
  if (i < n && valid_at(idx_valid, i)) {
    int64_t j = load_index(idx, idx_w, i);
    if (j < 0 || j >= src_len) atomicOr(flags, DFGPU_FLAG_OOB);
    else { ok = valid_at(src_valid, j); if (ok) len = (uint32_t)(src_off[j + 1] - src_off[j]); }
  }
  if (i < n) out_len[i] = len;
  if (out_valid) { uint64_t m = ballot64(ok); if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) out_valid[i >> 6] = m; }
}
// One wave per 64 output rows: their bytes are one contiguous span of the output, which the lanes copy byte-interleaved (coalesced
// stores; every byte finds its row by a 6-step search over the wave's 64 row starts held in registers).  A lane copying its own row
// byte by byte -- the obvious kernel -- moved 0.17 TB/s on 20 M short strings.
__global__ void __launch_bounds__(BLOCK) k_take_utf8_copy(const uint8_t* src, const int32_t* src_off, const void* idx, int idx_w,
                                                          const uint64_t* out_off64, int64_t n, int32_t* out_off, uint8_t* out, uint64_t total) {
  const int lane = lane_id();
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;                      // rows 0 .. n (row n only carries the final offset)
  const uint64_t o = i < n ? out_off64[i] : total;
  const uint64_t next = i + 1 < n ? out_off64[i + 1] : total;
  if (i <= n) out_off[i] = (int32_t)o;
  int64_t s = 0;
  if (i < n && next > o) s = (int64_t)src_off[load_index(idx, idx_w, i)];             // rows without bytes (NULL, empty) never read their index
  const uint64_t wbeg = __shfl((unsigned long long)o, 0, 64), wend = __shfl((unsigned long long)next, 63, 64);
  const uint32_t rel = (uint32_t)(o - wbeg);                                          // 64 rows span < 4 GB (the whole output is < 2 GB)
  for (uint64_t pb = wbeg; pb < wend; pb += WAVE) {                                    // wave-uniform trip count: the row starts are read from every lane
    const uint64_t p = pb + (uint64_t)lane; const bool act = p < wend;
    const uint32_t pr = (uint32_t)((act ? p : wend - 1) - wbeg);
    int lo = 0;                                                                        // largest r with rel_r <= pr: among rows starting at p the last one, which is the one with bytes
#pragma unroll
    for (int step = 32; step > 0; step >>= 1) { int mid = lo + step; uint32_t om = (uint32_t)__shfl((int)rel, mid & 63, 64); if (mid < WAVE && om <= pr) lo = mid; }
    const uint32_t orow = (uint32_t)__shfl((int)rel, lo, 64);
    const int64_t srow = (int64_t)__shfl((long long)s, lo, 64);
    if (act) out[p] = src[srow + (int64_t)(pr - orow)];
  }
}

dfgpu_array* take_impl(dfgpu_ctx* ctx, const dfgpu_array* a, const void* idx, int idx_w, const uint64_t* idx_valid, int64_t n) {
  if (n > 0xFFFFFFF0ll) fail(DFGPU_NOT_IMPLEMENTED, "take above 2^32-16 rows");
  bool need_valid = idx_valid != nullptr || a->validity != nullptr;
  int32_t vt = a->type == DFGPU_DICTIONARY ? a->key_type : a->type;
  ArrayHolder h(new_array(ctx, a->type, n, a->precision, a->scale));
  dfgpu_array* o = h.get(); o->key_type = a->key_type;
  if (need_valid) o->validity = alloc_buffer(ctx, bitmap_bytes(n), n == 0); else o->null_count = 0;
  uint64_t* ov = need_valid ? (uint64_t*)o->validity->ptr : nullptr;
  const uint64_t* sv = a->validity ? (const uint64_t*)a->validity->ptr : nullptr;
  dim3 grid(grid_for(n, BLOCK)), block(BLOCK);
  if (a->type == DFGPU_UTF8) {
    BufferPtr lens = alloc_buffer(ctx, (size_t)(n + 1) * 4), off64 = alloc_buffer(ctx, (size_t)(n + 1) * 8);
    if (n) hipLaunchKernelGGL(k_take_utf8_len, grid, block, 0, ctx->stream, (const int32_t*)a->offsets->ptr, sv, a->length, idx, idx_w, idx_valid, n, (uint32_t*)lens->ptr, ov, ctx->d_flags);
    exclusive_scan_u32(ctx, (const uint32_t*)lens->ptr, (uint64_t*)off64->ptr, n, ctx->d_scratch64 + 61);
    uint64_t total = read_scratch(ctx, 61);
    if (total > 0x7FFFFFFFull) fail(DFGPU_EXECUTION, "Arrow error: offset overflow: Utf8 output of %llu bytes exceeds i32 offsets", (unsigned long long)total);
    o->values = alloc_buffer(ctx, (size_t)total); o->values_bytes = (int64_t)total; o->offsets = alloc_buffer(ctx, (size_t)(n + 1) * 4, true);
    hipLaunchKernelGGL(k_take_utf8_copy, dim3(grid_for(n + 1, BLOCK)), block, 0, ctx->stream, (const uint8_t*)a->values->ptr, (const int32_t*)a->offsets->ptr, idx, idx_w,
                       (const uint64_t*)off64->ptr, n, (int32_t*)o->offsets->ptr, (uint8_t*)o->values->ptr, total);
    KERNEL_CHECK();
  } else if (vt == DFGPU_BOOL) {
    o->values = alloc_buffer(ctx, bitmap_bytes(n), n == 0);
    if (n) hipLaunchKernelGGL(k_take_bool, grid, block, 0, ctx->stream, (const uint64_t*)a->values->ptr, sv, a->length, idx, idx_w, idx_valid, n, (uint64_t*)o->values->ptr, ov, ctx->d_flags);
    KERNEL_CHECK();
  } else {
    int w = type_width(vt);
    o->values = alloc_buffer(ctx, (size_t)n * w);
    KernelTimer kt_(ctx, "k_take_fixed");
    if (n && !need_valid && a->length > 0 && (idx_w == 4 || idx_w == 8)) {
      dim3 pgrid(grid_for(n, BLOCK * TAKE_ROWS));
#define TAKE_PLAIN(W) case W: if (idx_w == 4) hipLaunchKernelGGL((k_take_fixed_plain<W, uint32_t>), pgrid, block, 0, ctx->stream, (const WT<W>::T*)a->values->ptr, a->length, (const uint32_t*)idx, n, (WT<W>::T*)o->values->ptr, ctx->d_flags); \
                              else hipLaunchKernelGGL((k_take_fixed_plain<W, uint64_t>), pgrid, block, 0, ctx->stream, (const WT<W>::T*)a->values->ptr, a->length, (const uint64_t*)idx, n, (WT<W>::T*)o->values->ptr, ctx->d_flags); break;
      switch (w) { TAKE_PLAIN(1) TAKE_PLAIN(2) TAKE_PLAIN(4) TAKE_PLAIN(8) TAKE_PLAIN(16) default: fail(DFGPU_INTERNAL, "take: width %d", w); }
#undef TAKE_PLAIN
    } else if (n) switch (w) {
#define TAKE_CASE(W) case W: hipLaunchKernelGGL((k_take_fixed<W>), grid, block, 0, ctx->stream, (const WT<W>::T*)a->values->ptr, sv, a->length, idx, idx_w, idx_valid, n, (WT<W>::T*)o->values->ptr, ov, ctx->d_flags); break;
      TAKE_CASE(1) TAKE_CASE(2) TAKE_CASE(4) TAKE_CASE(8) TAKE_CASE(16)
#undef TAKE_CASE
      default: fail(DFGPU_INTERNAL, "take: width %d", w);
    }
    KERNEL_CHECK();
  }
  if (a->dictionary) { o->dictionary = a->dictionary; dfgpu_array_retain(a->dictionary); }
  if (need_valid) o->null_count = -1;
  return h.release();
}

// ---------------------------------------------------------------- mask -> ascending selection vector
constexpr int SEL_WORDS = BLOCK;        // 64-bit mask words per workgroup (one per thread) = 16384 rows
__device__ inline uint64_t mask_word(const uint64_t* bits, int64_t w, int64_t n) {
  int64_t nw = (n + 63) >> 6; if (w >= nw) return 0;
  uint64_t x = bits[w];
  if (w == nw - 1 && (n & 63)) x &= (1ull << (n & 63)) - 1ull;
  return x;
}
__global__ void __launch_bounds__(BLOCK) k_sel_count(const uint64_t* bits, int64_t n, uint32_t* counts) {
  uint32_t c = (uint32_t)__popcll(mask_word(bits, (int64_t)blockIdx.x * SEL_WORDS + threadIdx.x, n));
  __shared__ uint32_t lds[BLOCK / WAVE];
  uint32_t tot; (void)block_exclusive_sum<uint32_t>(c, lds, &tot);
  if (threadIdx.x == 0) counts[blockIdx.x] = tot;
}
// Each lane owns one mask word.  A wave whose 4096 rows hold few set bits lets every lane walk its own bits; a dense
// wave goes word by word with all 64 lanes testing one bit each so that the index stores coalesce.
__global__ void __launch_bounds__(BLOCK) k_sel_write(const uint64_t* bits, int64_t n, const uint32_t* offsets, uint32_t* out) {
  int64_t w = (int64_t)blockIdx.x * SEL_WORDS + threadIdx.x;
  int lane = lane_id();
  uint64_t word = mask_word(bits, w, n);
  uint32_t pc = (uint32_t)__popcll(word);
  __shared__ uint32_t lds[BLOCK / WAVE];
  uint32_t tot; uint32_t ex = block_exclusive_sum<uint32_t>(pc, lds, &tot) + offsets[blockIdx.x];
  uint32_t wave_tot = wave_sum(pc);
  if (wave_tot == 0) return;
  if (wave_tot <= 8 * WAVE) {
    uint32_t row0 = (uint32_t)(w * 64);
    while (word) { out[ex++] = row0 + (uint32_t)__builtin_ctzll(word); word &= word - 1; }
  } else {
    int64_t wave_w0 = w - lane;
    for (int k = 0; k < WAVE; k++) {
      uint64_t m = __shfl(word, k, 64); uint32_t p = __shfl(ex, k, 64);
      if (m == 0) continue;
      if ((m >> lane) & 1) out[p + __popcll(m & lanemask_lt())] = (uint32_t)((wave_w0 + k) * 64 + lane);
    }
  }
}
dfgpu_array* mask_to_indices_impl(dfgpu_ctx* ctx, const uint64_t* bits, int64_t n) { return mask_to_indices_checked(ctx, bits, n, -1, nullptr); }
// check_slot >= 0: scratch word `check_slot` (< 60; written by a kernel the caller has enqueued) comes back in the same read-back as the count; a non-zero value means the caller's
// precondition failed -- nothing is written and nullptr is returned (*check_value tells why).  One host round trip for "is the input what I hoped" + "how many".
dfgpu_array* mask_to_indices_checked(dfgpu_ctx* ctx, const uint64_t* bits, int64_t n, int check_slot, uint64_t* check_value) {
  int64_t nw = (n + 63) / 64, nb = (nw + SEL_WORDS - 1) / SEL_WORDS;
  if (n == 0) { if (check_slot >= 0) { uint64_t v = read_scratch(ctx, check_slot); if (check_value) *check_value = v; if (v) return nullptr; } return new_fixed(ctx, DFGPU_UINT32, 0); }
  BufferPtr counts = alloc_buffer(ctx, (size_t)nb * 4);
  KernelTimer kt_(ctx, "k_sel_count+scan+write");
  hipLaunchKernelGGL(k_sel_count, dim3((unsigned)nb), dim3(BLOCK), 0, ctx->stream, bits, n, (uint32_t*)counts->ptr);
  exclusive_scan_u32_inplace32(ctx, (uint32_t*)counts->ptr, nb, ctx->d_scratch64 + 60);
  int64_t total;
  if (check_slot >= 0 && check_slot < 60) {
    const uint64_t* w = read_scratch_range(ctx, check_slot, 61 - check_slot);
    if (check_value) *check_value = w[0];
    if (w[0]) return nullptr;
    total = (int64_t)w[60 - check_slot];
  } else total = (int64_t)read_scratch(ctx, 60);
  ArrayHolder h(new_fixed(ctx, DFGPU_UINT32, total));
  if (total) hipLaunchKernelGGL(k_sel_write, dim3((unsigned)nb), dim3(BLOCK), 0, ctx->stream, bits, n, (const uint32_t*)counts->ptr, (uint32_t*)h.get()->values->ptr);
  h.get()->identity = total == n;                        // every row selected: the selection vector is 0 .. n-1
  { OrderStats st; st.sorted = true; st.repeats = false; st.exact = false; st.lo = 0; st.hi = n - 1; order_stats_set(h.get(), st); }      // set bits in row order: strictly ascending
  KERNEL_CHECK();
  return h.release();
}


// The same vector for a consumer that only reads entries below the count it has on the device already (a rank): the array is n entries long, the first popcount(bits) are
// written, and nothing is read back.  Not an Arrow array to hand out -- its length is an upper bound.
dfgpu_array* mask_to_indices_uncounted(dfgpu_ctx* ctx, const uint64_t* bits, int64_t n, uint64_t* d_count) {
  int64_t nw = (n + 63) / 64, nb = (nw + SEL_WORDS - 1) / SEL_WORDS;
  ArrayHolder h(new_fixed(ctx, DFGPU_UINT32, n));
  if (n == 0) { if (d_count) HIP_CHECK(hipMemsetAsync(d_count, 0, 8, ctx->stream)); return h.release(); }
  BufferPtr counts = alloc_buffer(ctx, (size_t)nb * 4);
  KernelTimer kt_(ctx, "k_sel_count+scan+write");
  hipLaunchKernelGGL(k_sel_count, dim3((unsigned)nb), dim3(BLOCK), 0, ctx->stream, bits, n, (uint32_t*)counts->ptr);
  exclusive_scan_u32_inplace32(ctx, (uint32_t*)counts->ptr, nb, d_count);
  hipLaunchKernelGGL(k_sel_write, dim3((unsigned)nb), dim3(BLOCK), 0, ctx->stream, bits, n, (const uint32_t*)counts->ptr, (uint32_t*)h.get()->values->ptr);
  KERNEL_CHECK();
  return h.release();
}

// ---------------------------------------------------------------- several columns through ONE index array
// A gather of n random rows costs one 64-byte sector per row and column whatever the column's width (~53 G sectors/s on MI355X beyond
// L2).  Fixed-width columns without NULLs that go through the same indices are therefore interleaved into row-major records first (one
// streaming pass), gathered as records -- one sector per row for up to 64 bytes of columns -- and split back while they are written.
// Several fixed-width columns without NULLs through one index array in ONE launch, column by column (no record packing): for result-sized gathers, where a launch costs
// more than the bytes it moves (a query's last operators take 3-5 columns through each row list).
struct TakeCols { int32_t n; const void* src[8]; void* dst[8]; int32_t width[8]; };
template <typename IT>
__global__ void __launch_bounds__(BLOCK) k_take_multi_plain(TakeCols tc, const IT* idx, int64_t m, int64_t n_src, uint32_t* flags) {
  const int64_t base = (int64_t)blockIdx.x * BLOCK * TAKE_ROWS + threadIdx.x;
  int64_t j[TAKE_ROWS]; bool oob = false;
#pragma unroll
  for (int q = 0; q < TAKE_ROWS; q++) { const int64_t i = base + (int64_t)q * BLOCK; j[q] = (int64_t)idx[i < m ? i : m - 1]; const bool bad = j[q] < 0 || j[q] >= n_src; oob |= bad && i < m; if (bad) j[q] = 0; }
  for (int c = 0; c < 8; c++) {
    if (c >= tc.n) break;
#pragma unroll
    for (int q = 0; q < TAKE_ROWS; q++) {
      const int64_t i = base + (int64_t)q * BLOCK; if (i >= m) continue;
      switch (tc.width[c]) {
        case 1: ((uint8_t*)tc.dst[c])[i] = ((const uint8_t*)tc.src[c])[j[q]]; break;
        case 2: ((uint16_t*)tc.dst[c])[i] = ((const uint16_t*)tc.src[c])[j[q]]; break;
        case 4: ((uint32_t*)tc.dst[c])[i] = ((const uint32_t*)tc.src[c])[j[q]]; break;
        case 8: ((uint64_t*)tc.dst[c])[i] = ((const uint64_t*)tc.src[c])[j[q]]; break;
        default: ((uint4*)tc.dst[c])[i] = ((const uint4*)tc.src[c])[j[q]]; break;
      }
    }
  }
  if (oob) atomicOr(flags, DFGPU_FLAG_OOB);
}
struct RowCols { int32_t n; int32_t row_bytes; const void* src[16]; void* dst[16]; int32_t width[16]; int32_t off[16]; };
__global__ void __launch_bounds__(BLOCK) k_rows_pack(RowCols rc, int64_t n, uint8_t* rows) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint8_t* r = rows + i * rc.row_bytes;
  for (int c = 0; c < 16; c++) {
    if (c >= rc.n) break;
    switch (rc.width[c]) {
      case 1: r[rc.off[c]] = ((const uint8_t*)rc.src[c])[i]; break;
      case 2: *(uint16_t*)(r + rc.off[c]) = ((const uint16_t*)rc.src[c])[i]; break;
      case 4: *(uint32_t*)(r + rc.off[c]) = ((const uint32_t*)rc.src[c])[i]; break;
      case 8: *(uint64_t*)(r + rc.off[c]) = ((const uint64_t*)rc.src[c])[i]; break;
      default: { const uint64_t* p = (const uint64_t*)rc.src[c] + 2 * i; uint64_t* q = (uint64_t*)(r + rc.off[c]); q[0] = p[0]; q[1] = p[1]; break; }
    }
  }
}
constexpr int RG_ROWS = 4;      // records per lane in flight
// every lane first loads its whole records (16-byte loads, RG_ROWS records in flight), then scatters the fields to the output columns
template <typename IDX, int NQ>
__global__ void __launch_bounds__(BLOCK) k_rows_gather(RowCols rc, const uint8_t* rows, const IDX* idx, int64_t m, int64_t n_src, uint32_t* flags) {
  const int64_t base = (int64_t)blockIdx.x * BLOCK * RG_ROWS + threadIdx.x;
  __shared__ uint4 srec[RG_ROWS][NQ][BLOCK];          // a lane's records, read back field by field (field offsets are run-time values: LDS, not registers)
  uint4 rec[RG_ROWS][NQ];
#pragma unroll
  for (int q = 0; q < RG_ROWS; q++) {
    int64_t i = base + (int64_t)q * BLOCK; int64_t j = (int64_t)idx[i < m ? i : m - 1];
    if (j < 0 || j >= n_src) { atomicOr(flags, DFGPU_FLAG_OOB); j = 0; }
    const uint4* r = (const uint4*)(rows + j * rc.row_bytes);
#pragma unroll
    for (int v = 0; v < NQ; v++) rec[q][v] = r[v];
  }
#pragma unroll
  for (int q = 0; q < RG_ROWS; q++)
#pragma unroll
    for (int v = 0; v < NQ; v++) srec[q][v][threadIdx.x] = rec[q][v];
  for (int c = 0; c < 16; c++) {
    if (c >= rc.n) break;
    const int w = rc.width[c], o = rc.off[c], v = o >> 4, ob = o & 15;
#pragma unroll
    for (int q = 0; q < RG_ROWS; q++) {
      int64_t i = base + (int64_t)q * BLOCK; if (i >= m) continue;
      const uint8_t* r = (const uint8_t*)&srec[q][v][threadIdx.x] + ob;
      switch (w) {
        case 1: ((uint8_t*)rc.dst[c])[i] = *r; break; case 2: ((uint16_t*)rc.dst[c])[i] = *(const uint16_t*)r; break;
        case 4: ((uint32_t*)rc.dst[c])[i] = *(const uint32_t*)r; break; case 8: ((uint64_t*)rc.dst[c])[i] = *(const uint64_t*)r; break;
        default: { const uint64_t* p = (const uint64_t*)r; uint64_t* d = (uint64_t*)rc.dst[c] + 2 * i; d[0] = p[0]; d[1] = p[1]; break; }
      }
    }
  }
}
}  // namespace dfgpu

namespace dfgpu {
// take(sorted column, strictly ascending indices) is sorted (strictly, if the source is) and lies inside the source's bounds; take(ascending indices, ascending indices) is
// ascending.  The source's own statistics are measured here -- once, they stay with the array -- when it is a caller's plain integer column at most 16x the result (a base-table
// key column behind a filter or a join): the alternative is the same pass over every result that reaches a join build.  A column an operator computed is never measured here
// (it dies with the step; measuring it would be a pass and a read-back per step).
void order_stats_through_take(dfgpu_ctx* ctx, const dfgpu_array* values, const dfgpu_array* indices, dfgpu_array* out) {
  if (!out || out == values || out->validity || indices->validity || !out->length) return;
  auto is = order_stats_get(indices);
  if (!is || !is->sorted || is->repeats) return;
  auto vs = order_stats_get(values);
  if (!vs && values->base_column && values->length >= (1 << 16) && values->length <= out->length * 16) vs = order_stats_measure(ctx, values);      // a caller's column only: the memo outlives this step
  if (!vs || !vs->sorted) return;
  OrderStats st = *vs; st.exact = false; order_stats_set(out, st);
}
}  // namespace dfgpu

using namespace dfgpu;
extern "C" {

dfgpu_status dfgpu_take(dfgpu_ctx* ctx, const dfgpu_array* values, const dfgpu_array* indices, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!values || !indices || !out) fail(DFGPU_INVALID_ARGUMENT, "take: null argument");
    int w = indices->type == DFGPU_UINT32 || indices->type == DFGPU_INT32 ? 4 : (indices->type == DFGPU_UINT64 || indices->type == DFGPU_INT64 ? 8 : 0);
    if (!w) fail(DFGPU_INVALID_ARGUMENT, "take: indices must be 32/64-bit integers");
    if (indices->identity && indices->length == values->length) { dfgpu_array_retain(const_cast<dfgpu_array*>(values)); *out = const_cast<dfgpu_array*>(values); return; }      // arrays are immutable: share
    *out = take_impl(ctx, values, indices->values->ptr, w, indices->validity ? (const uint64_t*)indices->validity->ptr : nullptr, indices->length);
    order_stats_through_take(ctx, values, indices, *out);
    check_flags(ctx, "take");
  });
}
dfgpu_status dfgpu_take_multi(dfgpu_ctx* ctx, const dfgpu_array* const* values, int32_t n, const dfgpu_array* indices, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!values || !indices || !out || n < 0) fail(DFGPU_INVALID_ARGUMENT, "take_multi: null argument");
    const int iw = indices->type == DFGPU_UINT32 || indices->type == DFGPU_INT32 ? 4 : (indices->type == DFGPU_UINT64 || indices->type == DFGPU_INT64 ? 8 : 0);
    if (!iw) fail(DFGPU_INVALID_ARGUMENT, "take_multi: indices must be 32/64-bit integers");
    const int64_t m = indices->length;
    std::vector<ArrayHolder> res((size_t)n);
    // columns that can travel as records: fixed width, no NULLs, no dictionary, all of one length; worth it for many rows out of a large source
    std::vector<int> rec; int64_t n_src = -1;
    if (!indices->validity && !indices->identity && m >= (1 << 18))
      for (int32_t c = 0; c < n; c++) {
        const dfgpu_array* a = values[c];
        if (!a || a->type == DFGPU_DICTIONARY || a->type == DFGPU_UTF8 || a->type == DFGPU_BOOL || a->validity || !type_width(a->type)) continue;
        if (n_src < 0) n_src = a->length;
        if (a->length == n_src && rec.size() < 16) rec.push_back(c);
      }
    if (rec.size() >= 2 && n_src >= (1 << 20) && m * 4 >= n_src) {        // packing is a pass over the WHOLE source: only when the gather reads a good part of it (a result-sized gather out of a base table takes the plain multi-column kernel below)
      RowCols rc{}; int off = 0;
      std::sort(rec.begin(), rec.end(), [&](int x, int y) { return type_width(values[x]->type) > type_width(values[y]->type); });      // widest first: every field aligned to its width
      std::vector<int> use;
      for (int c : rec) { int w = type_width(values[c]->type); if (off + w > 64) continue; rc.src[rc.n] = values[c]->values->ptr; rc.width[rc.n] = w; rc.off[rc.n] = off; off += w; rc.n++; use.push_back(c); }
      rc.row_bytes = (off + 15) / 16 * 16;             // whole 16-byte words: the gather loads a record as 1..4 uint4
      if (use.size() >= 2) {
        BufferPtr rows = alloc_buffer(ctx, (size_t)n_src * rc.row_bytes);
        { KernelTimer kt_(ctx, "k_rows_pack");
          hipLaunchKernelGGL(k_rows_pack, dim3(grid_for(n_src, BLOCK)), dim3(BLOCK), 0, ctx->stream, rc, n_src, (uint8_t*)rows->ptr); KERNEL_CHECK(); }
        for (size_t u = 0; u < use.size(); u++) { const dfgpu_array* a = values[use[u]]; res[(size_t)use[u]].a = new_fixed(ctx, a->type, m, a->precision, a->scale); rc.dst[u] = res[(size_t)use[u]].get()->values->ptr; }
        { KernelTimer kt_(ctx, "k_rows_gather");
          const int nq = rc.row_bytes / 16; dim3 g(grid_for(m, BLOCK * RG_ROWS));
#define RG(IDX, NQ) hipLaunchKernelGGL((k_rows_gather<IDX, NQ>), g, dim3(BLOCK), 0, ctx->stream, rc, (const uint8_t*)rows->ptr, (const IDX*)indices->values->ptr, m, n_src, ctx->d_flags)
          if (iw == 4) { if (nq == 1) RG(uint32_t, 1); else if (nq == 2) RG(uint32_t, 2); else if (nq == 3) RG(uint32_t, 3); else RG(uint32_t, 4); }
          else { if (nq == 1) RG(uint64_t, 1); else if (nq == 2) RG(uint64_t, 2); else if (nq == 3) RG(uint64_t, 3); else RG(uint64_t, 4); }
#undef RG
          KERNEL_CHECK(); }
      }
    }
    // the columns no record path took, if they are plain (fixed width, no NULLs, one source length): one launch for up to 8 of them
    if (!indices->validity && !indices->identity && m > 0) {
      std::vector<int> plain; int64_t ns = -1;
      for (int32_t c = 0; c < n; c++) {
        const dfgpu_array* a = values[c];
        if (res[(size_t)c].a || !a || a->type == DFGPU_DICTIONARY || a->type == DFGPU_UTF8 || a->type == DFGPU_BOOL || a->validity || !type_width(a->type) || a->length == 0) continue;
        if (ns < 0) ns = a->length;
        if (a->length == ns) plain.push_back(c);
      }
      for (size_t p0 = 0; p0 < plain.size(); p0 += 8) {
        const size_t cnt = std::min<size_t>(8, plain.size() - p0); if (cnt < 2) break;          // a single left-over column takes the ordinary kernel below
        TakeCols tcs{}; tcs.n = (int32_t)cnt;
        for (size_t u = 0; u < cnt; u++) { const dfgpu_array* a = values[plain[p0 + u]]; res[(size_t)plain[p0 + u]].a = new_fixed(ctx, a->type, m, a->precision, a->scale);
          tcs.src[u] = a->values->ptr; tcs.dst[u] = res[(size_t)plain[p0 + u]].get()->values->ptr; tcs.width[u] = type_width(a->type); }
        KernelTimer kt_(ctx, "k_take_fixed");
        if (iw == 4) hipLaunchKernelGGL((k_take_multi_plain<uint32_t>), dim3(grid_for(m, BLOCK * TAKE_ROWS)), dim3(BLOCK), 0, ctx->stream, tcs, (const uint32_t*)indices->values->ptr, m, ns, ctx->d_flags);
        else hipLaunchKernelGGL((k_take_multi_plain<uint64_t>), dim3(grid_for(m, BLOCK * TAKE_ROWS)), dim3(BLOCK), 0, ctx->stream, tcs, (const uint64_t*)indices->values->ptr, m, ns, ctx->d_flags);
        KERNEL_CHECK();
      }
    }
    for (int32_t c = 0; c < n; c++) {
      if (res[(size_t)c].a || !values[c]) continue;
      if (indices->identity && indices->length == values[c]->length) { dfgpu_array_retain(const_cast<dfgpu_array*>(values[c])); res[(size_t)c].a = const_cast<dfgpu_array*>(values[c]); continue; }
      res[(size_t)c].a = take_impl(ctx, values[c], indices->values->ptr, iw, indices->validity ? (const uint64_t*)indices->validity->ptr : nullptr, m);
    }
    check_flags(ctx, "take");
    for (int32_t c = 0; c < n; c++) out[c] = res[(size_t)c].release();
  });
}
dfgpu_status dfgpu_mask_to_indices(dfgpu_ctx* ctx, const dfgpu_array* mask, dfgpu_array** out) {
  return guard(ctx, [&] {
    BufferPtr m = effective_mask(ctx, mask, mask ? mask->length : 0);
    if (!m) fail(DFGPU_INVALID_ARGUMENT, "mask_to_indices: null mask");
    *out = mask_to_indices_impl(ctx, (const uint64_t*)m->ptr, mask->length);
  });
}
dfgpu_status dfgpu_filter(dfgpu_ctx* ctx, const dfgpu_array* values, const dfgpu_array* mask, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!values || !mask) fail(DFGPU_INVALID_ARGUMENT, "filter: null argument");
    BufferPtr m = effective_mask(ctx, mask, values->length);
    ArrayHolder sel(mask_to_indices_impl(ctx, (const uint64_t*)m->ptr, mask->length));
    *out = take_impl(ctx, values, sel.get()->values->ptr, 4, nullptr, sel.get()->length);
  });
}

}  // extern "C"
