// core.hip -- context, HBM-resident Arrow arrays, import/export, slice, concat.
#include "device_utils.h"
#include <mutex>

namespace dfgpu {

void fail(dfgpu_status code, const char* fmt, ...) {
  char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  throw Error(code, buf);
}

// ctx lifetime: the handle given to the user holds one reference, every owned Buffer another, so
// arrays may outlive dfgpu_ctx_destroy (Rust drops in arbitrary order).
struct CtxRefs { std::atomic<int64_t> n{1}; };
static std::mutex g_mu;
static std::vector<std::pair<dfgpu_ctx*, CtxRefs*>> g_ctx;
static CtxRefs* refs_of(dfgpu_ctx* c) { std::lock_guard<std::mutex> l(g_mu); for (auto& p : g_ctx) if (p.first == c) return p.second; return nullptr; }
static void ctx_unref(dfgpu_ctx* c) {
  CtxRefs* r = refs_of(c);
  if (!r || r->n.fetch_sub(1) != 1) return;
  { std::lock_guard<std::mutex> l(g_mu); for (size_t i = 0; i < g_ctx.size(); i++) if (g_ctx[i].first == c) { g_ctx.erase(g_ctx.begin() + i); break; } }
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->free_blocks) { for (auto& b : *c->free_blocks) (void)hipFree(b.second); delete c->free_blocks; }
  delete c->alloc_mu;
  if (c->d_flags) (void)hipFree(c->d_flags);
  if (c->d_scratch64) (void)hipFree(c->d_scratch64);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  for (auto& sp : c->spans) if (sp.start) { (void)hipEventDestroy(sp.start); (void)hipEventDestroy(sp.stop); }      // spans nobody resolved
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete r; delete c;
}

static size_t size_class(size_t bytes) {
  size_t n = 256;
  while (n < bytes) { size_t m = n + n / 2; if (m >= bytes && n >= 512) return m; n <<= 1; }
  return n;
}
Buffer::~Buffer() {
  if (owned && ptr && ctx) {
    // stream-ordered reuse: every consumer of this block was enqueued on ctx->stream before the next owner's work
    { std::lock_guard<std::mutex> l(*ctx->alloc_mu); ctx->free_blocks->emplace_back(bytes, ptr); ctx->cached_bytes += bytes; ctx->live_bytes -= bytes; }
    ctx_unref(ctx);
  }
}
BufferPtr alloc_buffer(dfgpu_ctx* ctx, size_t bytes, bool zero) {
  auto b = std::make_shared<Buffer>();
  size_t n = size_class(bytes ? bytes : 1);      // >= 256 B granules keep every buffer 16 B aligned + padded
  void* p = nullptr;
  {
    std::lock_guard<std::mutex> l(*ctx->alloc_mu);
    auto& fb = *ctx->free_blocks;
    for (size_t i = fb.size(); i-- > 0;) if (fb[i].first == n) { p = fb[i].second; fb[i] = fb.back(); fb.pop_back(); ctx->cached_bytes -= n; break; }
  }
  if (ctx->memory_limit > 0 && (int64_t)(ctx->live_bytes + n) > ctx->memory_limit) {
    if (p) { std::lock_guard<std::mutex> l(*ctx->alloc_mu); ctx->free_blocks->emplace_back(n, p); ctx->cached_bytes += n; }
    // the message of MemoryPool::try_grow's error (execution/src/memory_pool/pool.rs:243-249)
    fail(DFGPU_RESOURCES_EXHAUSTED, "Resources exhausted: Failed to allocate additional %zu bytes for dfgpu with %zu bytes already allocated - maximum available is %lld", n, ctx->live_bytes, (long long)ctx->memory_limit);
  }
  if (!p) {
    hipError_t e = hipMalloc(&p, n);
    if (e == hipErrorOutOfMemory) {          // give cached blocks back to the driver and retry once
      (void)hipGetLastError();
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
      { std::lock_guard<std::mutex> l(*ctx->alloc_mu); for (auto& x : *ctx->free_blocks) (void)hipFree(x.second); ctx->free_blocks->clear(); ctx->cached_bytes = 0; }
      e = hipMalloc(&p, n);
    }
    if (e != hipSuccess) fail(e == hipErrorOutOfMemory ? DFGPU_RESOURCES_EXHAUSTED : DFGPU_INTERNAL, "hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e));
  }
  { std::lock_guard<std::mutex> l(*ctx->alloc_mu); ctx->live_bytes += n; }
  b->ptr = p; b->bytes = n; b->ctx = ctx; b->owned = true;
  CtxRefs* r = refs_of(ctx); if (r) r->n.fetch_add(1);
  if (zero) HIP_CHECK(hipMemsetAsync(p, 0, n, ctx->stream));
  return b;
}
BufferPtr borrow_buffer(const void* ptr, size_t bytes) {
  auto b = std::make_shared<Buffer>(); b->ptr = const_cast<void*>(ptr); b->bytes = bytes; b->owned = false; return b;
}

dfgpu_array* new_array(dfgpu_ctx* ctx, int32_t type, int64_t length, int32_t precision, int32_t scale) {
  auto* a = new dfgpu_array(); a->ctx = ctx; a->type = type; a->length = length; a->precision = precision; a->scale = scale; return a;
}
dfgpu_array* new_fixed(dfgpu_ctx* ctx, int32_t type, int64_t length, int32_t precision, int32_t scale, bool with_validity) {
  ArrayHolder h(new_array(ctx, type, length, precision, scale));
  size_t vb = type == DFGPU_BOOL ? bitmap_bytes(length) : (size_t)length * type_width(type);
  h.get()->values = alloc_buffer(ctx, vb, type == DFGPU_BOOL);
  if (with_validity) h.get()->validity = alloc_buffer(ctx, bitmap_bytes(length), true); else h.get()->null_count = 0;
  return h.release();
}

void flush_flags(dfgpu_ctx* ctx) {
  if (!ctx->flags_pending) return;
  ctx->flags_pending = false;
  int saved = ctx->defer_flag_checks; ctx->defer_flag_checks = 0;
  std::string what = ctx->flags_what;
  struct Restore { dfgpu_ctx* c; int v; ~Restore() { c->defer_flag_checks = v; } } r{ctx, saved};
  check_flags(ctx, what.c_str());
}
void check_flags(dfgpu_ctx* ctx, const char* what) {
  if (ctx->defer_flag_checks > 0) { if (!ctx->flags_pending) ctx->flags_what = what; else if (ctx->flags_what.find(what) == std::string::npos) ctx->flags_what += std::string(", ") + what; ctx->flags_pending = true; return; }
  uint32_t f = 0;
  ctx->count_sync((std::string("sync:flags:") + what).c_str());
  fetch_to_pinned(ctx, 63, ctx->d_flags, 4);
  f = *(uint32_t*)(ctx->h_pinned + 63);
  if (!f) return;
  HIP_CHECK(hipMemsetAsync(ctx->d_flags, 0, 4, ctx->stream));
  if (f & DFGPU_FLAG_DIV_ZERO) fail(DFGPU_EXECUTION, "Arrow error: Divide by zero error (%s)", what);
  if (f & DFGPU_FLAG_OVERFLOW) fail(DFGPU_EXECUTION, "Arrow error: Arithmetic overflow (%s)", what);
  if (f & DFGPU_FLAG_CAST) fail(DFGPU_EXECUTION, "Arrow error: Cast error: value out of range (%s)", what);
  if (f & DFGPU_FLAG_OOB) fail(DFGPU_EXECUTION, "Arrow error: index out of bounds (%s)", what);
  if (f & DFGPU_FLAG_STALLED) fail(DFGPU_INTERNAL, "a workgroup gave up waiting for the tile counts of the workgroups in front of it; the result of that sort is not valid (%s; option sort_onesweep_rows=0 selects the three-launch passes)", what);
  fail(DFGPU_INTERNAL, "kernel raised flag %u (%s)", f, what);
}
__global__ void __launch_bounds__(512) k_post_words(const uint32_t* src, int nwords, uint32_t* h_dst, unsigned long long* h_seq, unsigned long long seq) {
  const int i = threadIdx.x;
  if (i < nwords) __hip_atomic_store(h_dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __syncthreads();
  if (i == 0) __hip_atomic_store(h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void fetch_to_pinned(dfgpu_ctx* ctx, int h_word, const void* d_src, size_t bytes) {
  if (bytes == 0 || (bytes & 3) || h_word < 0 || (size_t)h_word * 8 + bytes > (size_t)dfgpu_ctx::MAIL_WORDS * 8) fail(DFGPU_INTERNAL, "fetch_to_pinned: %zu bytes at word %d", bytes, h_word);
  if (!ctx->mailbox_readback) {
    HIP_CHECK(hipMemcpyAsync(ctx->h_pinned + h_word, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return;
  }
  const unsigned long long seq = ++ctx->mail_seq;
  unsigned long long* h_seq = (unsigned long long*)(ctx->h_pinned + dfgpu_ctx::MAIL_SEQ);
  hipLaunchKernelGGL(k_post_words, dim3(1), dim3(bytes <= 512 ? 128 : 512), 0, ctx->stream, (const uint32_t*)d_src, (int)(bytes / 4), (uint32_t*)(ctx->h_pinned + h_word), h_seq, seq);
  KERNEL_CHECK();
  // poll; every few thousand spins ask the stream: an error there must not leave the host spinning, and a drained stream means the words are in memory
  bool asked = false;
  for (uint64_t spins = 1; __atomic_load_n(h_seq, __ATOMIC_ACQUIRE) < seq; spins++) {
    if ((spins & 0xFFF) == 0) {
      hipError_t q = hipStreamQuery(ctx->stream); asked = true;
      if (q == hipSuccess) { HIP_CHECK(hipStreamSynchronize(ctx->stream)); break; }
      if (q != hipErrorNotReady) HIP_CHECK(q);
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  if (asked) (void)hipGetLastError();          // "not ready" is an answer, not an error: it must not be what the next KERNEL_CHECK finds
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}
void fetch_to_host(dfgpu_ctx* ctx, void* dst, const void* d_src, size_t bytes) {
  if (!bytes) return;
  if (ctx->mailbox_readback && bytes <= (size_t)dfgpu_ctx::MAIL_WORDS * 8 && !(bytes & 3)) { fetch_to_pinned(ctx, 0, d_src, bytes); memcpy(dst, ctx->h_pinned, bytes); return; }
  HIP_CHECK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
}
uint64_t read_scratch(dfgpu_ctx* ctx, int slot) {
  ctx->count_sync((std::string("sync:count") + std::to_string(slot)).c_str());
  fetch_to_pinned(ctx, slot, ctx->d_scratch64 + slot, 8);
  return ctx->h_pinned[slot];
}
// several consecutive slots with ONE read-back; returns a pointer to the pinned copies (valid until the next read-back)
const uint64_t* read_scratch_range(dfgpu_ctx* ctx, int first, int count) {
  ctx->count_sync((std::string("sync:count") + std::to_string(first) + ".." + std::to_string(first + count - 1)).c_str());
  fetch_to_pinned(ctx, first, ctx->d_scratch64 + first, (size_t)count * 8);
  return ctx->h_pinned + first;
}
static std::mutex g_stats_mu;
std::shared_ptr<const OrderStats> order_stats_get(const dfgpu_array* a) { std::lock_guard<std::mutex> l(g_stats_mu); return a->order_stats; }
void order_stats_set(const dfgpu_array* a, const OrderStats& st) { auto p = std::make_shared<const OrderStats>(st); std::lock_guard<std::mutex> l(g_stats_mu); const_cast<dfgpu_array*>(a)->order_stats = p; }
void zero_scratch(dfgpu_ctx* ctx) { HIP_CHECK(hipMemsetAsync(ctx->d_scratch64, 0, 64 * 8, ctx->stream)); }

ColView make_view(const dfgpu_array* a) {
  if (a->deferred_ids) materialize_ids(a->ctx, a);     // deferred group ids handed to a kernel that reads them as a column
  ColView v{};
  const dfgpu_array* d = a;
  if (a->type == DFGPU_DICTIONARY) {
    d = a->dictionary;
    v.keys = a->values->ptr; v.key_validity = a->validity ? (const uint64_t*)a->validity->ptr : nullptr; v.key_type = a->key_type;
  }
  v.type = d->type; v.width = type_width(d->type);
  v.values = d->values ? d->values->ptr : nullptr;
  v.validity = d->validity ? (const uint64_t*)d->validity->ptr : nullptr;
  v.offsets = d->offsets ? (const int32_t*)d->offsets->ptr : nullptr;
  v.precision = d->precision; v.scale = d->scale;
  return v;
}
KeySet make_keyset(const dfgpu_array* const* cols, int32_t n) {
  if (n < 1 || n > MAX_KEYS) fail(DFGPU_NOT_IMPLEMENTED, "between 1 and %d key columns are supported, got %d", MAX_KEYS, n);
  KeySet ks{}; ks.n = n;
  for (int i = 0; i < n; i++) {
    if (!cols[i]) fail(DFGPU_INVALID_ARGUMENT, "null key column");
    if (cols[i]->length != cols[0]->length) fail(DFGPU_INVALID_ARGUMENT, "key columns differ in length");
    ks.c[i] = make_view(cols[i]);
  }
  return ks;
}

__global__ void k_and_words(const uint64_t* a, const uint64_t* b, uint64_t* out, int64_t nw) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nw) out[i] = a[i] & b[i];
}
BufferPtr effective_mask(dfgpu_ctx* ctx, const dfgpu_array* mask, int64_t expect_len) {
  if (!mask) return nullptr;
  if (mask->type != DFGPU_BOOL) fail(DFGPU_INVALID_ARGUMENT, "mask must be a Boolean array");
  if (mask->length != expect_len) fail(DFGPU_INVALID_ARGUMENT, "mask length %lld != %lld", (long long)mask->length, (long long)expect_len);
  if (!mask->validity) return mask->values;
  int64_t nw = (mask->length + 63) / 64;
  BufferPtr out = alloc_buffer(ctx, (size_t)nw * 8);
  if (nw) hipLaunchKernelGGL(k_and_words, dim3(grid_for(nw, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)mask->values->ptr, (const uint64_t*)mask->validity->ptr, (uint64_t*)out->ptr, nw);
  KERNEL_CHECK();
  return out;
}

// OR `n` bits of src (all ones if src == null), starting at src bit 0, into dst starting at bit dst_off.
__global__ void k_or_bits(uint64_t* dst, int64_t dst_off, const uint64_t* src, int64_t n) {
  int64_t first_w = dst_off >> 6, last_w = (dst_off + n - 1) >> 6;
  int64_t w = first_w + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w > last_w) return;
  int64_t lo = w * 64 > dst_off ? w * 64 : dst_off, hi = (w + 1) * 64 < dst_off + n ? (w + 1) * 64 : dst_off + n;   // dst bit range
  int64_t s0 = lo - dst_off; int cnt = (int)(hi - lo);
  uint64_t bits;
  if (!src) bits = ~0ull;
  else { int64_t sw = s0 >> 6; int sh = (int)(s0 & 63); bits = src[sw] >> sh; if (sh && (s0 + cnt - 1) >> 6 != sw) bits |= src[sw + 1] << (64 - sh); }
  if (cnt < 64) bits &= (1ull << cnt) - 1ull;
  bits <<= (lo & 63);
  atomicOr((unsigned long long*)&dst[w], (unsigned long long)bits);
}
static void or_bits(dfgpu_ctx* ctx, uint64_t* dst, int64_t dst_off, const uint64_t* src, int64_t n) {
  if (n <= 0) return;
  int64_t words = ((dst_off + n - 1) >> 6) - (dst_off >> 6) + 1;
  hipLaunchKernelGGL(k_or_bits, dim3(grid_for(words, BLOCK)), dim3(BLOCK), 0, ctx->stream, dst, dst_off, src, n);
  KERNEL_CHECK();
}
// concat of Utf8 arrays.  An input's bytes are offsets[0] .. offsets[len] of its value buffer: a slice starts above 0, and values_bytes (what the host knows) is only an upper
// bound of the end.  Copying values_bytes per input and shifting the offsets by a running sum of them (what this did before) leaves a gap between two inputs, and the shared
// offset entry between them cannot say so: the last string of the first input swallowed the gap.  The bases are therefore computed on the device from the offsets themselves.
struct Utf8Part { const int32_t* offsets; const uint8_t* values; int64_t len; int64_t row; };
__global__ void k_concat_utf8_bases(const Utf8Part* parts, int n, int32_t* first, int32_t* base) {      // one thread: n is the number of batches, not of rows
  if (blockIdx.x || threadIdx.x) return;
  int64_t run = 0;
  for (int i = 0; i < n; i++) { const int32_t f = parts[i].len ? parts[i].offsets[0] : 0, l = parts[i].len ? parts[i].offsets[parts[i].len] : 0; first[i] = f; base[i] = (int32_t)run; run += l - f; }
  base[n] = (int32_t)run;
}
__global__ void __launch_bounds__(BLOCK) k_concat_utf8_part(const Utf8Part* parts, int i, const int32_t* first, const int32_t* base, int32_t* out_offsets, uint8_t* out_values) {
  const Utf8Part p = parts[i]; const int32_t f = first[i], b = base[i]; const int64_t used = (int64_t)base[i + 1] - b;
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x, stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t k = t; k <= p.len; k += stride) out_offsets[p.row + k] = p.offsets[k] - f + b;           // entry row + len == the next part's entry row: both write the same value
  // 16 output bytes per step, aligned in the output; the source is read byte-wise (its alignment relative to the output is arbitrary)
  uint8_t* dst = out_values + b; const uint8_t* src = p.values + f;
  const int64_t head = min(used, (int64_t)((16 - ((uintptr_t)dst & 15)) & 15));
  for (int64_t k = t; k < head; k += stride) dst[k] = src[k];
  const int64_t chunks = (used - head) >> 4;
  for (int64_t c = t; c < chunks; c += stride) {
    const uint8_t* q = src + head + (c << 4); uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = (uint32_t)q[4 * j] | ((uint32_t)q[4 * j + 1] << 8) | ((uint32_t)q[4 * j + 2] << 16) | ((uint32_t)q[4 * j + 3] << 24);
    *(uint4*)(dst + head + (c << 4)) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  for (int64_t k = head + (chunks << 4) + t; k < used; k += stride) dst[k] = src[k];
}
__global__ void k_popcount(const uint64_t* bits, int64_t n, unsigned long long* total) {
  int64_t nw = (n + 63) >> 6; unsigned long long c = 0;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = bits[w];
    if (w == nw - 1 && (n & 63)) x &= (1ull << (n & 63)) - 1ull;
    c += __popcll(x);
  }
  c = wave_sum(c);
  if (lane_id() == 0 && c) atomicAdd(total, c);
}
int64_t count_set_bits(dfgpu_ctx* ctx, const uint64_t* bits, int64_t n) {
  if (n == 0) return 0;
  HIP_CHECK(hipMemsetAsync(ctx->d_scratch64 + 62, 0, 8, ctx->stream));
  hipLaunchKernelGGL(k_popcount, dim3(grid_for((n + 63) / 64, BLOCK, 1024)), dim3(BLOCK), 0, ctx->stream, bits, n, (unsigned long long*)(ctx->d_scratch64 + 62));
  KERNEL_CHECK();
  return (int64_t)read_scratch(ctx, 62);
}

static void validate_type(int32_t t) { if (t < DFGPU_BOOL || t > DFGPU_DICTIONARY) fail(DFGPU_INVALID_ARGUMENT, "unknown type id %d", t); }

static dfgpu_array* import_desc(dfgpu_ctx* ctx, const dfgpu_array_desc* d, bool copy, const std::shared_ptr<void>& owner = nullptr) {
  validate_type(d->type);
  if (d->length < 0) fail(DFGPU_INVALID_ARGUMENT, "negative length");
  if (d->length > 0xFFFFFFF0ll) fail(DFGPU_NOT_IMPLEMENTED, "arrays above 2^32-16 rows are not supported (UInt32 row ids, joins/utils.rs probe indices)");
  ArrayHolder h(new_array(ctx, d->type, d->length, d->precision, d->scale));
  dfgpu_array* a = h.get();
  a->null_count = d->validity ? d->null_count : 0; a->key_type = d->key_type;
  a->base_column = true;          // handed in by the caller (a table column, a scan's output): it outlives the operators that read it, so statistics memoised on it pay off
  int64_t n = d->length;
  size_t vbytes;
  int32_t vt = d->type == DFGPU_DICTIONARY ? d->key_type : d->type;
  if (d->type == DFGPU_UTF8) vbytes = (size_t)d->values_bytes;
  else if (vt == DFGPU_BOOL) vbytes = (size_t)(n + 7) / 8;
  else { if (!type_width(vt)) fail(DFGPU_INVALID_ARGUMENT, "bad value type %d", vt); vbytes = (size_t)n * type_width(vt); }
  auto put = [&](const void* src, size_t bytes, size_t padded) -> BufferPtr {
    if (!copy) {
      if (((uintptr_t)src & 7) != 0) fail(DFGPU_INVALID_ARGUMENT, "device buffers must be 8-byte aligned");
      BufferPtr bb = borrow_buffer(src, bytes); bb->owner = owner; return bb;
    }
    BufferPtr b = alloc_buffer(ctx, padded, true);
    if (bytes) HIP_CHECK(hipMemcpyAsync(b->ptr, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return b;
  };
  if ((vbytes && !d->values)) fail(DFGPU_INVALID_ARGUMENT, "values buffer is null");
  a->values = put(d->values, vbytes, vt == DFGPU_BOOL && d->type != DFGPU_UTF8 ? bitmap_bytes(n) : vbytes);
  a->values_bytes = d->type == DFGPU_UTF8 ? d->values_bytes : 0;
  if (d->validity) a->validity = put(d->validity, (size_t)(n + 7) / 8, bitmap_bytes(n));
  if (d->type == DFGPU_UTF8) { if (!d->offsets) fail(DFGPU_INVALID_ARGUMENT, "utf8 needs offsets"); a->offsets = put(d->offsets, (size_t)(n + 1) * 4, (size_t)(n + 1) * 4); }
  if (d->type == DFGPU_DICTIONARY) {
    if (!d->dictionary) fail(DFGPU_INVALID_ARGUMENT, "dictionary array without dictionary");
    if (d->dictionary->type == DFGPU_DICTIONARY) fail(DFGPU_NOT_IMPLEMENTED, "nested dictionaries");
    if (!is_signed_int(d->key_type) && !is_unsigned_int(d->key_type)) fail(DFGPU_INVALID_ARGUMENT, "dictionary key type %d", d->key_type);
    a->dictionary = import_desc(ctx, d->dictionary, copy, owner);
  }
  if (copy) {
    if (n == 1 && d->type != DFGPU_UTF8 && d->type != DFGPU_DICTIONARY) {      // scalar Datum mirror
      a->has_host_scalar = true; memset(a->host_scalar, 0, 16); memcpy(a->host_scalar, d->values, vt == DFGPU_BOOL ? 1 : vbytes);
      a->host_scalar_valid = !d->validity || (d->validity[0] & 1);
    }
    HIP_CHECK(hipStreamSynchronize(ctx->stream));   // pageable host memory may be reused by the caller on return
  }
  return h.release();
}

}  // namespace dfgpu

using namespace dfgpu;

namespace dfgpu {
// ---- lists of fixed-width values in the Utf8 layout (the List<T> state of COUNT(DISTINCT), physical-expr/src/aggregate/count_distinct/native.rs state()): offsets are BYTE
// offsets into the packed values, so take / concat / slice / partition / exchange / Arrow export move such a column as they move strings.
__global__ void __launch_bounds__(BLOCK) k_list_lens(const int64_t* counts, int64_t n, uint32_t w, uint32_t* lens, uint32_t* flags) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i > n) return;
  if (i == n) { lens[i] = 0; return; }
  const int64_t c = counts[i];
  if (c < 0 || c > (int64_t)(0x7FFFFFF0u / w)) { atomicOr(flags, DFGPU_FLAG_OOB); lens[i] = 0; } else lens[i] = (uint32_t)c * w;
}
// element e of the flattened list (byte first + e * w) belongs to the row whose [offsets[r], offsets[r + 1]) holds it
__global__ void __launch_bounds__(BLOCK) k_list_row_of(const int32_t* offsets, int64_t n, int64_t ne, uint32_t w, uint32_t* row_of) {
  const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (e >= ne) return;
  const int64_t at = (int64_t)offsets[0] + e * w;
  int64_t lo = 0, hi = n - 1;                            // the last row with offsets[r] <= at (rows of no elements share their offset with the next: the last one wins only if it holds `at`)
  while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if ((int64_t)offsets[mid] <= at) lo = mid; else hi = mid - 1; }
  row_of[e] = (uint32_t)lo;
}
// ---- lists of STRINGS (the List<Utf8> state of COUNT(DISTINCT) over a Utf8 argument, count_distinct/bytes.rs:47-75): a row of the list column holds its strings back to back,
// each as a 4-byte little-endian length followed by its bytes -- still one Utf8-layout column, moved by everything that moves strings.
__global__ void __launch_bounds__(BLOCK) k_slist_sizes(const int32_t* voff, int64_t nv, uint32_t* enc) {        // enc[j] = 4 + length of value j; enc[nv] = 0
  const int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (j > nv) return;
  enc[j] = j == nv ? 0u : 4u + (uint32_t)(voff[j + 1] - voff[j]);
}
__global__ void __launch_bounds__(BLOCK) k_slist_offsets(const uint32_t* cpre /*[n + 1] first value of every row*/, const uint32_t* epos /*[nv + 1]*/, int64_t n, int32_t* out_off) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i <= n) out_off[i] = (int32_t)epos[cpre[i]];
}
__global__ void __launch_bounds__(BLOCK) k_slist_encode(const int32_t* voff, const uint8_t* vbytes, int64_t nv, const uint32_t* epos, uint8_t* out) {
  const int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (j >= nv) return;
  const uint32_t len = (uint32_t)(voff[j + 1] - voff[j]); uint8_t* o = out + epos[j]; const uint8_t* src = vbytes + voff[j];
  o[0] = (uint8_t)len; o[1] = (uint8_t)(len >> 8); o[2] = (uint8_t)(len >> 16); o[3] = (uint8_t)(len >> 24);
  for (uint32_t b = 0; b < len; b++) o[4 + b] = src[b];
}
// a row's strings: pass 0 counts them (cnt[i]), pass 1 writes, for value vstart[i] + k, its length, the byte it starts at inside the list's values and its row
__global__ void __launch_bounds__(BLOCK) k_slist_walk(const int32_t* loff, const uint8_t* lbytes, int64_t n, const uint32_t* vstart, uint32_t* cnt, uint32_t* lens, uint32_t* spos, uint32_t* row_of, uint32_t* flags) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i >= n) return;
  int64_t at = loff[i]; const int64_t end = loff[i + 1]; uint32_t k = 0;
  while (at < end) {
    if (at + 4 > end) { atomicOr(flags, DFGPU_FLAG_OOB); break; }
    const uint32_t len = (uint32_t)lbytes[at] | ((uint32_t)lbytes[at + 1] << 8) | ((uint32_t)lbytes[at + 2] << 16) | ((uint32_t)lbytes[at + 3] << 24);
    if (at + 4 + (int64_t)len > end) { atomicOr(flags, DFGPU_FLAG_OOB); break; }
    if (lens) { const uint32_t v = vstart[i] + k; lens[v] = len; spos[v] = (uint32_t)(at + 4); row_of[v] = (uint32_t)i; }
    k++; at += 4 + (int64_t)len;
  }
  if (cnt) cnt[i] = k;
}
__global__ void __launch_bounds__(BLOCK) k_slist_copy(const uint8_t* lbytes, const uint32_t* spos, const int32_t* voff, int64_t nv, uint8_t* out) {
  const int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (j >= nv) return;
  const uint32_t len = (uint32_t)(voff[j + 1] - voff[j]); const uint8_t* src = lbytes + spos[j]; uint8_t* o = out + voff[j];
  for (uint32_t b = 0; b < len; b++) o[b] = src[b];
}
}  // namespace dfgpu

extern "C" {

const char* dfgpu_version(void) { return "dfgpu 0.1 (gfx950)"; }

dfgpu_status dfgpu_ctx_create(int32_t device_id, void* stream, dfgpu_ctx** out) {
  if (!out) return DFGPU_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) return DFGPU_EXECUTION;
  auto* c = new dfgpu_ctx();
  c->device = device_id;
  dfgpu_status st = guard(c, [&] {
    HIP_CHECK(hipSetDevice(device_id));
    hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else { HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    c->alloc_mu = new std::mutex(); c->free_blocks = new std::vector<std::pair<size_t, void*>>();
    HIP_CHECK(hipMalloc((void**)&c->d_flags, 256)); HIP_CHECK(hipMemset(c->d_flags, 0, 256));
    HIP_CHECK(hipMalloc((void**)&c->d_scratch64, 64 * 8)); HIP_CHECK(hipMemset(c->d_scratch64, 0, 64 * 8));
    HIP_CHECK(hipHostMalloc((void**)&c->h_pinned, dfgpu_ctx::PINNED_WORDS * 8, hipHostMallocCoherent | hipHostMallocMapped)); memset(c->h_pinned, 0, dfgpu_ctx::PINNED_WORDS * 8);
  });
  if (st != DFGPU_OK) { fprintf(stderr, "dfgpu_ctx_create: %s\n", c->err.c_str()); delete c; return st; }
  { std::lock_guard<std::mutex> l(g_mu); g_ctx.emplace_back(c, new CtxRefs()); }
  *out = c;
  return DFGPU_OK;
}
void dfgpu_ctx_destroy(dfgpu_ctx* ctx) { if (ctx) ctx_unref(ctx); }
const char* dfgpu_last_error(const dfgpu_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }
void* dfgpu_ctx_stream(dfgpu_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
dfgpu_status dfgpu_ctx_synchronize(dfgpu_ctx* ctx) {
  if (!ctx) return DFGPU_INVALID_ARGUMENT;
  return guard(ctx, [&] {
    HIP_CHECK(hipSetDevice(ctx->device));
    // a pending flag check is itself a read-back behind everything enqueued: taken first, the stream synchronisation that follows finds the stream drained (no second wake-up)
    try { flush_flags(ctx); } catch (...) { (void)hipStreamSynchronize(ctx->stream); throw; }
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
  });
}
dfgpu_status dfgpu_ctx_set_option(dfgpu_ctx* ctx, const char* key, int64_t value) {
  return guard(ctx, [&] {
    std::string k = key ? key : "";
    if (k == "force_hash_collisions") ctx->force_hash_collisions = value != 0;
    else if (k == "first_seen_group_order") ctx->first_seen_group_order = value != 0;
    else if (k == "join_rank_index") ctx->join_rank_index = value != 0;
    else if (k == "join_rank_index_unsorted") ctx->join_rank_index_unsorted = value != 0;
    else if (k == "join_lazy_build_rows") ctx->join_lazy_build_rows = value != 0;
    else if (k == "join_selection_output") ctx->join_selection_output = value != 0;
    else if (k == "agg_order_inverse_map") ctx->agg_order_inverse_map = value != 0;
    else if (k == "join_bitmap_partitioned") ctx->join_bitmap_partitioned = value != 0;
    else if (k == "join_bitmap_partitioned_min_rows") ctx->join_bitmap_partitioned_min_rows = value;
    else if (k == "join_key_packing") ctx->join_key_packing = value != 0;
    else if (k == "group_run_detection") ctx->group_run_detection = value != 0;
    else if (k == "group_lazy_keys") ctx->group_lazy_keys = value != 0;
    else if (k == "group_dictionary_canon") ctx->group_dictionary_canon = value != 0;
    else if (k == "join_swap_small_semi") ctx->join_swap_small_semi = value != 0;
    else if (k == "fused_aggregate_min_rows") ctx->fused_aggregate_min_rows = value;
    else if (k == "sort_packed_keys") ctx->sort_packed_keys = value != 0;
    else if (k == "memory_limit") ctx->memory_limit = value;
    else if (k == "mailbox_readback") ctx->mailbox_readback = value != 0;
    else if (k == "trim_cache") {        // give the freed blocks the ctx keeps for reuse back to the driver (≙ MemoryPool::shrink): after the stream has drained, so no kernel still reads them
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
      std::lock_guard<std::mutex> l(*ctx->alloc_mu); for (auto& x : *ctx->free_blocks) (void)hipFree(x.second); ctx->free_blocks->clear(); ctx->cached_bytes = 0;
    }
    else if (k == "agg_spill_state_bytes") ctx->agg_spill_state_bytes = value;
    else if (k == "sort_estimate_ranges") ctx->sort_estimate_ranges = value != 0;
    else if (k == "sort_topk_words_min_rows") ctx->sort_topk_words_min_rows = value < 2 ? 2 : value;
    else if (k == "partition_two_round_staging") ctx->partition_two_round_staging = value != 0;
    else if (k == "sort_payload_in_last_pass") ctx->sort_payload_in_last_pass = value != 0;
    else if (k == "sort_onesweep_fused_finish") ctx->sort_onesweep_fused_finish = value != 0;
    else if (k == "sort_one_block_max_rows") ctx->sort_one_block_max_rows = value < 0 ? 0 : value;
    else if (k == "sort_onesweep_min_rows") ctx->sort_onesweep_min_rows = value < 2 ? 2 : value;
    else if (k == "sort_onesweep_rows") ctx->sort_onesweep_rows = value == 16 ? 16 : value > 0 ? 8 : 0;
    else if (k == "sort_fused_small_passes") ctx->sort_fused_small_passes = value != 0;
    else if (k == "sort_packed_min_rows") ctx->sort_packed_min_rows = value < 2 ? 2 : value;
    else if (k == "sort_spill_bytes") ctx->sort_spill_bytes = value;
    else if (k == "spm_merge_rows") { if (value < 0) fail(DFGPU_INVALID_ARGUMENT, "spm_merge_rows: >= 0"); ctx->spm_merge_rows = value; }
    else if (k == "sort_spill_ranges") { if (value < 1 || value > 4096) fail(DFGPU_INVALID_ARGUMENT, "sort_spill_ranges: 1 .. 4096"); ctx->sort_spill_ranges = value; }
    else if (k == "agg_spill_ranges") { if (value < 1 || value > 4096) fail(DFGPU_INVALID_ARGUMENT, "agg_spill_ranges: 1 .. 4096"); ctx->agg_spill_ranges = value; }
    else if (k == "collect_metrics") ctx->collect_metrics = value != 0;
    else if (k == "agg_partitioned") ctx->agg_partitioned = value != 0;
    else if (k == "agg_partitioned_force") ctx->agg_partitioned_force = value != 0;
    else if (k == "agg_partitioned_min_rows") ctx->agg_partitioned_min_rows = value;
    else if (k == "agg_pack_estimate_min_rows") ctx->agg_pack_estimate_min_rows = value;
    else if (k == "join_partitioned") ctx->join_partitioned = value != 0;
    else if (k == "join_partitioned_min_build") ctx->join_partitioned_min_build = value;
    else if (k == "join_partitioned_min_probe") ctx->join_partitioned_min_probe = value;
    else if (k == "join_partitioned_big") ctx->join_partitioned_big = value != 0;
    else if (k == "join_partitioned_hashed") ctx->join_partitioned_hashed = value != 0;
    else if (k == "join_partitioned_hash_mask") ctx->join_partitioned_hash_mask = value <= 0 ? ~0ull : (uint64_t)value;
    else if (k == "join_partition_rows") ctx->join_partition_rows = value;
    else if (k == "defer_flag_checks") {            // nests: +1 enters a deferred region, 0 leaves it and raises what the region deferred
      if (value) ctx->defer_flag_checks++;
      else { if (ctx->defer_flag_checks > 0) ctx->defer_flag_checks--; if (ctx->defer_flag_checks == 0) flush_flags(ctx); }
    }
    else fail(DFGPU_INVALID_ARGUMENT, "unknown option '%s'", k.c_str());
  });
}

dfgpu_status dfgpu_ctx_get_option(dfgpu_ctx* ctx, const char* key, int64_t* out) {
  return guard(ctx, [&] {
    std::string k = key ? key : "";
    if (!out) fail(DFGPU_INVALID_ARGUMENT, "ctx_get_option: null out");
    if (k == "force_hash_collisions") *out = ctx->force_hash_collisions;
    else if (k == "first_seen_group_order") *out = ctx->first_seen_group_order;
    else if (k == "join_rank_index") *out = ctx->join_rank_index;
    else if (k == "join_rank_index_unsorted") *out = ctx->join_rank_index_unsorted ? 1 : 0;
    else if (k == "join_lazy_build_rows") *out = ctx->join_lazy_build_rows ? 1 : 0;
    else if (k == "join_selection_output") *out = ctx->join_selection_output ? 1 : 0;
    else if (k == "agg_order_inverse_map") *out = ctx->agg_order_inverse_map ? 1 : 0;
    else if (k == "join_bitmap_partitioned") *out = ctx->join_bitmap_partitioned ? 1 : 0;
    else if (k == "join_bitmap_partitioned_min_rows") *out = ctx->join_bitmap_partitioned_min_rows;
    else if (k == "join_key_packing") *out = ctx->join_key_packing;
    else if (k == "group_run_detection") *out = ctx->group_run_detection;
    else if (k == "group_dictionary_canon") *out = ctx->group_dictionary_canon;
    else if (k == "join_swap_small_semi") *out = ctx->join_swap_small_semi;
    else if (k == "fused_aggregate_min_rows") *out = ctx->fused_aggregate_min_rows;
    else if (k == "sort_packed_keys") *out = ctx->sort_packed_keys;
    else if (k == "memory_limit") *out = ctx->memory_limit;
    else if (k == "mailbox_readback") *out = ctx->mailbox_readback ? 1 : 0;
    else if (k == "agg_spill_state_bytes") *out = ctx->agg_spill_state_bytes;
    else if (k == "sort_estimate_ranges") *out = ctx->sort_estimate_ranges ? 1 : 0;
    else if (k == "sort_packed_min_rows") *out = ctx->sort_packed_min_rows;
    else if (k == "sort_spill_bytes") *out = ctx->sort_spill_bytes;
    else if (k == "spm_merge_rows") *out = ctx->spm_merge_rows;
    else if (k == "sort_spill_ranges") *out = ctx->sort_spill_ranges;
    else if (k == "agg_spill_ranges") *out = ctx->agg_spill_ranges;
    else if (k == "collect_metrics") *out = ctx->collect_metrics;
    else if (k == "agg_preaggregate_distinct") *out = ctx->pa_last_distinct;      // read only
    else if (k == "live_bytes") *out = (int64_t)ctx->live_bytes;              // read only: device bytes held by live buffers of this ctx
    else if (k == "cached_bytes") *out = (int64_t)ctx->cached_bytes;          // read only: freed blocks kept for reuse
    else if (k == "live_bytes") *out = (int64_t)ctx->live_bytes;              // read only: device memory the ctx's arrays hold right now
    else if (k == "agg_partitioned") *out = ctx->agg_partitioned;
    else if (k == "agg_partitioned_force") *out = ctx->agg_partitioned_force;
    else if (k == "agg_partitioned_min_rows") *out = ctx->agg_partitioned_min_rows;
    else if (k == "agg_pack_estimate_min_rows") *out = ctx->agg_pack_estimate_min_rows;
    else if (k == "join_partitioned") *out = ctx->join_partitioned;
    else if (k == "join_partitioned_min_build") *out = ctx->join_partitioned_min_build;
    else if (k == "join_partitioned_min_probe") *out = ctx->join_partitioned_min_probe;
    else if (k == "join_partitioned_hashed") *out = ctx->join_partitioned_hashed ? 1 : 0;
    else if (k == "join_partitioned_hash_mask") *out = ctx->join_partitioned_hash_mask == ~0ull ? 0 : (int64_t)ctx->join_partitioned_hash_mask;
    else if (k == "join_partition_rows") *out = ctx->join_partition_rows;
    else if (k == "defer_flag_checks") *out = ctx->defer_flag_checks;
    else fail(DFGPU_INVALID_ARGUMENT, "unknown option '%s'", k.c_str());
  });
}

dfgpu_status dfgpu_ctx_set_row_selection(dfgpu_ctx* ctx, const dfgpu_array* mask) {
  return guard(ctx, [&] {
    if (!mask) { ctx->row_selection.reset(); ctx->row_selection_len = 0; return; }
    ctx->row_selection = effective_mask(ctx, mask, mask->length); ctx->row_selection_len = mask->length;
  });
}
dfgpu_status dfgpu_mask_count(dfgpu_ctx* ctx, const dfgpu_array* mask, int64_t* out) {
  return guard(ctx, [&] {
    if (!mask || !out) fail(DFGPU_INVALID_ARGUMENT, "mask_count: null argument");
    BufferPtr m = effective_mask(ctx, mask, mask->length);
    *out = mask->length ? count_set_bits(ctx, (const uint64_t*)m->ptr, mask->length) : 0;
  });
}
// A span id outlives the call that made it (the plan layer resolves them when metrics are read, or when the plan goes away): every entry point first asks the registry
// whether the ctx still exists, and the table is guarded -- a plan's partitions meter concurrently.  Events of spans nobody resolved are destroyed with the ctx.
dfgpu_status dfgpu_span_begin(dfgpu_ctx* ctx, int64_t* out_span) {
  if (!ctx || !refs_of(ctx)) return DFGPU_INVALID_ARGUMENT;
  return guard(ctx, [&] {
    if (!out_span) fail(DFGPU_INVALID_ARGUMENT, "span_begin: null argument");
    dfgpu_ctx::Span sp; HIP_CHECK(hipEventCreate(&sp.start)); HIP_CHECK(hipEventCreate(&sp.stop));
    HIP_CHECK(hipEventRecord(sp.start, ctx->stream));
    std::lock_guard<std::mutex> l(ctx->span_mu);
    size_t k = 0; for (; k < ctx->spans.size(); k++) if (!ctx->spans[k].start) break;
    if (k == ctx->spans.size()) ctx->spans.push_back(sp); else ctx->spans[k] = sp;
    *out_span = (int64_t)k;
  });
}
dfgpu_status dfgpu_span_end(dfgpu_ctx* ctx, int64_t span) {
  if (!ctx || !refs_of(ctx)) return DFGPU_INVALID_ARGUMENT;
  return guard(ctx, [&] {
    std::lock_guard<std::mutex> l(ctx->span_mu);
    if (span < 0 || (size_t)span >= ctx->spans.size() || !ctx->spans[(size_t)span].start) fail(DFGPU_INVALID_ARGUMENT, "span_end: unknown span");
    HIP_CHECK(hipEventRecord(ctx->spans[(size_t)span].stop, ctx->stream));
  });
}
dfgpu_status dfgpu_span_elapsed_ns(dfgpu_ctx* ctx, int64_t span, int64_t* out_ns) {
  if (!ctx || !refs_of(ctx)) return DFGPU_INVALID_ARGUMENT;
  return guard(ctx, [&] {
    dfgpu_ctx::Span sp;
    { std::lock_guard<std::mutex> l(ctx->span_mu);
      if (!out_ns || span < 0 || (size_t)span >= ctx->spans.size() || !ctx->spans[(size_t)span].start) fail(DFGPU_INVALID_ARGUMENT, "span_elapsed: unknown span");
      sp = ctx->spans[(size_t)span]; ctx->spans[(size_t)span].start = ctx->spans[(size_t)span].stop = nullptr; }      // the slot is free again; the events are this call's
    hipError_t e = hipEventSynchronize(sp.stop); float ms = 0; if (e == hipSuccess) e = hipEventElapsedTime(&ms, sp.start, sp.stop);
    (void)hipEventDestroy(sp.start); (void)hipEventDestroy(sp.stop);
    HIP_CHECK(e);
    *out_ns = (int64_t)((double)ms * 1e6);
  });
}
dfgpu_status dfgpu_profile_enable(dfgpu_ctx* ctx, int32_t on) {
  if (!ctx) return DFGPU_INVALID_ARGUMENT;
  ctx->profile = on != 0; return DFGPU_OK;
}
dfgpu_status dfgpu_profile_select(dfgpu_ctx* ctx, const char* kernel_name) {
  if (!ctx) return DFGPU_INVALID_ARGUMENT;
  ctx->profile_only = kernel_name ? kernel_name : ""; return DFGPU_OK;
}
dfgpu_status dfgpu_profile_read(dfgpu_ctx* ctx, char* buf, int64_t capacity) {
  return guard(ctx, [&] {
    if (!buf || capacity < 1) fail(DFGPU_INVALID_ARGUMENT, "profile_read: no buffer");
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    std::vector<std::string> names; std::vector<double> ms; std::vector<int64_t> cnt;
    for (auto& r : ctx->prof) {
      float t = 0; if (hipEventElapsedTime(&t, r.start, r.stop) != hipSuccess) t = 0;
      (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop);
      size_t k = 0; for (; k < names.size(); k++) if (names[k] == r.name) break;
      if (k == names.size()) { names.push_back(r.name); ms.push_back(0); cnt.push_back(0); }
      ms[k] += t; cnt[k]++;
    }
    ctx->prof.clear();
    std::string out;
    for (auto& kv : ctx->sync_counts) { char line[256]; snprintf(line, sizeof line, "%s %lld 0.0\n", kv.first.c_str(), (long long)kv.second); out += line; }
    ctx->sync_counts.clear();
    for (size_t k = 0; k < names.size(); k++) { char line[256]; snprintf(line, sizeof line, "%s %lld %.6f\n", names[k].c_str(), (long long)cnt[k], ms[k]); out += line; }
    if ((int64_t)out.size() + 1 > capacity) out.resize((size_t)capacity - 1);
    memcpy(buf, out.c_str(), out.size() + 1);
  });
}

dfgpu_status dfgpu_array_import_host(dfgpu_ctx* ctx, const dfgpu_array_desc* host, dfgpu_array** out) {
  if (!ctx) return DFGPU_INVALID_ARGUMENT;
  return guard(ctx, [&] { if (!host || !out) fail(DFGPU_INVALID_ARGUMENT, "array_import_host: null argument"); HIP_CHECK(hipSetDevice(ctx->device)); *out = import_desc(ctx, host, true); });
}
dfgpu_status dfgpu_array_wrap_device(dfgpu_ctx* ctx, const dfgpu_array_desc* dev, dfgpu_array** out) {
  if (!ctx) return DFGPU_INVALID_ARGUMENT;
  return guard(ctx, [&] { if (!dev || !out) fail(DFGPU_INVALID_ARGUMENT, "array_wrap_device: null argument"); *out = import_desc(ctx, dev, false); });
}
dfgpu_status dfgpu_array_wrap_device_owned(dfgpu_ctx* ctx, const dfgpu_array_desc* dev, void (*release)(void*), void* cookie, dfgpu_array** out) {
  // the token is created first: if the import fails the callback still fires exactly once (the caller gave the reference away)
  std::shared_ptr<void> owner;
  if (release) owner = std::shared_ptr<void>(cookie ? cookie : (void*)&owner, [release, cookie](void*) { release(cookie); });
  if (!ctx) return DFGPU_INVALID_ARGUMENT;          // `owner` goes out of scope: the callback fires, as on every other failure
  return guard(ctx, [&] { if (!dev || !out) fail(DFGPU_INVALID_ARGUMENT, "array_wrap_device_owned: null argument"); *out = import_desc(ctx, dev, false, owner); });
}
dfgpu_status dfgpu_array_describe(const dfgpu_array* a, dfgpu_array_desc* o) {
  if (!a || !o) return DFGPU_INVALID_ARGUMENT;
  auto fill = [](const dfgpu_array* x, dfgpu_array_desc* d) {
    memset(d, 0, sizeof *d);
    d->type = x->type; d->precision = x->precision; d->scale = x->scale; d->key_type = x->key_type; d->length = x->length; d->null_count = x->null_count;
    d->values = x->values ? x->values->ptr : nullptr; d->validity = x->validity ? (const uint8_t*)x->validity->ptr : nullptr;
    d->offsets = x->offsets ? (const int32_t*)x->offsets->ptr : nullptr; d->values_bytes = x->values_bytes;
  };
  fill(a, o);
  if (a->dictionary) { auto* m = const_cast<dfgpu_array*>(a); fill(a->dictionary, &m->dict_desc); o->dictionary = &m->dict_desc; }
  return DFGPU_OK;
}
dfgpu_status dfgpu_array_export_host(dfgpu_ctx* ctx, const dfgpu_array* a, void* values, uint8_t* validity, int32_t* offsets) {
  if (!ctx) return DFGPU_INVALID_ARGUMENT;
  return guard(ctx, [&] {
    if (!a) fail(DFGPU_INVALID_ARGUMENT, "array_export_host: null array");
    HIP_CHECK(hipSetDevice(ctx->device));
    materialize_ids(ctx, a);
    flush_flags(ctx);                                   // data never leaves the device past a deferred kernel error
    int64_t n = a->length; int32_t vt = a->type == DFGPU_DICTIONARY ? a->key_type : a->type;
    size_t vbytes = a->type == DFGPU_UTF8 ? (size_t)a->values_bytes : (vt == DFGPU_BOOL ? (size_t)(n + 7) / 8 : (size_t)n * type_width(vt));
    if (values && vbytes) HIP_CHECK(hipMemcpyAsync(values, a->values->ptr, vbytes, hipMemcpyDeviceToHost, ctx->stream));
    if (validity && a->validity && n) HIP_CHECK(hipMemcpyAsync(validity, a->validity->ptr, (size_t)(n + 7) / 8, hipMemcpyDeviceToHost, ctx->stream));
    if (offsets && a->offsets) HIP_CHECK(hipMemcpyAsync(offsets, a->offsets->ptr, (size_t)(n + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
  });
}
void dfgpu_array_retain(dfgpu_array* a) { if (a) a->refs.fetch_add(1); }
void dfgpu_array_release(dfgpu_array* a) {
  if (!a) return;
  if (a->refs.fetch_sub(1) == 1) { if (a->dictionary) dfgpu_array_release(a->dictionary); delete a; }
}
int32_t dfgpu_array_is_identity(const dfgpu_array* a) { return a && a->identity ? 1 : 0; }
int64_t dfgpu_array_length(const dfgpu_array* a) { return a ? a->length : 0; }
int64_t dfgpu_array_null_count(dfgpu_ctx* ctx, const dfgpu_array* a) {
  if (!a) return 0;
  if (a->null_count >= 0) return a->null_count;
  if (!a->validity) return 0;
  int64_t nc = -1;
  guard(ctx, [&] { nc = a->length - count_set_bits(ctx, (const uint64_t*)a->validity->ptr, a->length); const_cast<dfgpu_array*>(a)->null_count = nc; });
  return nc;
}

dfgpu_status dfgpu_array_make_dictionary(dfgpu_ctx* ctx, const dfgpu_array* keys, const dfgpu_array* values, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!keys || !values || !out) fail(DFGPU_INVALID_ARGUMENT, "make_dictionary: null argument");
    if (!(keys->type >= DFGPU_INT8 && keys->type <= DFGPU_UINT64)) fail(DFGPU_INVALID_ARGUMENT, "make_dictionary: keys must be an integer array (got type %d)", keys->type);
    if (values->type == DFGPU_DICTIONARY) fail(DFGPU_INVALID_ARGUMENT, "make_dictionary: values are a dictionary array already");
    ArrayHolder h(new_array(ctx, DFGPU_DICTIONARY, keys->length, values->precision, values->scale));
    dfgpu_array* a = h.get();
    a->key_type = keys->type; a->values = keys->values; a->validity = keys->validity; a->null_count = keys->validity ? -1 : 0;
    a->dictionary = const_cast<dfgpu_array*>(values); dfgpu_array_retain(a->dictionary);
    *out = h.release();
  });
}
dfgpu_status dfgpu_array_new_null(dfgpu_ctx* ctx, int32_t type, int32_t precision, int32_t scale, int64_t length, dfgpu_array** out) {
  return guard(ctx, [&] {
    validate_type(type);
    if (type == DFGPU_DICTIONARY) fail(DFGPU_NOT_IMPLEMENTED, "new_null dictionary");
    ArrayHolder h(new_array(ctx, type, length, precision, scale));
    size_t vb = type == DFGPU_UTF8 ? 0 : (type == DFGPU_BOOL ? bitmap_bytes(length) : (size_t)length * type_width(type));
    h.get()->values = alloc_buffer(ctx, vb, true);
    h.get()->validity = alloc_buffer(ctx, bitmap_bytes(length), true);
    if (type == DFGPU_UTF8) h.get()->offsets = alloc_buffer(ctx, (size_t)(length + 1) * 4, true);
    h.get()->null_count = length;
    *out = h.release();
  });
}

dfgpu_status dfgpu_array_new_zeros(dfgpu_ctx* ctx, int32_t type, int32_t precision, int32_t scale, int64_t length, dfgpu_array** out) {
  return guard(ctx, [&] {
    validate_type(type);
    if (!type_width(type) && type != DFGPU_BOOL) fail(DFGPU_INVALID_ARGUMENT, "new_zeros: fixed-width type expected");
    ArrayHolder h(new_fixed(ctx, type, length, precision, scale));
    if (type != DFGPU_BOOL && length) HIP_CHECK(hipMemsetAsync(h.get()->values->ptr, 0, (size_t)length * type_width(type), ctx->stream));
    *out = h.release();
  });
}

dfgpu_status dfgpu_array_iota(dfgpu_ctx* ctx, int64_t length, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!out || length < 0 || length > 0xFFFFFFF0ll) fail(DFGPU_INVALID_ARGUMENT, "array_iota: bad argument");
    HIP_CHECK(hipSetDevice(ctx->device));
    ArrayHolder h(new_fixed(ctx, DFGPU_UINT32, length));
    launch_iota_u32(ctx, (uint32_t*)h.get()->values->ptr, length, 0);
    h.get()->identity = true;
    *out = h.release();
  });
}
dfgpu_status dfgpu_array_slice(dfgpu_ctx* ctx, const dfgpu_array* a, int64_t offset, int64_t length, dfgpu_array** out) {
  if (!ctx) return DFGPU_INVALID_ARGUMENT;
  return guard(ctx, [&] {
    if (!a || !out) fail(DFGPU_INVALID_ARGUMENT, "array_slice: null argument");
    if (offset < 0 || length < 0 || offset + length > a->length) fail(DFGPU_INVALID_ARGUMENT, "slice [%lld, +%lld) outside array of %lld rows", (long long)offset, (long long)length, (long long)a->length);
    int32_t vt0 = a->type == DFGPU_DICTIONARY ? a->key_type : a->type;
    bool bit_aligned_needed = a->validity != nullptr || vt0 == DFGPU_BOOL;
    if (offset % 64 != 0 && bit_aligned_needed) {     // bitmaps cannot be re-based at bit granularity: copy through take with an iota index
      ArrayHolder idx(new_fixed(ctx, DFGPU_UINT32, length));
      launch_iota_u32(ctx, (uint32_t*)idx.get()->values->ptr, length, (uint32_t)offset);
      *out = take_impl(ctx, a, idx.get()->values->ptr, 4, nullptr, length);
      return;
    }
    ArrayHolder h(new_array(ctx, a->type, length, a->precision, a->scale));
    dfgpu_array* s = h.get(); s->key_type = a->key_type; s->values_bytes = a->values_bytes;
    auto sub = [&](const BufferPtr& b, size_t byte_off) { auto r = std::make_shared<Buffer>(); r->ptr = (char*)b->ptr + byte_off; r->bytes = b->bytes - byte_off; r->owned = false; r->parent = b; return r; };
    int32_t vt = a->type == DFGPU_DICTIONARY ? a->key_type : a->type;
    if (a->type == DFGPU_UTF8) { s->values = a->values; s->offsets = sub(a->offsets, (size_t)offset * 4); }
    else if (vt == DFGPU_BOOL) s->values = sub(a->values, (size_t)offset / 8);
    else s->values = sub(a->values, (size_t)offset * type_width(vt));
    if (a->validity) { s->validity = sub(a->validity, (size_t)offset / 8); s->null_count = -1; } else s->null_count = 0;
    if (a->dictionary) { s->dictionary = a->dictionary; dfgpu_array_retain(a->dictionary); }
    *out = h.release();
  });
}

dfgpu_status dfgpu_concat(dfgpu_ctx* ctx, const dfgpu_array* const* arrays, int32_t n, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (n < 1) fail(DFGPU_INVALID_ARGUMENT, "concat of zero arrays");
    const dfgpu_array* f = arrays[0];
    int64_t total = 0, total_bytes = 0; bool any_validity = false;
    for (int i = 0; i < n; i++) {
      const dfgpu_array* a = arrays[i];
      if (a->type != f->type || a->precision != f->precision || a->scale != f->scale) fail(DFGPU_INVALID_ARGUMENT, "concat: column types differ");
      if (a->type == DFGPU_DICTIONARY && (a->dictionary != f->dictionary || a->key_type != f->key_type)) fail(DFGPU_NOT_IMPLEMENTED, "concat of dictionary arrays with different dictionaries");
      total += a->length; total_bytes += a->values_bytes; any_validity |= (a->validity != nullptr);
    }
    if (total > 0xFFFFFFF0ll) fail(DFGPU_NOT_IMPLEMENTED, "concat above 2^32-16 rows");
    ArrayHolder h(new_array(ctx, f->type, total, f->precision, f->scale));
    dfgpu_array* o = h.get(); o->key_type = f->key_type;
    int32_t vt = f->type == DFGPU_DICTIONARY ? f->key_type : f->type;
    int w = type_width(vt);
    if (f->type == DFGPU_UTF8) { o->values = alloc_buffer(ctx, (size_t)total_bytes); o->offsets = alloc_buffer(ctx, (size_t)(total + 1) * 4, true); o->values_bytes = total_bytes; }
    else if (vt == DFGPU_BOOL) o->values = alloc_buffer(ctx, bitmap_bytes(total), true);
    else o->values = alloc_buffer(ctx, (size_t)total * w);
    if (any_validity) o->validity = alloc_buffer(ctx, bitmap_bytes(total), true); else o->null_count = 0;
    int64_t row = 0;
    BufferPtr dparts, dbases;
    if (f->type == DFGPU_UTF8) {               // o->values_bytes stays the upper bound sum(values_bytes): the exact total is offsets[total], known to the device
      if (total_bytes > 0x7FFFFFF0ll) fail(DFGPU_NOT_IMPLEMENTED, "concat of Utf8 columns above 2^31 bytes");
      std::vector<Utf8Part> parts; int64_t r = 0;
      for (int i = 0; i < n; i++) { parts.push_back(Utf8Part{arrays[i]->length ? (const int32_t*)arrays[i]->offsets->ptr : nullptr, arrays[i]->values ? (const uint8_t*)arrays[i]->values->ptr : nullptr, arrays[i]->length, r}); r += arrays[i]->length; }
      dparts = alloc_buffer(ctx, parts.size() * sizeof(Utf8Part)); dbases = alloc_buffer(ctx, (size_t)(2 * n + 2) * 4);
      HIP_CHECK(hipMemcpyAsync(dparts->ptr, parts.data(), parts.size() * sizeof(Utf8Part), hipMemcpyHostToDevice, ctx->stream));          // pageable source: staged by the runtime before the call returns
      int32_t* first = (int32_t*)dbases->ptr; int32_t* base = first + n;
      hipLaunchKernelGGL(k_concat_utf8_bases, dim3(1), dim3(1), 0, ctx->stream, (const Utf8Part*)dparts->ptr, n, first, base);
      for (int i = 0; i < n; i++) if (arrays[i]->length)
        hipLaunchKernelGGL(k_concat_utf8_part, dim3(grid_for(std::max<int64_t>(arrays[i]->length + 1, arrays[i]->values_bytes / 16 + 1), BLOCK, 2048)), dim3(BLOCK), 0, ctx->stream,
                           (const Utf8Part*)dparts->ptr, i, (const int32_t*)first, (const int32_t*)base, (int32_t*)o->offsets->ptr, (uint8_t*)o->values->ptr);
      KERNEL_CHECK();
    }
    for (int i = 0; i < n; i++) {
      const dfgpu_array* a = arrays[i];
      if (a->length == 0) continue;
      if (f->type == DFGPU_UTF8) {
      } else if (vt == DFGPU_BOOL) or_bits(ctx, (uint64_t*)o->values->ptr, row, (const uint64_t*)a->values->ptr, a->length);
      else HIP_CHECK(hipMemcpyAsync((char*)o->values->ptr + (size_t)row * w, a->values->ptr, (size_t)a->length * w, hipMemcpyDeviceToDevice, ctx->stream));
      if (any_validity) or_bits(ctx, (uint64_t*)o->validity->ptr, row, a->validity ? (const uint64_t*)a->validity->ptr : nullptr, a->length);
      row += a->length;
    }
    if (f->dictionary) { o->dictionary = f->dictionary; dfgpu_array_retain(f->dictionary); }
    *out = h.release();
  });
}

/* see include/dfgpu.h */
dfgpu_status dfgpu_list_from_counts(dfgpu_ctx* ctx, const dfgpu_array* counts, const dfgpu_array* values, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!counts || !values || !out) fail(DFGPU_INVALID_ARGUMENT, "list_from_counts: null argument");
    if (counts->type != DFGPU_INT64) fail(DFGPU_INVALID_ARGUMENT, "list_from_counts: counts must be Int64");
    if (values->type == DFGPU_UTF8) {            // a list of strings: every string as (u32 length, bytes)
      // (a validity buffer on `values` is not consulted: the values of a list are the non-NULL ones by construction -- COUNT(DISTINCT) interns its pairs under an IS NOT NULL mask)
      const int64_t n = counts->length, nv = values->length;
      BufferPtr enc = alloc_buffer(ctx, (size_t)(nv + 1) * 4 + 16), cpre = alloc_buffer(ctx, (size_t)(n + 1) * 4 + 16);
      hipLaunchKernelGGL(k_slist_sizes, dim3(grid_for(nv + 1, BLOCK)), dim3(BLOCK), 0, ctx->stream, nv ? (const int32_t*)values->offsets->ptr : (const int32_t*)nullptr, nv, (uint32_t*)enc->ptr);
      exclusive_scan_u32_inplace32(ctx, (uint32_t*)enc->ptr, nv + 1, ctx->d_scratch64 + 13);
      hipLaunchKernelGGL(k_list_lens, dim3(grid_for(n + 1, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const int64_t*)counts->values->ptr, n, 1u, (uint32_t*)cpre->ptr, ctx->d_flags);
      exclusive_scan_u32_inplace32(ctx, (uint32_t*)cpre->ptr, n + 1, ctx->d_scratch64 + 14);
      KERNEL_CHECK();
      const uint64_t* tot = read_scratch_range(ctx, 13, 2);
      const int64_t ebytes = (int64_t)tot[0], nvals = (int64_t)tot[1];
      if (nvals != nv) fail(DFGPU_INVALID_ARGUMENT, "list_from_counts: the counts add up to %lld values, the values hold %lld", (long long)nvals, (long long)nv);
      if (ebytes > 0x7FFFFFF0ll) fail(DFGPU_NOT_IMPLEMENTED, "list_from_counts: %lld bytes of list values exceed 32-bit offsets", (long long)ebytes);
      ArrayHolder h(new_array(ctx, DFGPU_UTF8, n)); dfgpu_array* o = h.get();
      o->offsets = alloc_buffer(ctx, (size_t)(n + 1) * 4 + 16); o->values = alloc_buffer(ctx, (size_t)ebytes + 16); o->values_bytes = ebytes; o->null_count = 0;
      hipLaunchKernelGGL(k_slist_offsets, dim3(grid_for(n + 1, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)cpre->ptr, (const uint32_t*)enc->ptr, n, (int32_t*)o->offsets->ptr);
      if (nv) hipLaunchKernelGGL(k_slist_encode, dim3(grid_for(nv, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const int32_t*)values->offsets->ptr, (const uint8_t*)values->values->ptr, nv, (const uint32_t*)enc->ptr, (uint8_t*)o->values->ptr);
      KERNEL_CHECK();
      check_flags(ctx, "list_from_counts");
      *out = h.release();
      return;
    }
    if (values->type == DFGPU_DICTIONARY || values->type == DFGPU_BOOL) fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: lists of type %d values on the device (fixed-width values only)", values->type);
    const int w = type_width(values->type); const int64_t n = counts->length, bytes = values->length * w;
    if (bytes > 0x7FFFFFF0ll) fail(DFGPU_NOT_IMPLEMENTED, "list_from_counts: %lld bytes of values exceed 32-bit offsets", (long long)bytes);
    ArrayHolder h(new_array(ctx, DFGPU_UTF8, n)); dfgpu_array* o = h.get();
    o->offsets = alloc_buffer(ctx, (size_t)(n + 1) * 4 + 16); o->values = values->values ? values->values : alloc_buffer(ctx, 16); o->values_bytes = bytes; o->null_count = 0;
    hipLaunchKernelGGL(k_list_lens, dim3(grid_for(n + 1, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const int64_t*)counts->values->ptr, n, (uint32_t)w, (uint32_t*)o->offsets->ptr, ctx->d_flags);
    exclusive_scan_u32_inplace32(ctx, (uint32_t*)o->offsets->ptr, n + 1, ctx->d_scratch64 + 13);
    KERNEL_CHECK();
    if ((int64_t)read_scratch(ctx, 13) != bytes) fail(DFGPU_INVALID_ARGUMENT, "list_from_counts: the counts add up to %lld bytes, the values hold %lld", (long long)read_scratch(ctx, 13), (long long)bytes);
    check_flags(ctx, "list_from_counts");
    *out = h.release();
  });
}
dfgpu_status dfgpu_list_flatten(dfgpu_ctx* ctx, const dfgpu_array* list, int32_t value_type, int32_t precision, int32_t scale, dfgpu_array** out_values, dfgpu_array** out_row_of) {
  return guard(ctx, [&] {
    if (!list || !out_values || !out_row_of) fail(DFGPU_INVALID_ARGUMENT, "list_flatten: null argument");
    if (list->type != DFGPU_UTF8) fail(DFGPU_INVALID_ARGUMENT, "list_flatten: a list column travels in the Utf8 layout, got type %d", list->type);
    if (value_type == DFGPU_UTF8) {             // the strings of every row back out of their (length, bytes) form
      const int64_t n = list->length;
      ArrayHolder v(new_array(ctx, DFGPU_UTF8, 0)), r(new_fixed(ctx, DFGPU_UINT32, 0));
      if (n) {
        const int32_t* lo = (const int32_t*)list->offsets->ptr; const uint8_t* lb = list->values ? (const uint8_t*)list->values->ptr : nullptr;
        BufferPtr cnt = alloc_buffer(ctx, (size_t)(n + 1) * 4 + 16);
        HIP_CHECK(hipMemsetAsync(cnt->ptr, 0, (size_t)(n + 1) * 4, ctx->stream));
        hipLaunchKernelGGL(k_slist_walk, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, lo, lb, n, (const uint32_t*)nullptr, (uint32_t*)cnt->ptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, ctx->d_flags);
        exclusive_scan_u32_inplace32(ctx, (uint32_t*)cnt->ptr, n + 1, ctx->d_scratch64 + 13);
        KERNEL_CHECK();
        const int64_t nv = (int64_t)read_scratch(ctx, 13);
        check_flags(ctx, "list_flatten (a list row is not a sequence of (length, bytes) strings)");
        if (nv > 0xFFFFFFF0ll) fail(DFGPU_NOT_IMPLEMENTED, "list_flatten: %lld values", (long long)nv);
        r.a = (dfgpu_array_release(r.release()), new_fixed(ctx, DFGPU_UINT32, nv));
        dfgpu_array* o = v.get(); o->length = nv; o->null_count = 0;
        o->offsets = alloc_buffer(ctx, (size_t)(nv + 1) * 4 + 16);
        BufferPtr spos = alloc_buffer(ctx, (size_t)(nv + 1) * 4 + 16);
        HIP_CHECK(hipMemsetAsync(o->offsets->ptr, 0, (size_t)(nv + 1) * 4, ctx->stream));
        hipLaunchKernelGGL(k_slist_walk, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, lo, lb, n, (const uint32_t*)cnt->ptr, (uint32_t*)nullptr, (uint32_t*)o->offsets->ptr, (uint32_t*)spos->ptr, (uint32_t*)r.get()->values->ptr, ctx->d_flags);
        exclusive_scan_u32_inplace32(ctx, (uint32_t*)o->offsets->ptr, nv + 1, ctx->d_scratch64 + 13);
        KERNEL_CHECK();
        const int64_t vb = (int64_t)read_scratch(ctx, 13);
        o->values = alloc_buffer(ctx, (size_t)vb + 16); o->values_bytes = vb;
        if (nv) hipLaunchKernelGGL(k_slist_copy, dim3(grid_for(nv, BLOCK)), dim3(BLOCK), 0, ctx->stream, lb, (const uint32_t*)spos->ptr, (const int32_t*)o->offsets->ptr, nv, (uint8_t*)o->values->ptr);
        KERNEL_CHECK();
      } else { v.get()->offsets = alloc_buffer(ctx, 16, true); v.get()->values = alloc_buffer(ctx, 16); v.get()->null_count = 0; }
      *out_values = v.release(); *out_row_of = r.release();
      return;
    }
    if (value_type == DFGPU_DICTIONARY || value_type == DFGPU_BOOL || value_type < DFGPU_INT8) fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: lists of type %d values on the device", value_type);
    const int w = type_width(value_type); const int64_t n = list->length;
    int32_t ends[2] = {0, 0};
    if (n) { HIP_CHECK(hipMemcpyAsync(&ends[0], list->offsets->ptr, 4, hipMemcpyDeviceToHost, ctx->stream)); HIP_CHECK(hipMemcpyAsync(&ends[1], (const int32_t*)list->offsets->ptr + n, 4, hipMemcpyDeviceToHost, ctx->stream));
      ctx->count_sync("sync:list_flatten"); HIP_CHECK(hipStreamSynchronize(ctx->stream)); }
    const int64_t bytes = (int64_t)ends[1] - ends[0];
    if (bytes < 0 || bytes % w) fail(DFGPU_EXECUTION, "list_flatten: %lld bytes of list values are not a whole number of %d-byte values", (long long)bytes, w);
    const int64_t ne = bytes / w;
    ArrayHolder v(new_fixed(ctx, value_type, ne)), r(new_fixed(ctx, DFGPU_UINT32, ne)); v.get()->precision = precision; v.get()->scale = scale;
    if (ne) { HIP_CHECK(hipMemcpyAsync(v.get()->values->ptr, (const char*)list->values->ptr + ends[0], (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream));
      hipLaunchKernelGGL(k_list_row_of, dim3(grid_for(ne, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const int32_t*)list->offsets->ptr, n, ne, (uint32_t)w, (uint32_t*)r.get()->values->ptr); KERNEL_CHECK(); }
    *out_values = v.release(); *out_row_of = r.release();
  });
}

}  // extern "C"
