// exchange.hip -- the RepartitionExec exchange between the ranks of one job, behind the C ABI (dfgpu_exchange).
//
// Reference: RepartitionExec moves the per-destination slices of every input batch through in-process channels
// (physical-plan/src/repartition/mod.rs:442-580 pull_from_input, :684-760 the per-partition output streams).  With one process per GPU
// the same slices are the send segments of an all-to-all(v): dfgpu_partition_columns groups every column by destination rank in one
// pass (hash % world, the same create_hashes on every rank), the row-count matrix goes first, then ONE grouped collective moves every
// column buffer (RCCL: ncclSend / ncclRecv of all columns and peers inside one ncclGroup -- xGMI is point-to-point, so all 7 links of a
// GPU carry their peer's segment at once).  Received rows are ordered by source rank; row i of every column still belongs to one row.
//
// Transport: RCCL (dfgpu_comm_create_rccl; librccl is looked up at run time, no link dependency) or two caller-provided callbacks
// (dfgpu_comm_create_custom: the CPU tests run the same entry point over torch.distributed's gloo backend, staged through the host).
#include "device_utils.h"
#include <dlfcn.h>

namespace dfgpu {

// the RCCL entry points used (rccl.h: ncclResult_t = int, ncclComm_t = opaque pointer, ncclUniqueId = 128 bytes, ncclUint8 = 1)
struct Id128 { char b[128]; };
struct Rccl {
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, /* ncclUniqueId by value */ Id128, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
static Rccl& rccl() {
  static Rccl r; static bool tried = false;
  if (tried) return r;
  tried = true;
  void* h = nullptr;
  auto sym = [&](const char* n) -> void* { void* p = dlsym(RTLD_DEFAULT, n); if (!p && h) p = dlsym(h, n); return p; };
  if (!dlsym(RTLD_DEFAULT, "ncclCommInitRank")) { h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL); if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL); }
  r.GetUniqueId = (int (*)(void*))sym("ncclGetUniqueId"); r.CommInitRank = (int (*)(void**, int, Id128, int))sym("ncclCommInitRank"); r.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
  r.GroupStart = (int (*)())sym("ncclGroupStart"); r.GroupEnd = (int (*)())sym("ncclGroupEnd");
  r.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend"); r.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
  r.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))sym("ncclAllGather"); r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.AllGather;
  return r;
}
static void nccl_check(int rc, const char* what) { if (rc != 0) fail(DFGPU_EXECUTION, "RCCL %s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "error"); }

// validity bitmap <-> one byte per row (a nullable column's validity travels as a 1-byte column)
__global__ void __launch_bounds__(BLOCK) k_bits_to_bytes(const uint64_t* bits, int64_t n, uint8_t* out) { int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i < n) out[i] = bit_get(bits, i); }
__global__ void __launch_bounds__(BLOCK) k_bytes_to_bits(const uint8_t* in, int64_t n, uint64_t* bits) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; uint64_t m = ballot64(i < n && in[i] != 0);
  if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) bits[i >> 6] = m;
}

__global__ void __launch_bounds__(BLOCK) k_offsets_to_lengths(const int32_t* off, int64_t n, int32_t* len) { int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i < n) len[i] = off[i + 1] - off[i]; }

}  // namespace dfgpu
using namespace dfgpu;

struct dfgpu_comm {
  int32_t rank = 0, world = 1; bool custom = false;
  void* nccl = nullptr;            // ncclComm_t
  dfgpu_comm_vtable vt{};
};

extern "C" {

dfgpu_status dfgpu_comm_unique_id(uint8_t* out_id128) {
  if (!out_id128) return DFGPU_INVALID_ARGUMENT;
  if (!rccl().ok) return DFGPU_NOT_IMPLEMENTED;
  return rccl().GetUniqueId(out_id128) == 0 ? DFGPU_OK : DFGPU_EXECUTION;
}
dfgpu_status dfgpu_comm_create_rccl(dfgpu_ctx* ctx, const uint8_t* id128, int32_t rank, int32_t world, dfgpu_comm** out) {
  return guard(ctx, [&] {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) fail(DFGPU_INVALID_ARGUMENT, "comm_create_rccl: bad argument");
    if (!rccl().ok) fail(DFGPU_NOT_IMPLEMENTED, "librccl.so is not available in this process");
    HIP_CHECK(hipSetDevice(ctx->device));
    std::unique_ptr<dfgpu_comm> c(new dfgpu_comm()); c->rank = rank; c->world = world;
    Id128 id; memcpy(id.b, id128, 128);
    nccl_check(rccl().CommInitRank(&c->nccl, world, id, rank), "ncclCommInitRank");
    *out = c.release();
  });
}
dfgpu_status dfgpu_comm_create_custom(const dfgpu_comm_vtable* vt, dfgpu_comm** out) {
  if (!vt || !out || !vt->all_gather_host || !vt->all_to_all_v || vt->world < 1 || vt->rank < 0 || vt->rank >= vt->world) return DFGPU_INVALID_ARGUMENT;
  dfgpu_comm* c = new dfgpu_comm(); c->custom = true; c->vt = *vt; c->rank = vt->rank; c->world = vt->world; *out = c; return DFGPU_OK;
}
void dfgpu_comm_free(dfgpu_comm* c) { if (!c) return; if (c->nccl && rccl().ok) (void)rccl().CommDestroy(c->nccl); delete c; }
int32_t dfgpu_comm_rank(const dfgpu_comm* c) { return c ? c->rank : -1; }
int32_t dfgpu_comm_world(const dfgpu_comm* c) { return c ? c->world : 0; }

dfgpu_status dfgpu_exchange(dfgpu_ctx* ctx, dfgpu_comm* comm, const dfgpu_array* const* keys, int32_t nkeys, const dfgpu_array* const* cols, int32_t ncols,
                            const dfgpu_array* opt_mask, dfgpu_array** out_cols, int64_t* out_counts /* [2 * world]: rows sent to / received from every rank; may be NULL */) {
  return guard(ctx, [&] {
    if (!comm || !out_cols) fail(DFGPU_INVALID_ARGUMENT, "exchange: null argument");
    constexpr int MAXC = 64;
    const int32_t W = comm->world; if (W > 256) fail(DFGPU_NOT_IMPLEMENTED, "exchange over more than 256 ranks");       // the same on every rank of the communicator
    // A bad column count is this rank's failure alone: like every failure before the collective it travels in the status word (the metadata layout below is
    // sized by MAXC, not by the local ncols, so the all-gather itself has one length on every rank whatever each rank was handed).
    const int32_t ncols_arg = ncols; if (ncols < 1 || ncols > MAXC) ncols = 0;
    const bool have = cols != nullptr && keys != nullptr;          // a rank whose input produced no batch still takes part: it learns the column types from the others
    const int64_t n = have ? keys[0]->length : 0;
    // ---- 1. every column grouped by destination rank: fixed-width columns without NULLs in the partition pass itself, every other column (NULLs, Utf8,
    // Boolean, dictionary) by a take through the pass's row numbers.  A failure here must not keep this rank away from the collective below (its peers
    // would wait in it for ever): it is caught, travels as a status word, and every rank fails together.
    std::vector<ArrayHolder> gh((size_t)ncols); std::vector<int64_t> send_rows((size_t)W, 0);
    std::vector<std::vector<int64_t>> ubytes((size_t)ncols);                       // Utf8 columns: value bytes per destination
    std::vector<BufferPtr> ulen((size_t)ncols), vbytes((size_t)ncols), bbytes((size_t)ncols);      // Utf8 lengths (int32 per row), validity / Boolean values as one byte per row
    int64_t local_status = DFGPU_OK; std::string local_err;
    auto lane_type = [](const dfgpu_array* a) { return a->type == DFGPU_DICTIONARY ? a->dictionary->type : a->type; };
    try {
      if (ncols_arg < 1 || ncols_arg > MAXC) fail(ncols_arg < 1 ? DFGPU_INVALID_ARGUMENT : DFGPU_NOT_IMPLEMENTED, "exchange of %d columns (1 .. %d)", ncols_arg, MAXC);
      if (have) {
        std::vector<ArrayHolder> plain((size_t)ncols);             // dictionary columns travel as their values
        std::vector<const dfgpu_array*> src((size_t)ncols);
        for (int32_t c = 0; c < ncols; c++) {
          const dfgpu_array* a = cols[c];
          if (!a || a->length != n) fail(DFGPU_INVALID_ARGUMENT, "exchange: column %d is missing or differs in length from the keys (%lld rows, type %d; keys hold %lld)", c, a ? (long long)a->length : -1ll, a ? a->type : 0, (long long)n);
          const int32_t lt = lane_type(a);
          if (lt != DFGPU_UTF8 && lt != DFGPU_BOOL && !type_width(lt)) fail(DFGPU_NOT_IMPLEMENTED, "exchange of column %d (type %d)", c, lt);
          if (a->type == DFGPU_DICTIONARY) {        // decode: take(dictionary values, keys) -- a NULL key or a NULL dictionary entry is a NULL cell
            ArrayHolder kv(new_array(ctx, a->key_type, n)); kv.get()->values = a->values; kv.get()->validity = a->validity; kv.get()->null_count = a->validity ? -1 : 0;
            const dfgpu_array* kidx = kv.get(); ArrayHolder k32;
            if (type_width(a->key_type) < 4) { dfgpu_array* w = nullptr; dfgpu_status st = dfgpu_cast(ctx, kv.get(), DFGPU_INT32, 0, 0, &w); if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str()); k32.a = w; kidx = w; }
            dfgpu_array* v = nullptr; dfgpu_status st = dfgpu_take(ctx, a->dictionary, kidx, &v); if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str());
            plain[(size_t)c].a = v; a = v;
          }
          src[(size_t)c] = a;
        }
        std::vector<dfgpu_array*> grouped((size_t)ncols, nullptr); dfgpu_array* idx = nullptr;
        dfgpu_status st = dfgpu_partition_columns(ctx, keys, nkeys, W, src.data(), ncols, opt_mask, grouped.data(), &idx, send_rows.data());
        ArrayHolder idx_h(idx);
        for (int32_t c = 0; c < ncols; c++) gh[(size_t)c].a = grouped[(size_t)c];
        if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str());
        int64_t sent = 0; for (int32_t p = 0; p < W; p++) sent += send_rows[(size_t)p];
        std::vector<int> utf;
        for (int32_t c = 0; c < ncols; c++) {
          if (!gh[(size_t)c].get()) { dfgpu_array* g = nullptr; st = dfgpu_take(ctx, src[(size_t)c], idx, &g); if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str()); gh[(size_t)c].a = g; }
          const dfgpu_array* g = gh[(size_t)c].get();
          if (g->validity) { vbytes[(size_t)c] = alloc_buffer(ctx, (size_t)sent + 1); if (sent) hipLaunchKernelGGL(k_bits_to_bytes, dim3(grid_for(sent, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)g->validity->ptr, sent, (uint8_t*)vbytes[(size_t)c]->ptr); }
          if (g->type == DFGPU_BOOL) { bbytes[(size_t)c] = alloc_buffer(ctx, (size_t)sent + 1); if (sent) hipLaunchKernelGGL(k_bits_to_bytes, dim3(grid_for(sent, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)g->values->ptr, sent, (uint8_t*)bbytes[(size_t)c]->ptr); }
          if (g->type == DFGPU_UTF8) { ulen[(size_t)c] = alloc_buffer(ctx, (size_t)(sent + 1) * 4); if (sent) hipLaunchKernelGGL(k_offsets_to_lengths, dim3(grid_for(sent, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const int32_t*)g->offsets->ptr, sent, (int32_t*)ulen[(size_t)c]->ptr); utf.push_back(c); }
          KERNEL_CHECK();
        }
        if (!utf.empty()) {        // value bytes per destination = differences of the grouped offsets at the destination bounds: one small read-back for all Utf8 columns
          std::vector<int32_t> bounds(utf.size() * (size_t)(W + 1), 0);
          if (sent) {
            for (size_t u = 0; u < utf.size(); u++) { int64_t row = 0; const int32_t* off = (const int32_t*)gh[(size_t)utf[u]].get()->offsets->ptr;
              for (int32_t p = 0; p <= W; p++) { HIP_CHECK(hipMemcpyAsync(&bounds[u * (size_t)(W + 1) + (size_t)p], off + row, 4, hipMemcpyDeviceToHost, ctx->stream)); if (p < W) row += send_rows[(size_t)p]; } }
            ctx->count_sync("sync:exchange_utf8_bounds");
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
          }
          for (size_t u = 0; u < utf.size(); u++) { auto& ub = ubytes[(size_t)utf[u]]; ub.assign((size_t)W, 0); for (int32_t p = 0; p < W; p++) ub[(size_t)p] = (int64_t)bounds[u * (size_t)(W + 1) + (size_t)p + 1] - bounds[u * (size_t)(W + 1) + (size_t)p]; }
        }
      }
    } catch (const Error& e) { local_status = e.code ? e.code : DFGPU_INTERNAL; local_err = e.msg; std::fill(send_rows.begin(), send_rows.end(), 0); }
    // ---- 2. what the ranks must agree on, in one small all-gather: the status word, the row-count matrix, the column types, which columns are nullable
    // anywhere, the Utf8 byte counts
    const int64_t ML = W + 2 + 4 * (int64_t)MAXC + (int64_t)MAXC * W;
    std::vector<int64_t> mine((size_t)ML, 0), all((size_t)ML * W, 0);
    auto gather = [&](std::vector<int64_t>& m, std::vector<int64_t>& a, int64_t len, const char* what) {
      if (comm->custom) { if (comm->vt.all_gather_host(comm->vt.user, m.data(), len * 8, a.data()) != 0) fail(DFGPU_EXECUTION, "exchange: the transport's all_gather_host failed (%s)", what); return; }
      BufferPtr ds = alloc_buffer(ctx, (size_t)len * 8), dr = alloc_buffer(ctx, (size_t)len * W * 8);
      HIP_CHECK(hipMemcpyAsync(ds->ptr, m.data(), (size_t)len * 8, hipMemcpyHostToDevice, ctx->stream));
      nccl_check(rccl().AllGather(ds->ptr, dr->ptr, (size_t)len * 8, 1 /* ncclUint8 */, comm->nccl, ctx->stream), "ncclAllGather");
      HIP_CHECK(hipMemcpyAsync(a.data(), dr->ptr, (size_t)len * W * 8, hipMemcpyDeviceToHost, ctx->stream));
      ctx->count_sync("sync:exchange_counts");
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
    };
    const size_t O_NCOLS = (size_t)W, O_STATUS = (size_t)W + 1, O_META = (size_t)W + 2, O_UB = O_META + 4 * (size_t)MAXC;
    for (int32_t p = 0; p < W; p++) mine[(size_t)p] = send_rows[(size_t)p];
    mine[O_NCOLS] = have && local_status == DFGPU_OK ? ncols : 0; mine[O_STATUS] = local_status;
    if (have && local_status == DFGPU_OK) for (int32_t c = 0; c < ncols; c++) {
      const dfgpu_array* g = gh[(size_t)c].get(); int64_t* f = &mine[O_META + 4 * (size_t)c]; f[0] = g->type; f[1] = g->precision; f[2] = g->scale; f[3] = g->validity ? 1 : 0;
      if (!ubytes[(size_t)c].empty()) for (int32_t p = 0; p < W; p++) mine[O_UB + (size_t)c * W + (size_t)p] = ubytes[(size_t)c][(size_t)p];
    }
    gather(mine, all, ML, "row counts");
    for (int32_t c = 0; c < ncols; c++) out_cols[c] = nullptr;
    if (out_counts) for (int32_t p = 0; p < 2 * W; p++) out_counts[p] = 0;
    for (int32_t s2 = 0; s2 < W; s2++) if (all[(size_t)s2 * ML + O_STATUS] != DFGPU_OK) {        // every rank leaves here, none enters the data collective
      if (s2 == comm->rank) fail((dfgpu_status)local_status, "%s", local_err.c_str());
      fail((dfgpu_status)all[(size_t)s2 * ML + O_STATUS], "exchange: rank %d failed before the collective (status %lld); nothing was exchanged", s2, (long long)all[(size_t)s2 * ML + O_STATUS]);
    }
    const int64_t* ref = nullptr;                      // the first rank that holds a schema
    for (int32_t s2 = 0; s2 < W && !ref; s2++) if (all[(size_t)s2 * ML + O_NCOLS] > 0) ref = &all[(size_t)s2 * ML];
    if (!ref) return;                                  // nobody has rows: nothing moves, out_cols stay NULL
    // every rank reads the same matrix, so every check below gives the same verdict on every rank: all fail together, none enters the data collective alone
    for (int32_t s2 = 0; s2 < W; s2++) { const int64_t nc2 = all[(size_t)s2 * ML + O_NCOLS]; if (nc2 > 0 && nc2 != ref[O_NCOLS]) fail(DFGPU_INVALID_ARGUMENT, "exchange: rank %d passes %lld columns, another one %lld", s2, (long long)nc2, (long long)ref[O_NCOLS]); }
    if (ref[O_NCOLS] != ncols) fail(DFGPU_INVALID_ARGUMENT, "exchange: this rank passes %d columns, the ranks with rows %lld", ncols, (long long)ref[O_NCOLS]);      // a rank without rows (cols == NULL) and another count
    std::vector<int64_t> recv_rows((size_t)W, 0); int64_t total = 0, sent = 0;
    for (int32_t s2 = 0; s2 < W; s2++) { recv_rows[(size_t)s2] = all[(size_t)s2 * ML + comm->rank]; total += recv_rows[(size_t)s2]; }
    for (int32_t p = 0; p < W; p++) sent += send_rows[(size_t)p];
    struct Meta { int32_t type, precision, scale; bool nullable; }; std::vector<Meta> meta((size_t)ncols);
    int64_t agree = total > 0xFFFFFFF0ll ? DFGPU_RESOURCES_EXHAUSTED : DFGPU_OK;           // every rank checks every rank's view: the same verdict everywhere
    for (int32_t r2 = 0; r2 < W; r2++) { int64_t t2 = 0; for (int32_t s2 = 0; s2 < W; s2++) t2 += all[(size_t)s2 * ML + (size_t)r2]; if (t2 > 0xFFFFFFF0ll) agree = DFGPU_RESOURCES_EXHAUSTED; }
    if (agree != DFGPU_OK) fail(DFGPU_RESOURCES_EXHAUSTED, "exchange: more than 2^32-16 rows arrive at one rank; more ranks or smaller batches");
    for (int32_t c = 0; c < ncols; c++) {
      const int64_t* f = ref + O_META + 4 * (size_t)c; meta[(size_t)c] = Meta{ (int32_t)f[0], (int32_t)f[1], (int32_t)f[2], false };
      for (int32_t s2 = 0; s2 < W; s2++) { const int64_t* g = &all[(size_t)s2 * ML]; if (g[O_NCOLS] > 0) { if (g[O_META + 4 * (size_t)c] != f[0]) fail(DFGPU_INVALID_ARGUMENT, "exchange: column %d has type %lld on one rank and %lld on another", c, (long long)f[0], (long long)g[O_META + 4 * (size_t)c]); meta[(size_t)c].nullable |= g[O_META + 4 * (size_t)c + 3] != 0; } }
      if (meta[(size_t)c].type == DFGPU_UTF8) for (int32_t r2 = 0; r2 < W; r2++) {          // Utf8's 32-bit offsets at EVERY receiver, checked by every rank
        int64_t b2 = 0; for (int32_t s2 = 0; s2 < W; s2++) b2 += all[(size_t)s2 * ML + O_UB + (size_t)c * W + (size_t)r2];
        if (b2 > 0x7FFFFFF0ll) fail(DFGPU_RESOURCES_EXHAUSTED, "exchange: %lld bytes of column %d arrive at rank %d: beyond Utf8's 32-bit offsets", (long long)b2, c, r2);
      }
    }
    // ---- 3. the lanes: per column its values (fixed width), or its Boolean bytes, or its Utf8 lengths + value bytes; plus validity bytes where nullable anywhere.
    // Every receive buffer exists BEFORE the grouped collective starts; a rank that cannot allocate says so in a second status round -- always run: an
    // allocation can fail without a memory limit too, and a per-ctx option may differ between ranks -- so that no rank enters the group alone.
    struct Lane { const uint8_t* sp; uint8_t* rp; std::vector<int64_t> sb, rb; };
    std::vector<Lane> lanes; std::vector<BufferPtr> keep;
    std::vector<BufferPtr> r_vals((size_t)ncols), r_len((size_t)ncols), r_valid((size_t)ncols), o_bits((size_t)ncols), o_vbits((size_t)ncols); std::vector<int64_t> r_bytes((size_t)ncols, 0);      // o_bits / o_vbits: the Boolean values / validity bitmaps of the output arrays (every allocation of the call happens before the status round)
    BufferPtr ones; int64_t alloc_status = DFGPU_OK; std::string alloc_err;
    auto fixed_lane = [&](const void* sp, void* rp, int64_t w) { Lane l{ (const uint8_t*)sp, (uint8_t*)rp, std::vector<int64_t>((size_t)W), std::vector<int64_t>((size_t)W) }; for (int32_t p = 0; p < W; p++) { l.sb[(size_t)p] = send_rows[(size_t)p] * w; l.rb[(size_t)p] = recv_rows[(size_t)p] * w; } lanes.push_back(std::move(l)); };
    try {
      for (int32_t c = 0; c < ncols; c++) {
        const Meta& m = meta[(size_t)c]; const dfgpu_array* g = gh[(size_t)c].get();
        if (m.type == DFGPU_UTF8) {
          r_len[(size_t)c] = alloc_buffer(ctx, (size_t)(total + 1) * 4);
          fixed_lane(g ? ulen[(size_t)c]->ptr : nullptr, r_len[(size_t)c]->ptr, 4);
          Lane l{ g ? (const uint8_t*)g->values->ptr : nullptr, nullptr, std::vector<int64_t>((size_t)W, 0), std::vector<int64_t>((size_t)W, 0) };
          for (int32_t p = 0; p < W; p++) { l.sb[(size_t)p] = ubytes[(size_t)c].empty() ? 0 : ubytes[(size_t)c][(size_t)p]; l.rb[(size_t)p] = all[(size_t)p * ML + O_UB + (size_t)c * W + (size_t)comm->rank]; r_bytes[(size_t)c] += l.rb[(size_t)p]; }
          r_vals[(size_t)c] = alloc_buffer(ctx, (size_t)r_bytes[(size_t)c] + 8); l.rp = (uint8_t*)r_vals[(size_t)c]->ptr;
          lanes.push_back(std::move(l));
        } else if (m.type == DFGPU_BOOL) {
          r_vals[(size_t)c] = alloc_buffer(ctx, (size_t)total + 1); o_bits[(size_t)c] = alloc_buffer(ctx, bitmap_bytes(total), true);
          fixed_lane(g ? bbytes[(size_t)c]->ptr : nullptr, r_vals[(size_t)c]->ptr, 1);
        } else {
          const int64_t w = type_width(m.type);
          r_vals[(size_t)c] = alloc_buffer(ctx, (size_t)total * w + 8);
          fixed_lane(g ? g->values->ptr : nullptr, r_vals[(size_t)c]->ptr, w);
        }
        if (!m.nullable) continue;
        r_valid[(size_t)c] = alloc_buffer(ctx, (size_t)total + 1); o_vbits[(size_t)c] = alloc_buffer(ctx, bitmap_bytes(total), true);
        const void* vp = vbytes[(size_t)c] ? vbytes[(size_t)c]->ptr : nullptr;
        if (!vp && sent) {          // nullable on another rank only: this rank's rows are all valid
          if (!ones) { ones = alloc_buffer(ctx, (size_t)sent); HIP_CHECK(hipMemsetAsync(ones->ptr, 1, (size_t)sent, ctx->stream)); }
          vp = ones->ptr;
        }
        fixed_lane(vp, r_valid[(size_t)c]->ptr, 1);
      }
    } catch (const Error& e) { alloc_status = e.code ? e.code : DFGPU_INTERNAL; alloc_err = e.msg; }
    {
      std::vector<int64_t> m2(1, alloc_status), a2((size_t)W, 0);
      gather(m2, a2, 1, "allocation status");
      for (int32_t s2 = 0; s2 < W; s2++) if (a2[(size_t)s2] != DFGPU_OK) {
        if (s2 == comm->rank) fail((dfgpu_status)alloc_status, "%s", alloc_err.c_str());
        fail((dfgpu_status)a2[(size_t)s2], "exchange: rank %d could not allocate its receive buffers (status %lld); nothing was exchanged", s2, (long long)a2[(size_t)s2]);
      }
    }
    // ---- 4. one grouped collective over every lane
    { KernelTimer kt_(ctx, "exchange_all_to_all");
      struct Group { bool open = false; ~Group() { if (open) (void)rccl().GroupEnd(); } } group;       // an error between Start and End still closes the group: the communicator stays usable
      if (comm->custom) HIP_CHECK(hipStreamSynchronize(ctx->stream));            // the callbacks read the send buffers outside this stream
      else { nccl_check(rccl().GroupStart(), "ncclGroupStart"); group.open = true; }
      for (size_t li = 0; li < lanes.size(); li++) {
        const Lane& l = lanes[li];
        std::vector<int64_t> so((size_t)W, 0), ro((size_t)W, 0); int64_t a = 0, b2 = 0;
        for (int32_t p = 0; p < W; p++) { so[(size_t)p] = a; a += l.sb[(size_t)p]; ro[(size_t)p] = b2; b2 += l.rb[(size_t)p]; }
        if (comm->custom) { if (comm->vt.all_to_all_v(comm->vt.user, l.sp, so.data(), l.sb.data(), l.rp, ro.data(), l.rb.data()) != 0) fail(DFGPU_EXECUTION, "exchange: the transport's all_to_all_v failed on lane %zu", li); continue; }
        for (int32_t p = 0; p < W; p++) {
          if (l.sb[(size_t)p]) nccl_check(rccl().Send(l.sp + so[(size_t)p], (size_t)l.sb[(size_t)p], 1, p, comm->nccl, ctx->stream), "ncclSend");
          if (l.rb[(size_t)p]) nccl_check(rccl().Recv(l.rp + ro[(size_t)p], (size_t)l.rb[(size_t)p], 1, p, comm->nccl, ctx->stream), "ncclRecv");
        }
      }
      if (group.open) { group.open = false; nccl_check(rccl().GroupEnd(), "ncclGroupEnd"); }
    }
    // ---- 5. the received lanes become arrays: validity bytes -> bitmaps, Boolean bytes -> bits, Utf8 lengths -> offsets
    for (int32_t c = 0; c < ncols; c++) {
      const Meta& m = meta[(size_t)c];
      ArrayHolder o(new_array(ctx, m.type, total, m.precision, m.scale));
      if (m.type == DFGPU_UTF8) {
        o.get()->values = r_vals[(size_t)c]; o.get()->values_bytes = r_bytes[(size_t)c];
        HIP_CHECK(hipMemsetAsync((uint8_t*)r_len[(size_t)c]->ptr + (size_t)total * 4, 0, 4, ctx->stream));
        exclusive_scan_u32_inplace32(ctx, (uint32_t*)r_len[(size_t)c]->ptr, total + 1, nullptr);
        o.get()->offsets = r_len[(size_t)c];
      } else if (m.type == DFGPU_BOOL) {
        o.get()->values = o_bits[(size_t)c];
        if (total) hipLaunchKernelGGL(k_bytes_to_bits, dim3(grid_for(total, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint8_t*)r_vals[(size_t)c]->ptr, total, (uint64_t*)o.get()->values->ptr);
      } else o.get()->values = r_vals[(size_t)c];
      o.get()->null_count = 0;
      if (m.nullable) {
        o.get()->validity = o_vbits[(size_t)c]; o.get()->null_count = -1;
        if (total) hipLaunchKernelGGL(k_bytes_to_bits, dim3(grid_for(total, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint8_t*)r_valid[(size_t)c]->ptr, total, (uint64_t*)o.get()->validity->ptr);
      }
      KERNEL_CHECK();
      out_cols[c] = o.release();
    }
    if (out_counts) for (int32_t p = 0; p < W; p++) { out_counts[p] = send_rows[(size_t)p]; out_counts[W + p] = recv_rows[(size_t)p]; }
  });
}

}  // extern "C"
