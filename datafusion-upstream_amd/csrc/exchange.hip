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
    if (!comm || !out_cols || ncols < 1) fail(DFGPU_INVALID_ARGUMENT, "exchange: null argument");
    constexpr int MAXC = 64;
    const int32_t W = comm->world; if (W > 256) fail(DFGPU_NOT_IMPLEMENTED, "exchange over more than 256 ranks");
    if (ncols > MAXC) fail(DFGPU_NOT_IMPLEMENTED, "exchange of more than %d columns", MAXC);
    const bool have = cols != nullptr && keys != nullptr;          // a rank whose input produced no batch still takes part: it learns the column types from the others
    const int64_t n = have ? keys[0]->length : 0;
    // ---- 1. every column grouped by destination rank in one pass
    std::vector<ArrayHolder> gh((size_t)ncols), vbytes((size_t)ncols); ArrayHolder idx_h; std::vector<int64_t> send_rows((size_t)W, 0);
    if (have) {
      for (int32_t c = 0; c < ncols; c++) {
        const dfgpu_array* a = cols[c];
        if (!a || a->length != n) fail(DFGPU_INVALID_ARGUMENT, "exchange: column %d is missing or differs in length from the keys (%lld rows, type %d; keys hold %lld)", c, a ? (long long)a->length : -1ll, a ? a->type : 0, (long long)n);
        if (a->type == DFGPU_DICTIONARY || a->type == DFGPU_UTF8 || a->type == DFGPU_BOOL || !type_width(a->type))
          fail(DFGPU_NOT_IMPLEMENTED, "exchange of column %d (type %d): fixed-width columns only; cast dictionary / Utf8 columns or use the host-side exchange", c, a->type);
      }
      // a nullable column = its values (validity detached) + the validity as one byte per row, both partitioned like any other column
      std::vector<const dfgpu_array*> pcols; std::vector<ArrayHolder> tmp; std::vector<int> lane_col, lane_is_valid;
      for (int32_t c = 0; c < ncols; c++) {
        const dfgpu_array* a = cols[c];
        if (a->validity) {
          ArrayHolder data(new_array(ctx, a->type, n, a->precision, a->scale)); data.get()->values = a->values; data.get()->null_count = 0;
          ArrayHolder vb(new_fixed(ctx, DFGPU_UINT8, n));
          if (n) hipLaunchKernelGGL(k_bits_to_bytes, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)a->validity->ptr, n, (uint8_t*)vb.get()->values->ptr);
          KERNEL_CHECK();
          pcols.push_back(data.get()); lane_col.push_back(c); lane_is_valid.push_back(0); tmp.push_back(std::move(data));
          pcols.push_back(vb.get()); lane_col.push_back(c); lane_is_valid.push_back(1); tmp.push_back(std::move(vb));
        } else { pcols.push_back(a); lane_col.push_back(c); lane_is_valid.push_back(0); }
      }
      std::vector<dfgpu_array*> grouped(pcols.size(), nullptr); dfgpu_array* idx = nullptr;
      dfgpu_status st = dfgpu_partition_columns(ctx, keys, nkeys, W, pcols.data(), (int32_t)pcols.size(), opt_mask, grouped.data(), &idx, send_rows.data());
      idx_h.a = idx;
      for (size_t l = 0; l < grouped.size(); l++) { if (st == DFGPU_OK && !grouped[l]) st = DFGPU_INTERNAL; (lane_is_valid[l] ? vbytes : gh)[(size_t)lane_col[l]].a = grouped[l]; }
      if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str());
    }
    // ---- 2. what the ranks must agree on, in one small all-gather: the row-count matrix, the column types, which columns are nullable anywhere
    const int64_t ML = W + 1 + 4 * MAXC;
    std::vector<int64_t> mine((size_t)ML, 0), all((size_t)ML * W, 0);
    for (int32_t p = 0; p < W; p++) mine[(size_t)p] = send_rows[(size_t)p];
    mine[(size_t)W] = have ? ncols : 0;
    if (have) for (int32_t c = 0; c < ncols; c++) { const dfgpu_array* a = cols[c]; int64_t* f = &mine[(size_t)W + 1 + 4 * (size_t)c]; f[0] = a->type; f[1] = a->precision; f[2] = a->scale; f[3] = a->validity ? 1 : 0; }
    if (comm->custom) { if (comm->vt.all_gather_host(comm->vt.user, mine.data(), ML * 8, all.data()) != 0) fail(DFGPU_EXECUTION, "exchange: the transport's all_gather_host failed"); }
    else {
      BufferPtr ds = alloc_buffer(ctx, (size_t)ML * 8), dr = alloc_buffer(ctx, (size_t)ML * W * 8);
      HIP_CHECK(hipMemcpyAsync(ds->ptr, mine.data(), (size_t)ML * 8, hipMemcpyHostToDevice, ctx->stream));
      nccl_check(rccl().AllGather(ds->ptr, dr->ptr, (size_t)ML * 8, 1 /* ncclUint8 */, comm->nccl, ctx->stream), "ncclAllGather");
      HIP_CHECK(hipMemcpyAsync(all.data(), dr->ptr, (size_t)ML * W * 8, hipMemcpyDeviceToHost, ctx->stream));
      ctx->count_sync("sync:exchange_counts");
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    const int64_t* ref = nullptr;                      // the first rank that holds a schema
    for (int32_t s2 = 0; s2 < W && !ref; s2++) if (all[(size_t)s2 * ML + W] > 0) ref = &all[(size_t)s2 * ML];
    for (int32_t c = 0; c < ncols; c++) out_cols[c] = nullptr;
    if (out_counts) for (int32_t p = 0; p < 2 * W; p++) out_counts[p] = 0;
    if (!ref) return;                                  // nobody has rows: nothing moves, out_cols stay NULL
    if (ref[W] != ncols) fail(DFGPU_INVALID_ARGUMENT, "exchange: this rank passes %d columns, another one %lld", ncols, (long long)ref[W]);
    std::vector<int64_t> recv_rows((size_t)W, 0); int64_t total = 0, sent = 0;
    for (int32_t s2 = 0; s2 < W; s2++) { recv_rows[(size_t)s2] = all[(size_t)s2 * ML + comm->rank]; total += recv_rows[(size_t)s2]; }
    for (int32_t p = 0; p < W; p++) sent += send_rows[(size_t)p];
    if (total > 0xFFFFFFF0ll) fail(DFGPU_RESOURCES_EXHAUSTED, "exchange: %lld rows arrive at rank %d; more ranks or smaller batches", (long long)total, comm->rank);
    struct Meta { int32_t type, precision, scale; bool nullable; }; std::vector<Meta> meta((size_t)ncols);
    for (int32_t c = 0; c < ncols; c++) {
      const int64_t* f = ref + W + 1 + 4 * c; meta[(size_t)c] = Meta{ (int32_t)f[0], (int32_t)f[1], (int32_t)f[2], false };
      for (int32_t s2 = 0; s2 < W; s2++) { const int64_t* g = &all[(size_t)s2 * ML]; if (g[W] > 0) { if (g[W + 1 + 4 * c] != f[0]) fail(DFGPU_INVALID_ARGUMENT, "exchange: column %d has type %lld on one rank and %lld on another", c, (long long)f[0], (long long)g[W + 1 + 4 * c]); meta[(size_t)c].nullable |= g[W + 4 + 4 * c] != 0; } }
    }
    // ---- 3. one grouped collective over every lane (values of every column, validity bytes of the columns that are nullable anywhere)
    std::vector<ArrayHolder> recv((size_t)ncols), recv_valid((size_t)ncols);
    auto offsets = [&](int64_t w, std::vector<int64_t>& so, std::vector<int64_t>& sb, std::vector<int64_t>& ro, std::vector<int64_t>& rb) {
      so.assign((size_t)W, 0); sb.assign((size_t)W, 0); ro.assign((size_t)W, 0); rb.assign((size_t)W, 0); int64_t a = 0, b = 0;
      for (int32_t p = 0; p < W; p++) { so[(size_t)p] = a * w; sb[(size_t)p] = send_rows[(size_t)p] * w; a += send_rows[(size_t)p]; ro[(size_t)p] = b * w; rb[(size_t)p] = recv_rows[(size_t)p] * w; b += recv_rows[(size_t)p]; }
    };
    { KernelTimer kt_(ctx, "exchange_all_to_all");
      if (comm->custom) HIP_CHECK(hipStreamSynchronize(ctx->stream));            // the callbacks read the send buffers outside this stream
      else nccl_check(rccl().GroupStart(), "ncclGroupStart");
      BufferPtr ones;
      auto move = [&](const uint8_t* sp, uint8_t* rp, int64_t w) {
        std::vector<int64_t> so, sb, ro, rb; offsets(w, so, sb, ro, rb);
        if (comm->custom) { if (comm->vt.all_to_all_v(comm->vt.user, sp, so.data(), sb.data(), rp, ro.data(), rb.data()) != 0) fail(DFGPU_EXECUTION, "exchange: the transport's all_to_all_v failed"); return; }
        for (int32_t p = 0; p < W; p++) {
          if (sb[(size_t)p]) nccl_check(rccl().Send(sp + so[(size_t)p], (size_t)sb[(size_t)p], 1, p, comm->nccl, ctx->stream), "ncclSend");
          if (rb[(size_t)p]) nccl_check(rccl().Recv(rp + ro[(size_t)p], (size_t)rb[(size_t)p], 1, p, comm->nccl, ctx->stream), "ncclRecv");
        }
      };
      for (int32_t c = 0; c < ncols; c++) {
        const Meta& m = meta[(size_t)c]; const int64_t w = type_width(m.type);
        recv[(size_t)c].a = new_fixed(ctx, m.type, total, m.precision, m.scale);
        move(gh[(size_t)c].get() ? (const uint8_t*)gh[(size_t)c].get()->values->ptr : nullptr, (uint8_t*)recv[(size_t)c].get()->values->ptr, w);
        if (!m.nullable) continue;
        recv_valid[(size_t)c].a = new_fixed(ctx, DFGPU_UINT8, total);
        const uint8_t* vp = vbytes[(size_t)c].get() ? (const uint8_t*)vbytes[(size_t)c].get()->values->ptr : nullptr;
        if (!vp && sent) {          // nullable on another rank only: this rank's rows are all valid
          if (!ones) { ones = alloc_buffer(ctx, (size_t)sent); HIP_CHECK(hipMemsetAsync(ones->ptr, 1, (size_t)sent, ctx->stream)); if (comm->custom) HIP_CHECK(hipStreamSynchronize(ctx->stream)); }
          vp = (const uint8_t*)ones->ptr;
        }
        move(vp, (uint8_t*)recv_valid[(size_t)c].get()->values->ptr, 1);
      }
      if (!comm->custom) nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
    }
    // ---- 4. validity bytes -> bitmaps
    for (int32_t c = 0; c < ncols; c++) {
      dfgpu_array* o = recv[(size_t)c].release();
      if (meta[(size_t)c].nullable) {
        o->validity = alloc_buffer(ctx, bitmap_bytes(total), true); o->null_count = -1;
        if (total) hipLaunchKernelGGL(k_bytes_to_bits, dim3(grid_for(total, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint8_t*)recv_valid[(size_t)c].get()->values->ptr, total, (uint64_t*)o->validity->ptr);
        KERNEL_CHECK();
      }
      out_cols[c] = o;
    }
    if (out_counts) for (int32_t p = 0; p < W; p++) { out_counts[p] = send_rows[(size_t)p]; out_counts[W + p] = recv_rows[(size_t)p]; }
  });
}

}  // extern "C"
