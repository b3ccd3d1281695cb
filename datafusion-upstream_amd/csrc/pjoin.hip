// pjoin.hip -- radix-partitioned hash join for unsorted / sparse integer keys: build + probe of every partition out of LDS.
//
// Reference semantics: JoinHashMap build + lookup_join_hashmap (physical-plan/src/joins/utils.rs:121-229, hash_join.rs:1024-1118);
// the (build, probe) pairs come back ordered by probe row exactly as the reference emits them.
//
// Why: the global open-addressing table of join.hip costs every probe row a random 64-B sector out of a table far larger than L2
// (150 M probes of a 15 M-row build: ~53 G sectors/s on MI355X whatever the table's size beyond L2, i.e. >= 2.8 ms, measured 4.5 ms with
// the key verification; profiles/experiments/random_gather_microbench.hip).  Here both sides are split by hash bits (radix_partition.h:
// one pass, LDS-staged write combining) into P partitions of <= 16 K build rows, so that one partition's table -- 32 K four-byte slots
// (tag : 18 | partition-local build row : 14) -- sits in the 160 KB LDS of a CU.  One workgroup per partition builds the table from the
// partition's (key, row) records and streams the partition's probe rows against it; a tag match is verified against the build record
// (exact: full 64-bit key compare), which for a hit also yields the original build row.
//
// Order: partitioning destroys probe order.  Scattering hits back by probe row (found[row] = build row) was measured at 1.1 ms for 30 M hits:
// every 4-byte store lands in a different DRAM sector (read-modify-write behind ECC).  Instead the hits are written out contiguously per
// partition as (probe row, build row) records, re-partitioned by probe-row RANGE (the same radix_partition pass, bucket = row >> shift,
// so bucket order is row order), and every bucket is put in row order inside LDS: a bitmap of the bucket's rows + per-word ranks gives
// each hit its output slot.  Output order is a function of the input only.
// Build keys must be unique (every PK-FK join); a repeated build key, a partition beyond the table's capacity (adversarial hash skew) or an
// unsupported key type leave the table to join.hip's general path.
#include "join_table.h"
#include "radix_partition.h"

namespace dfgpu {

constexpr uint32_t PJ_EMPTY = 0xFFFFFFFFu;
constexpr int PJ_IDX_BITS = 14;                       // partition-local build row
constexpr uint32_t PJ_IDX_MASK = (1u << PJ_IDX_BITS) - 1u;
constexpr uint32_t PJ_TAG_MASK = (1u << (32 - PJ_IDX_BITS)) - 1u;
constexpr uint32_t PJ_MAX_PART_ROWS = PJ_IDX_MASK - 1;      // keeps (tag, idx) != PJ_EMPTY
constexpr int PJ_MAX_SBITS = 15;                      // 32 K slots x 4 B = 128 KB
constexpr int PJ_NT = 1024;                           // one workgroup per CU while a 128 KB table is resident
constexpr int PJ_U = 4;                               // probe rows per lane in flight

// ---- build side check: one workgroup per partition inserts the partition's keys into the LDS table exactly as the probe kernel will;
// reports a repeated key ([1]) or an over-full partition ([2]) and the largest partition ([0])
// Slots are read four at a time (one ds_read_b128 per step of a walk): a key's home is the first slot of the 4-slot group its hash selects and
// it sits in the first free slot from there on, so a lookup that walks group by group may stop at the first EMPTY it sees.
__device__ inline void pj_build_table(uint32_t* tab, uint32_t M, const uint64_t* bkey, uint32_t nb, unsigned long long* flags) {
  for (uint32_t j0 = threadIdx.x; j0 < nb; j0 += PJ_NT * 4) {
    uint64_t k[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { uint32_t j = j0 + u * PJ_NT; k[u] = bkey[j < nb ? j : nb - 1]; }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      uint32_t j = j0 + u * PJ_NT; if (j >= nb) break;
      uint64_t h = mix64(k[u]); uint32_t s = (uint32_t)h & M & ~3u, tag = (uint32_t)(h >> PJ_MAX_SBITS) & PJ_TAG_MASK, ent = (tag << PJ_IDX_BITS) | j;
      for (;;) {
        uint32_t old = atomicCAS(&tab[s], PJ_EMPTY, ent);
        if (old == PJ_EMPTY) break;
        if ((old >> PJ_IDX_BITS) == tag && bkey[old & PJ_IDX_MASK] == k[u]) { if (flags) flags[1] = 1ull; break; }     // the key is in the table already
        s = (s + 1) & M;
      }
    }
  }
}
__global__ void __launch_bounds__(PJ_NT) k_pj_check(const uint64_t* bkey, const uint32_t* bstart, unsigned long long* flags) {
  extern __shared__ uint32_t pj_tab[];
  const uint32_t b0 = bstart[blockIdx.x], nb = bstart[blockIdx.x + 1] - b0;
  if (threadIdx.x == 0) atomicMax(&flags[0], (unsigned long long)nb);
  if (nb > PJ_MAX_PART_ROWS) { if (threadIdx.x == 0) flags[2] = 1ull; return; }
  int sbits = 6; while ((1u << sbits) < 2 * nb && sbits < PJ_MAX_SBITS) sbits++;
  if ((1u << sbits) < nb + nb / 8 + 4) { if (threadIdx.x == 0) flags[2] = 1ull; return; }
  const uint32_t S = 1u << sbits;
  for (uint32_t s = threadIdx.x; s < S; s += PJ_NT) pj_tab[s] = PJ_EMPTY;
  __syncthreads();
  pj_build_table(pj_tab, S - 1, bkey + b0, nb, flags);
}

// ---- probe: one workgroup per partition.  Hits leave as u64 records (probe row | build row << 32), contiguous from the partition's first
// probe slot (a probe row matches at most once: unique build keys); mcount[p] = hits of partition p.
__global__ void __launch_bounds__(PJ_NT) k_pj_join(const uint64_t* bkey, const uint32_t* brow, const uint32_t* bstart, const uint64_t* pkey, const uint32_t* prow, const uint32_t* pstart,
                                                 int sbits, uint64_t* mraw, uint32_t* mcount) {
  extern __shared__ uint32_t pj_tab[];
  const uint32_t S = 1u << sbits, M = S - 1;
  uint32_t* const cursor = pj_tab + S;                 // hits emitted so far (behind the table: keeps the table 16-byte aligned)
  const int p = blockIdx.x, lane = lane_id();
  const uint32_t q0 = pstart[p], q1 = pstart[p + 1];
  if (q0 == q1) { if (threadIdx.x == 0) mcount[p] = 0; return; }
  for (uint32_t s = threadIdx.x; s < S; s += PJ_NT) pj_tab[s] = PJ_EMPTY;
  if (threadIdx.x == 0) *cursor = 0;
  __syncthreads();
  const uint32_t b0 = bstart[p], nb = bstart[p + 1] - b0;
  const uint64_t* bk = bkey + b0; const uint32_t* br = brow + b0;
  pj_build_table(pj_tab, M, bk, nb, nullptr);
  __syncthreads();
  uint64_t* const mout = mraw + q0;
  // PJ_U rows per lane: keys and row ids of the NEXT step are loaded before the current step walks the table, so the walk (LDS only)
  // and the verification round trip of the hits (one L2 load per hit, all issued together) hide behind them
  uint64_t k[PJ_U], kn[PJ_U]; uint32_t r[PJ_U], rn[PJ_U];
  uint32_t i0 = q0 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < PJ_U; u++) { uint32_t i = i0 + u * PJ_NT; uint32_t ic = i < q1 ? i : q1 - 1; kn[u] = pkey[ic]; rn[u] = prow[ic]; }
  for (; i0 - threadIdx.x < q1; i0 += PJ_NT * PJ_U) {          // wave-uniform trip count: the ballots below need every lane
    uint32_t cand[PJ_U], pos[PJ_U]; bool on[PJ_U];
#pragma unroll
    for (int u = 0; u < PJ_U; u++) { k[u] = kn[u]; r[u] = rn[u]; on[u] = i0 + u * PJ_NT < q1; }
    const uint32_t inext = i0 + PJ_NT * PJ_U;
#pragma unroll
    for (int u = 0; u < PJ_U; u++) { uint32_t i = inext + u * PJ_NT; uint32_t ic = i < q1 ? i : q1 - 1; kn[u] = pkey[ic]; rn[u] = prow[ic]; }
    uint32_t s[PJ_U], tag[PJ_U]; bool walking[PJ_U]; bool any = false;
#pragma unroll
    for (int u = 0; u < PJ_U; u++) {
      uint64_t h = mix64(k[u]); s[u] = (uint32_t)h & M & ~3u; tag[u] = (uint32_t)(h >> PJ_MAX_SBITS) & PJ_TAG_MASK;
      walking[u] = on[u]; cand[u] = PJ_EMPTY; pos[u] = 0; any |= on[u];
    }
    while (__ballot(any)) {           // the PJ_U walks of a lane advance together: one 16-byte LDS read each per step
      any = false;
#pragma unroll
      for (int u = 0; u < PJ_U; u++) if (walking[u]) {
        const uint4 v = *(const uint4*)&pj_tab[s[u]];
        const uint32_t c4[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int j = 0; j < 4; j++) if (walking[u]) {
          if (c4[j] == PJ_EMPTY) walking[u] = false;
          else if ((c4[j] >> PJ_IDX_BITS) == tag[u]) { cand[u] = c4[j]; pos[u] = s[u] + j; walking[u] = false; }
        }
        s[u] = (s[u] + 4) & M; any |= walking[u];
      }
    }
    uint64_t vk[PJ_U]; uint32_t vr[PJ_U]; bool hit[PJ_U];
#pragma unroll
    for (int u = 0; u < PJ_U; u++) { vk[u] = 0; vr[u] = 0; if (cand[u] != PJ_EMPTY) { uint32_t j = cand[u] & PJ_IDX_MASK; vk[u] = bk[j]; vr[u] = br[j]; } }
#pragma unroll
    for (int u = 0; u < PJ_U; u++) {
      hit[u] = cand[u] != PJ_EMPTY && vk[u] == k[u];
      if (cand[u] == PJ_EMPTY || hit[u]) continue;
      // a different key with the same 18-bit tag (2^-18 per occupied slot passed): keep walking, verifying every tag match
      uint32_t s1 = (pos[u] + 1) & M, c = pj_tab[s1];
      while (c != PJ_EMPTY) {
        if ((c >> PJ_IDX_BITS) == tag[u] && bk[c & PJ_IDX_MASK] == k[u]) { vr[u] = br[c & PJ_IDX_MASK]; hit[u] = true; break; }
        s1 = (s1 + 1) & M; c = pj_tab[s1];
      }
    }
    uint64_t bal[PJ_U]; uint32_t tot = 0;
#pragma unroll
    for (int u = 0; u < PJ_U; u++) { bal[u] = ballot64(hit[u]); tot += (uint32_t)__popcll(bal[u]); }
    if (tot) {
      uint32_t base = 0; if (lane == 0) base = atomicAdd(cursor, tot);
      base = __shfl(base, 0, 64);
#pragma unroll
      for (int u = 0; u < PJ_U; u++) { if (hit[u]) mout[base + (uint32_t)__popcll(bal[u] & lanemask_lt())] = (uint64_t)r[u] | ((uint64_t)vr[u] << 32); base += (uint32_t)__popcll(bal[u]); }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) mcount[p] = *cursor;
}
// hits of partition p: mraw[pstart[p] .. + mcount[p]) -> mcompact[moff[p] ..)
__global__ void __launch_bounds__(BLOCK) k_pj_mcopy(const uint64_t* mraw, const uint32_t* pstart, const uint32_t* mcount, const uint64_t* moff, uint64_t* mcompact) {
  const int p = blockIdx.x; const uint32_t m = mcount[p]; const uint64_t* src = mraw + pstart[p]; uint64_t* dst = mcompact + moff[p];
  for (uint32_t i = threadIdx.x; i < m; i += BLOCK) dst[i] = src[i];
}
// ---- row order inside one bucket of 2^shift probe rows: bitmap of the rows that hit + ranks
__global__ void __launch_bounds__(PJ_NT) k_pj_order(const uint64_t* mb, const uint32_t* starts, int shift, uint32_t* out_probe, uint64_t* out_build) {
  extern __shared__ uint32_t pj_tab[];
  const uint32_t nw = 1u << (shift - 6);
  unsigned long long* bits = (unsigned long long*)pj_tab; uint32_t* pre = pj_tab + 2 * nw;
  __shared__ uint32_t wsum[PJ_NT / WAVE];
  const uint32_t b = blockIdx.x, m0 = starts[b], m1 = starts[b + 1], r0 = b << shift;
  if (m0 == m1) return;
  for (uint32_t w = threadIdx.x; w < nw; w += PJ_NT) bits[w] = 0ull;
  __syncthreads();
  for (uint32_t i = m0 + threadIdx.x; i < m1; i += PJ_NT) { uint32_t d = (uint32_t)(mb[i] & 0xFFFFFFFFull) - r0; atomicOr(&bits[d >> 6], 1ull << (d & 63)); }
  __syncthreads();
  {   // exclusive prefix of the word popcounts: thread t owns the words [t * per, (t + 1) * per)
    const uint32_t per = (nw + PJ_NT - 1) / PJ_NT, w0 = threadIdx.x * per; uint32_t sum = 0;
    for (uint32_t j = 0; j < per; j++) if (w0 + j < nw) sum += (uint32_t)__popcll(bits[w0 + j]);
    uint32_t inc = wave_inclusive_sum(sum);
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t run = inc - sum; for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) run += wsum[w];
    for (uint32_t j = 0; j < per; j++) if (w0 + j < nw) { pre[w0 + j] = run; run += (uint32_t)__popcll(bits[w0 + j]); }
  }
  __syncthreads();
  for (uint32_t i = m0 + threadIdx.x; i < m1; i += PJ_NT) {
    uint64_t rec = mb[i]; uint32_t r = (uint32_t)(rec & 0xFFFFFFFFull), d = r - r0;
    uint32_t o = m0 + pre[d >> 6] + (uint32_t)__popcll(bits[d >> 6] & ((1ull << (d & 63)) - 1ull));
    out_probe[o] = r; out_build[o] = rec >> 32;
  }
}

static void pj_set_lds_limit(const void* fn) { HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512)); }

static bool pj_key_type_ok(const dfgpu_array* a) {
  switch (a->type) {
    case DFGPU_INT8: case DFGPU_INT16: case DFGPU_INT32: case DFGPU_INT64: case DFGPU_DATE32:
    case DFGPU_UINT8: case DFGPU_UINT16: case DFGPU_UINT32: case DFGPU_UINT64: return true;
    default: return false;
  }
}

// partition the rows of an integer key column (selected and non-NULL ones) by rp_pid(mix64(widened key))
static RpResult pj_partition(dfgpu_ctx* ctx, const dfgpu_array* key, const uint64_t* mask, uint32_t P, const RpCols& cols, uint64_t* d_total, const char* th, const char* ts, const char* tw) {
  const uint64_t* valid = key->validity ? (const uint64_t*)key->validity->ptr : nullptr; const int64_t n = key->length;
#define PJ_PART(T) return rp_partition(ctx, RpHashInt<T>{ (const T*)key->values->ptr, valid, mask }, n, P, cols, false, d_total, th, ts, tw)
  switch (key->type) {
    case DFGPU_INT8: PJ_PART(int8_t); case DFGPU_INT16: PJ_PART(int16_t); case DFGPU_INT32: case DFGPU_DATE32: PJ_PART(int32_t);
    case DFGPU_UINT8: PJ_PART(uint8_t); case DFGPU_UINT16: PJ_PART(uint16_t); case DFGPU_UINT32: PJ_PART(uint32_t);
    default: PJ_PART(int64_t);          // INT64 / UINT64: the same 64-bit pattern
  }
#undef PJ_PART
}

bool pj_build(dfgpu_ctx* ctx, dfgpu_join_table* t) {
  const int64_t n = t->n_build;
  if (!ctx->join_partitioned || ctx->force_hash_collisions || t->nkeys != 1 || t->null_equals_null) return false;
  if (n < ctx->join_partitioned_min_build || n > 0xFFFFFFF0ll) return false;
  const dfgpu_array* key0 = t->keys[0];
  if (!pj_key_type_ok(key0)) return false;
  int64_t per = ctx->join_partition_rows; if (per < 16) per = 16; if (per > 14000) per = 14000;
  int64_t P64 = (n + per - 1) / per; if (P64 < 1) P64 = 1;
  if (P64 > ctx->num_cus && P64 <= RP_MAX_P) P64 = std::min<int64_t>(RP_MAX_P, (P64 + ctx->num_cus - 1) / ctx->num_cus * ctx->num_cus);      // whole rounds of one workgroup per CU
  if (P64 > RP_MAX_P) return false;                    // larger builds: two partition passes (not built yet) -> general path
  auto part = std::make_unique<PartitionedBuild>();
  part->P = (uint32_t)P64;
  BufferPtr keys = alloc_buffer(ctx, (size_t)n * 8), rows = alloc_buffer(ctx, (size_t)n * 4);
  RpCols cols{}; cols.n = 1; cols.rowid_dst = (uint32_t*)rows->ptr;
  cols.c[0] = RpCol{ key0->values->ptr, keys->ptr, 8, RP_HASHKEY, key0->type };
  zero_scratch(ctx);
  RpResult r = pj_partition(ctx, key0, t->build_mask ? (const uint64_t*)t->build_mask->ptr : nullptr, part->P, cols, ctx->d_scratch64 + 3, "pj_build_hist", "pj_build_scan", "pj_build_scatter");
  { KernelTimer kt_(ctx, "pj_build_check");
    static bool once = false; if (!once) { pj_set_lds_limit((const void*)k_pj_check); once = true; }
    hipLaunchKernelGGL(k_pj_check, dim3(part->P), dim3(PJ_NT), (size_t)(1u << PJ_MAX_SBITS) * 4, ctx->stream, (const uint64_t*)keys->ptr, (const uint32_t*)r.starts->ptr, (unsigned long long*)ctx->d_scratch64);
    KERNEL_CHECK(); }
  HIP_CHECK(hipMemcpyAsync(ctx->h_pinned, ctx->d_scratch64, 32, hipMemcpyDeviceToHost, ctx->stream));
  ctx->count_sync("sync:pj_build_check");
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  const uint64_t max_rows = ctx->h_pinned[0], dup = ctx->h_pinned[1], over = ctx->h_pinned[2], moved = ctx->h_pinned[3];
  if (dup || over) return false;
  int sbits = 6; while ((1ull << sbits) < 2 * max_rows && sbits < PJ_MAX_SBITS) sbits++;
  part->sbits = sbits; part->rows = (int64_t)moved; part->recs = keys; part->row_ids = rows; part->starts = r.starts;
  t->part = std::move(part);
  t->unique = true;
  t->mem += (int64_t)n * 12 + (int64_t)(P64 + 1) * 4;
  return true;
}

bool pj_probe_eligible(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* probe_key, int64_t n) {
  if (!t->part || !ctx->join_partitioned || n < ctx->join_partitioned_min_probe || n > (int64_t)RP_MAX_P << 19) return false;    // row-range buckets of <= 2^19 rows (k_pj_order's LDS bitmap)
  return probe_key->type == t->keys[0]->type;          // same physical integer type, no dictionary
}

void pj_probe(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* pk, const uint64_t* mask, dfgpu_array** out_build, dfgpu_array** out_probe) {
  const PartitionedBuild& part = *t->part;
  const int64_t n = pk->length;
  BufferPtr keys = alloc_buffer(ctx, (size_t)n * 8), rows = alloc_buffer(ctx, (size_t)n * 4);
  RpCols cols{}; cols.n = 1; cols.rowid_dst = (uint32_t*)rows->ptr;
  cols.c[0] = RpCol{ pk->values->ptr, keys->ptr, 8, RP_HASHKEY, pk->type };
  RpResult r = pj_partition(ctx, pk, mask, part.P, cols, ctx->d_scratch64 + 9, "pj_probe_hist", "pj_probe_scan", "pj_probe_scatter");
  BufferPtr mraw = alloc_buffer(ctx, (size_t)n * 8), mcount = alloc_buffer(ctx, (size_t)part.P * 4), moff = alloc_buffer(ctx, (size_t)(part.P + 1) * 8);
  { KernelTimer kt_(ctx, "pj_join");
    static bool once = false; if (!once) { pj_set_lds_limit((const void*)k_pj_join); once = true; }
    hipLaunchKernelGGL(k_pj_join, dim3(part.P), dim3(PJ_NT), (size_t)(1u << part.sbits) * 4 + 16, ctx->stream, (const uint64_t*)part.recs->ptr, (const uint32_t*)part.row_ids->ptr, (const uint32_t*)part.starts->ptr,
                       (const uint64_t*)keys->ptr, (const uint32_t*)rows->ptr, (const uint32_t*)r.starts->ptr, part.sbits, (uint64_t*)mraw->ptr, (uint32_t*)mcount->ptr);
    KERNEL_CHECK(); }
  exclusive_scan_u32(ctx, (const uint32_t*)mcount->ptr, (uint64_t*)moff->ptr, part.P, ctx->d_scratch64 + 10);
  const int64_t total = (int64_t)read_scratch(ctx, 10);
  ArrayHolder ob(new_fixed(ctx, DFGPU_UINT64, total)), op(new_fixed(ctx, DFGPU_UINT32, total));
  if (total) {
    // the hits of all partitions back to back (reusing the partitioned key buffer), then bucketed by probe-row range
    uint64_t* mc = (uint64_t*)keys->ptr;
    { KernelTimer kt_(ctx, "pj_mcopy");
      hipLaunchKernelGGL(k_pj_mcopy, dim3(part.P), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)mraw->ptr, (const uint32_t*)r.starts->ptr, (const uint32_t*)mcount->ptr, (const uint64_t*)moff->ptr, mc);
      KERNEL_CHECK(); }
    int shift = 12; while ((((int64_t)n + (1ll << shift) - 1) >> shift) > (int64_t)RP_MAX_P) shift++;
    const uint32_t P2 = (uint32_t)((n + (1ll << shift) - 1) >> shift);
    RpCols mcols{}; mcols.n = 1; mcols.c[0] = RpCol{ mc, mraw->ptr, 8, RP_HASHKEY, 0 };
    RpResult rb = rp_partition(ctx, RpHashRowBucket{ mc, shift }, total, P2, mcols, false, ctx->d_scratch64 + 11, "pj_order_hist", "pj_order_scan", "pj_order_scatter");
    KernelTimer kt_(ctx, "pj_order");
    static bool once = false; if (!once) { pj_set_lds_limit((const void*)k_pj_order); once = true; }
    hipLaunchKernelGGL(k_pj_order, dim3(P2), dim3(PJ_NT), (size_t)(1u << (shift - 6)) * 12, ctx->stream, (const uint64_t*)mraw->ptr, (const uint32_t*)rb.starts->ptr, shift,
                       (uint32_t*)op.get()->values->ptr, (uint64_t*)ob.get()->values->ptr);
    KERNEL_CHECK();
  }
  op.get()->identity = total == n;
  *out_build = ob.release(); *out_probe = op.release();
}

}  // namespace dfgpu
