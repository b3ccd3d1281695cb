// pjoin.hip -- radix-partitioned hash join for unsorted / sparse integer keys: build + probe of every partition out of LDS.
//
// Reference semantics: JoinHashMap build + lookup_join_hashmap (physical-plan/src/joins/utils.rs:121-229, hash_join.rs:1024-1118);
// the (build, probe) pairs come back ordered by probe row exactly as the reference emits them.
//
// Why: the global open-addressing table of join.hip costs every probe row a random 64-B sector out of a table far larger than L2
// (150 M probes of a 15 M-row build: ~53 G sectors/s on MI355X whatever the table's size beyond L2, i.e. >= 2.8 ms, measured 4.5 ms with
// the key verification; profiles/experiments/random_gather_microbench.hip).  Here both sides are split by hash bits (radix_partition.h:
// one pass, LDS-staged write combining) into P partitions of <= 16 K build rows, so that one partition's table -- 32 K four-byte slots
// (tag : 17 | partition-local build row : 14) -- sits in the 160 KB LDS of a CU.  One workgroup per partition builds the table from the
// partition's (key, row) records and streams the partition's probe rows against it; a tag match is verified against the build record
// (exact: full 64-bit key compare), which for a hit also yields the original build row.
//
// Order: partitioning destroys probe order, so a hit stores found[probe row] = build row (4-byte scatter into an array pre-set to NONE) and an
// order-preserving compaction of found[] emits the pairs: output order is a function of the input only.  (Tried instead: hits written out
// per partition, re-partitioned by probe-row range and ranked inside LDS per range -- 1.29 ms against 0.95 ms for 30 M hits out of 150 M rows.)
// Build keys must be unique (every PK-FK join); a repeated build key, a partition beyond the table's capacity (adversarial hash skew) or an
// unsupported key type leave the table to join.hip's general path.
#include "join_table.h"
#include "radix_partition.h"

namespace dfgpu {

constexpr uint32_t PJ_EMPTY = 0xFFFFFFFFu;
constexpr int PJ_IDX_BITS = 14;                       // partition-local build row
constexpr uint32_t PJ_IDX_MASK = (1u << PJ_IDX_BITS) - 1u;
constexpr uint32_t PJ_TAG_MASK = (1u << (31 - PJ_IDX_BITS)) - 1u;       // 17 bits: an entry never has its top bit set, PJ_EMPTY always has
constexpr uint32_t PJ_MAX_PART_ROWS = PJ_IDX_MASK - 1;      // keeps (tag, idx) != PJ_EMPTY
constexpr int PJ_MAX_SBITS = 15;                      // 32 K slots x 4 B = 128 KB
constexpr int PJ_NT = 1024;                           // one workgroup per CU while a 128 KB table is resident
constexpr int PJ_U = 4;                               // probe rows per lane in flight

// ---- build side check: one workgroup per partition inserts the partition's keys into the LDS table exactly as the probe kernel will;
// reports a repeated key ([1]) or an over-full partition ([2]) and the largest partition ([0])
// Slots are read four at a time (one ds_read_b128 per step of a walk): a key's home is the first slot of the 4-slot group its hash selects and
// it sits in the first free slot from there on, so a lookup that walks group by group may stop at the first EMPTY it sees.
__device__ inline uint64_t pj_key(const RpRec12& r) { return (uint64_t)r.lo | ((uint64_t)r.hi << 32); }
// slot group and tag of a key inside its partition: three 32-bit multiplies (the partition was chosen by mix64's top bits; the walk is
// issue bound, 64-bit multiplies per probe row cost as much as the walk itself).  Collisions only cost a verification.
__device__ inline void pj_hash(uint32_t lo, uint32_t hi, uint32_t M, int sbits, uint32_t* group, uint32_t* tagsh) {
  uint32_t a = lo ^ (hi * 0x9E3779B1u), x = a * 0x85EBCA6Bu; x ^= x >> 13;
  uint32_t y = x * 0xC2B2AE35u;
  *group = (y >> (32 - sbits)) & M & ~3u;
  *tagsh = ((y ^ (y >> 16) ^ a) & PJ_TAG_MASK) << PJ_IDX_BITS;
}
// Slots are read four at a time (one ds_read_b128 per step of a walk): a key's home is the first slot of the 4-slot group its hash selects and
// it sits in the first free slot from there on, so a lookup that walks group by group may stop at the first EMPTY it sees.
__device__ inline void pj_build_table(uint32_t* tab, uint32_t M, int sbits, const RpRec12* brec, uint32_t nb, unsigned long long* flags) {
  for (uint32_t j0 = threadIdx.x; j0 < nb; j0 += PJ_NT * 4) {
    RpRec12 k[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { uint32_t j = j0 + u * PJ_NT; k[u] = brec[j < nb ? j : nb - 1]; }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      uint32_t j = j0 + u * PJ_NT; if (j >= nb) break;
      uint32_t s, tagsh; pj_hash(k[u].lo, k[u].hi, M, sbits, &s, &tagsh); const uint32_t ent = tagsh | j;
      for (;;) {
        uint32_t old = atomicCAS(&tab[s], PJ_EMPTY, ent);
        if (old == PJ_EMPTY) break;
        if ((old & ~PJ_IDX_MASK) == tagsh) { RpRec12 o = brec[old & PJ_IDX_MASK]; if (o.lo == k[u].lo && o.hi == k[u].hi) { if (flags) flags[1] = 1ull; break; } }     // the key is in the table already
        s = (s + 1) & M;
      }
    }
  }
}
// one 4-slot group of a walk: true = the walk ends here, with *cand = the first slot whose tag matches (PJ_EMPTY if an EMPTY came first)
__device__ inline bool pj_group(const uint4 v, uint32_t tagsh, uint32_t s, uint32_t* cand, uint32_t* pos) {
  // branch-free: per slot 2 = tag match, 1 = EMPTY (top bit); the first slot that is either decides
  const uint32_t c0 = ((v.x & ~PJ_IDX_MASK) == tagsh ? 2u : 0u) | (v.x >> 31), c1 = ((v.y & ~PJ_IDX_MASK) == tagsh ? 2u : 0u) | (v.y >> 31);
  const uint32_t c2 = ((v.z & ~PJ_IDX_MASK) == tagsh ? 2u : 0u) | (v.z >> 31), c3 = ((v.w & ~PJ_IDX_MASK) == tagsh ? 2u : 0u) | (v.w >> 31);
  uint32_t code = c3, val = v.w, j = 3;
  code = c2 ? c2 : code; val = c2 ? v.z : val; j = c2 ? 2u : j;
  code = c1 ? c1 : code; val = c1 ? v.y : val; j = c1 ? 1u : j;
  code = c0 ? c0 : code; val = c0 ? v.x : val; j = c0 ? 0u : j;
  *cand = (code & 2u) ? val : PJ_EMPTY; *pos = s + j;
  return code != 0;
}
__global__ void __launch_bounds__(PJ_NT) k_pj_check(const RpRec12* brec, const uint32_t* bstart, unsigned long long* flags) {
  extern __shared__ uint32_t pj_tab[];
  const uint32_t b0 = bstart[blockIdx.x], nb = bstart[blockIdx.x + 1] - b0;
  if (threadIdx.x == 0) atomicMax(&flags[0], (unsigned long long)nb);
  if (nb > PJ_MAX_PART_ROWS) { if (threadIdx.x == 0) flags[2] = 1ull; return; }
  int sbits = 6; while ((1u << sbits) < 2 * nb && sbits < PJ_MAX_SBITS) sbits++;
  if ((1u << sbits) < nb + nb / 8 + 4) { if (threadIdx.x == 0) flags[2] = 1ull; return; }
  const uint32_t S = 1u << sbits;
  for (uint32_t s = threadIdx.x; s < S; s += PJ_NT) pj_tab[s] = PJ_EMPTY;
  __syncthreads();
  pj_build_table(pj_tab, S - 1, sbits, brec + b0, nb, flags);
}

// ---- probe: one workgroup per partition
__global__ void __launch_bounds__(PJ_NT) k_pj_join(const RpRec12* brec, const uint32_t* bstart, const RpRec12* prec, const uint32_t* pstart, int sbits, uint32_t* found) {
  extern __shared__ uint4 tab4[];                      // 16-byte aligned: one ds_read_b128 per group
  uint32_t* const pj_tab = (uint32_t*)tab4;
  const uint32_t S = 1u << sbits, M = S - 1;
  const int p = blockIdx.x;
  const uint32_t q0 = pstart[p], q1 = pstart[p + 1];
  if (q0 == q1) return;
  for (uint32_t s = threadIdx.x; s < S; s += PJ_NT) pj_tab[s] = PJ_EMPTY;
  __syncthreads();
  const uint32_t b0 = bstart[p], nb = bstart[p + 1] - b0;
  const RpRec12* br = brec + b0;
  pj_build_table(pj_tab, M, sbits, br, nb, nullptr);
  __syncthreads();
  // PJ_U rows per lane and step.  Vector-memory loads return in issue order, so the records are fetched TWO steps ahead and a step
  // issues its loads as [verification of this step's hits (L2), records of step + 2 (HBM)]: the verification never waits behind an HBM
  // fetch of the same step, and a step's records have had a whole step to arrive.
  RpRec12 rc[PJ_U], rn[PJ_U], rnn[PJ_U];
  uint32_t i0 = q0 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < PJ_U; u++) { uint32_t i = i0 + u * PJ_NT; rn[u] = prec[i < q1 ? i : q1 - 1]; }
#pragma unroll
  for (int u = 0; u < PJ_U; u++) { uint32_t i = i0 + (PJ_U + u) * PJ_NT; rnn[u] = prec[i < q1 ? i : q1 - 1]; }
  for (; i0 < q1; i0 += PJ_NT * PJ_U) {
    uint32_t cand[PJ_U], pos[PJ_U]; bool on[PJ_U];
#pragma unroll
    for (int u = 0; u < PJ_U; u++) { rc[u] = rn[u]; rn[u] = rnn[u]; on[u] = i0 + u * PJ_NT < q1; }
    // first group of every row: PJ_U independent 16-byte LDS reads, straight-line; the few rows whose first group is full of other
    // keys (no EMPTY, no tag match) go on in the loop below
    uint32_t s[PJ_U], tagsh[PJ_U]; uint4 v[PJ_U]; bool more = false;
#pragma unroll
    for (int u = 0; u < PJ_U; u++) { pj_hash(rc[u].lo, rc[u].hi, M, sbits, &s[u], &tagsh[u]); v[u] = tab4[s[u] >> 2]; }
    bool walking[PJ_U];
#pragma unroll
    for (int u = 0; u < PJ_U; u++) { const bool done = pj_group(v[u], tagsh[u], s[u], &cand[u], &pos[u]); walking[u] = on[u] & !done; cand[u] = on[u] ? cand[u] : PJ_EMPTY; more |= walking[u]; }
    while (__ballot(more)) {
      more = false;
#pragma unroll
      for (int u = 0; u < PJ_U; u++) if (walking[u]) {
        s[u] = (s[u] + 4) & M;
        walking[u] = !pj_group(tab4[s[u] >> 2], tagsh[u], s[u], &cand[u], &pos[u]); more |= walking[u];
      }
    }
    RpRec12 vb[PJ_U];
#pragma unroll
    for (int u = 0; u < PJ_U; u++) { vb[u] = RpRec12{0, 0, 0}; if (cand[u] != PJ_EMPTY) vb[u] = br[cand[u] & PJ_IDX_MASK]; }
    __builtin_amdgcn_sched_barrier(0);                 // keep the verification loads ahead of the record fetch below
    const uint32_t i2 = i0 + 2 * PJ_NT * PJ_U;
#pragma unroll
    for (int u = 0; u < PJ_U; u++) { uint32_t i = i2 + u * PJ_NT; rnn[u] = prec[i < q1 ? i : q1 - 1]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < PJ_U; u++) {
      if (cand[u] == PJ_EMPTY) continue;
      if (vb[u].lo == rc[u].lo && vb[u].hi == rc[u].hi) { found[rc[u].row] = vb[u].row; continue; }
      // a different key with the same 17-bit tag (2^-17 per occupied slot passed): keep walking, verifying every tag match
      uint32_t s1 = (pos[u] + 1) & M, c = pj_tab[s1];
      while (c != PJ_EMPTY) {
        if ((c & ~PJ_IDX_MASK) == tagsh[u]) { RpRec12 w = br[c & PJ_IDX_MASK]; if (w.lo == rc[u].lo && w.hi == rc[u].hi) { found[rc[u].row] = w.row; break; } }
        s1 = (s1 + 1) & M; c = pj_tab[s1];
      }
    }
  }
}

// ---- order-preserving compaction of found[]: 4096 rows per workgroup, 16 consecutive rows per lane; the hits of a workgroup are staged in
// LDS in row order and leave with consecutive lanes writing consecutive output slots
__global__ void __launch_bounds__(BLOCK) k_pj_found_count(const uint32_t* found, int64_t n, uint32_t* counts) {
  int64_t base = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * 16; uint32_t c = 0;
  if (base + 16 <= n) { const uint4* p = (const uint4*)(found + base);
#pragma unroll
    for (int q = 0; q < 4; q++) { uint4 v = p[q]; c += (v.x != PJ_EMPTY) + (v.y != PJ_EMPTY) + (v.z != PJ_EMPTY) + (v.w != PJ_EMPTY); } }
  else for (int64_t i = base; i < n; i++) c += found[i] != PJ_EMPTY;
  __shared__ uint32_t lds[BLOCK / WAVE];
  uint32_t tot; (void)block_exclusive_sum<uint32_t>(c, lds, &tot);
  if (threadIdx.x == 0) counts[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(BLOCK) k_pj_found_write(const uint32_t* found, int64_t n, const uint32_t* offs, uint32_t* out_probe, uint64_t* out_build) {
  __shared__ uint32_t sp[4096], sb[4096]; __shared__ uint32_t lds[BLOCK / WAVE];
  int64_t base = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * 16; uint32_t v[16]; uint32_t c = 0;
  if (base + 16 <= n) { const uint4* p = (const uint4*)(found + base);
#pragma unroll
    for (int q = 0; q < 4; q++) { uint4 x = p[q]; v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w; } }
  else {
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = base + q < n ? found[base + q] : PJ_EMPTY; }
#pragma unroll
  for (int q = 0; q < 16; q++) c += v[q] != PJ_EMPTY;
  uint32_t tot; uint32_t ex = block_exclusive_sum<uint32_t>(c, lds, &tot);
#pragma unroll
  for (int q = 0; q < 16; q++) if (v[q] != PJ_EMPTY) { sp[ex] = (uint32_t)(base + q); sb[ex] = v[q]; ex++; }
  __syncthreads();
  const uint32_t o = offs[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < tot; i += BLOCK) { out_probe[o + i] = sp[i]; out_build[o + i] = sb[i]; }
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
  if (P64 > ctx->num_cus && P64 <= 2048) P64 = std::min<int64_t>(2048, (P64 + ctx->num_cus - 1) / ctx->num_cus * ctx->num_cus);      // whole rounds of one workgroup per CU
  if (P64 > 2048) return false;                        // larger builds: finer partitions cost more than the general path saves
  auto part = std::make_unique<PartitionedBuild>();
  part->P = (uint32_t)P64;
  BufferPtr recs = alloc_buffer(ctx, (size_t)n * 12);
  RpCols cols{}; cols.n = 1; cols.pack12_dst = (RpRec12*)recs->ptr;
  cols.c[0] = RpCol{ key0->values->ptr, nullptr, 8, RP_HASHKEY, key0->type };
  zero_scratch(ctx);
  RpResult r = pj_partition(ctx, key0, t->build_mask ? (const uint64_t*)t->build_mask->ptr : nullptr, part->P, cols, ctx->d_scratch64 + 3, "pj_build_hist", "pj_build_scan", "pj_build_scatter");
  { KernelTimer kt_(ctx, "pj_build_check");
    static bool once = false; if (!once) { pj_set_lds_limit((const void*)k_pj_check); once = true; }
    hipLaunchKernelGGL(k_pj_check, dim3(part->P), dim3(PJ_NT), (size_t)(1u << PJ_MAX_SBITS) * 4, ctx->stream, (const RpRec12*)recs->ptr, (const uint32_t*)r.starts->ptr, (unsigned long long*)ctx->d_scratch64);
    KERNEL_CHECK(); }
  HIP_CHECK(hipMemcpyAsync(ctx->h_pinned, ctx->d_scratch64, 32, hipMemcpyDeviceToHost, ctx->stream));
  ctx->count_sync("sync:pj_build_check");
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  const uint64_t max_rows = ctx->h_pinned[0], dup = ctx->h_pinned[1], over = ctx->h_pinned[2], moved = ctx->h_pinned[3];
  if (dup || over) return false;
  int sbits = 6; while ((1ull << sbits) < 2 * max_rows && sbits < PJ_MAX_SBITS) sbits++;
  part->sbits = sbits; part->rows = (int64_t)moved; part->recs = recs; part->starts = r.starts;
  t->part = std::move(part);
  t->unique = true;
  t->mem += (int64_t)n * 12 + (int64_t)(P64 + 1) * 4;
  return true;
}

bool pj_probe_eligible(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* probe_key, int64_t n) {
  if (!t->part || !ctx->join_partitioned || n < ctx->join_partitioned_min_probe || n > 0xFFFF0000ll) return false;      // 32-bit slot arithmetic with two steps of look-ahead
  return probe_key->type == t->keys[0]->type;          // same physical integer type, no dictionary
}

void pj_probe(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* pk, const uint64_t* mask, dfgpu_array** out_build, dfgpu_array** out_probe) {
  const PartitionedBuild& part = *t->part;
  const int64_t n = pk->length;
  BufferPtr recs = alloc_buffer(ctx, (size_t)n * 12), found = alloc_buffer(ctx, (size_t)n * 4);
  RpCols cols{}; cols.n = 1; cols.pack12_dst = (RpRec12*)recs->ptr;
  cols.c[0] = RpCol{ pk->values->ptr, nullptr, 8, RP_HASHKEY, pk->type };
  RpResult r = pj_partition(ctx, pk, mask, part.P, cols, ctx->d_scratch64 + 9, "pj_probe_hist", "pj_probe_scan", "pj_probe_scatter");
  { KernelTimer kt_(ctx, "pj_join");
    HIP_CHECK(hipMemsetAsync(found->ptr, 0xFF, (size_t)n * 4, ctx->stream));
    static bool once = false; if (!once) { pj_set_lds_limit((const void*)k_pj_join); once = true; }
    hipLaunchKernelGGL(k_pj_join, dim3(part.P), dim3(PJ_NT), (size_t)(1u << part.sbits) * 4, ctx->stream, (const RpRec12*)part.recs->ptr, (const uint32_t*)part.starts->ptr,
                       (const RpRec12*)recs->ptr, (const uint32_t*)r.starts->ptr, part.sbits, (uint32_t*)found->ptr);
    KERNEL_CHECK(); }
  recs.reset();
  KernelTimer kt_(ctx, "pj_compact");
  const int64_t nb = (n + 4095) / 4096;
  BufferPtr counts = alloc_buffer(ctx, (size_t)nb * 4);
  hipLaunchKernelGGL(k_pj_found_count, dim3((unsigned)nb), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)found->ptr, n, (uint32_t*)counts->ptr);
  exclusive_scan_u32_inplace32(ctx, (uint32_t*)counts->ptr, nb, ctx->d_scratch64 + 10);
  KERNEL_CHECK();
  const int64_t total = (int64_t)read_scratch(ctx, 10);
  ArrayHolder ob(new_fixed(ctx, DFGPU_UINT64, total)), op(new_fixed(ctx, DFGPU_UINT32, total));
  if (total) hipLaunchKernelGGL(k_pj_found_write, dim3((unsigned)nb), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)found->ptr, n, (const uint32_t*)counts->ptr, (uint32_t*)op.get()->values->ptr, (uint64_t*)ob.get()->values->ptr);
  KERNEL_CHECK();
  op.get()->identity = total == n;
  *out_build = ob.release(); *out_probe = op.release();
}

}  // namespace dfgpu
