// pjoin.hip -- radix-partitioned hash join for unsorted integer keys: build + probe of every partition out of LDS.
//
// Reference semantics: JoinHashMap build + lookup_join_hashmap (physical-plan/src/joins/utils.rs:121-229, hash_join.rs:1024-1118);
// the (build, probe) pairs come back ordered by probe row, then build row, exactly as the reference emits them.
//
// Why: the global open-addressing table of join.hip costs every probe row a random 64-B sector out of a table far larger than L2
// (150 M probes of a 15 M-row build: ~53 G sectors/s on MI355X whatever the table's size beyond L2, i.e. >= 2.8 ms).  Here both sides are
// split by hash bits into P <= 2048 partitions so that one partition's table -- 16 K four-byte slots (tag : 17 | partition-local index : 14)
// -- sits in 64 KB of LDS, two workgroups per CU.  Builds beyond 2048 x 14 000 rows take up to 4096 partitions (round 4: the offset loops of the scatter, the restore and
// the partition starts instantiated with four partitions per thread; 40 M x 150 M rows: 4.7 ms against 8.1 ms through the global table); beyond 4096 x 14 000 they decline.
//
// Round 3 pipeline (profiles/experiments/pjoin2_microbench.hip holds the measurements that chose it; 15 M x 150 M rows, 20 % match):
//   partition   k_pj_hist     per chunk of 16 tiles (8192 rows each): LDS histogram in u16 counters with the next tile's keys in flight; writes
//                             pre[tile][p] = rows of partition p in the chunk's tiles in front of this tile (TILE-major: a tile reads its 2048
//                             offsets as 8 KB of coalesced loads -- the partition-major matrix cost 2048 separate 64-B lines per tile) and
//                             tot[chunk][p]; no device-wide scan of the 37 M-cell matrix, only the 2 M chunk totals are prefixed (0.49 -> 0.30 ms)
//               k_pj_scatter  counting sort of the tile in LDS, staged in two rounds of half a tile (64 KB of LDS: two workgroups per CU, one
//                             loading while the other writes), 12-byte records (key, row) written as runs (1.21 -> 0.82 ms)
//   join        k_pj_join     one workgroup per partition; every wave owns a contiguous range of probe-row chunks = a contiguous slice of the
//                             partition's records (tiles lie in row order inside a partition) and emits its hits (probe row, build ref) in
//                             record order behind the slice's first record: no barrier in the probe loop, no found[probe row] scatter (that
//                             random 4-byte store cost 0.85 ms of sector read-modify-write per 30 M hits); hstart[p][c] = first hit of chunk c
//   restore     k_pj_restore  one workgroup per chunk of 2^14 probe rows gathers the chunk's hits from the 2048 partition lists (runs of
//                             consecutive hits), ranks them by probe row through a bitmap of the chunk's rows (at most one hit per probe row:
//                             repeated build keys travel as ONE group reference) and writes the pairs in probe order through an LDS window
// Repeated build keys (foreign-key builds): k_pj_groups numbers the distinct keys of every partition and lays the build rows out as a CSR
// (rows of a key ascending); the LDS tables then hold one entry per distinct key, a hit carries the group, and k_pj_expand emits the group's
// rows after the order has been restored -- pairs and their order identical to the general path.
#include "join_table.h"
#include "radix_partition.h"

namespace dfgpu {

constexpr uint32_t PJ_EMPTY = 0xFFFFFFFFu;
constexpr int PJ_IDX_BITS = 14;                       // partition-local record
constexpr uint32_t PJ_IDX_MASK = (1u << PJ_IDX_BITS) - 1u;
constexpr uint32_t PJ_TAG_MASK = (1u << (31 - PJ_IDX_BITS)) - 1u;       // 17 bits: an entry never has its top bit set, PJ_EMPTY always has
constexpr uint32_t PJ_MAX_PART_ROWS = PJ_IDX_MASK - 1;      // keeps (tag, idx) != PJ_EMPTY
constexpr int PJ_MAX_SBITS = 15;                      // 32 K slots x 4 B = 128 KB (one workgroup per CU); 14 = two per CU
constexpr int PJ_NT = 1024, PJ_R = 8, PJ_TILE = PJ_NT * PJ_R, PJ_G = 16, PJ_NW = PJ_NT / WAVE;
constexpr uint32_t PJ_MAX_P = 2048;                   // two partitions per thread in the offset loops
constexpr uint32_t PJ_MAX_P_BIG = 4096;               // builds beyond 2048 x 14 000 rows: the same kernels with four partitions per thread (template MP); 57 M rows at 14 000 per partition
constexpr int PJ_CHS = 14;                            // restore chunk = 2^14 probe rows = two tiles
static_assert(PJ_TILE == 1 << 13 && PJ_CHS >= 13, "k_pj_join turns a chunk into its first tile by a shift");
constexpr int PJ_U = 4;                               // probe records per lane in flight
constexpr uint32_t PJ_DUP_MAX_ROWS = 8192;            // repeated keys: a partition's records, slot counters and CSR cursors share 96 KB of LDS
constexpr uint32_t PJ_MAX_GROUP = 256;                // rows of one key a single thread puts in order

struct PjOffsets { BufferPtr pre, cpre, pstart; int64_t ntiles = 0, nchunks = 0; };      // pre [ntiles][P], cpre [nchunks][P], pstart [P + 1]

// ---------------------------------------------------------------------------------------------------------------- partition
template <typename H>
__global__ void __launch_bounds__(PJ_NT) k_pj_hist(H hs, int64_t n, uint32_t P, int64_t ntiles, uint32_t* pre /*[ntiles][P]*/, uint32_t* tot /*[nchunks][P]*/) {
  extern __shared__ uint32_t pj_lds[];        // [P][G] u16, two counters per word
  const int64_t t0 = (int64_t)blockIdx.x * PJ_G;
  for (int x = threadIdx.x; x < (int)P * PJ_G / 2; x += PJ_NT) pj_lds[x] = 0;
  // counter index of every row of a tile (0xFFFFFFFF: not selected); the next tile's keys are loaded and hashed before this tile's LDS atomics issue
  auto tile_counters = [&](int g, uint32_t* c) {
    const int64_t base = (t0 + g) * (int64_t)PJ_TILE;
#pragma unroll
    for (int q = 0; q < PJ_R; q++) { const int64_t i = base + (int64_t)q * PJ_NT + threadIdx.x; uint32_t pid = 0; uint64_t hk; c[q] = (i < n && hs(i, P, &pid, &hk)) ? pid * PJ_G + g : 0xFFFFFFFFu; }
  };
  uint32_t cn[PJ_R];
  tile_counters(0, cn);
  __syncthreads();
  for (int g = 0; g < PJ_G; g++) {
    if ((t0 + g) * (int64_t)PJ_TILE >= n) break;
    uint32_t c[PJ_R];
#pragma unroll
    for (int q = 0; q < PJ_R; q++) c[q] = cn[q];
    if (g + 1 < PJ_G) tile_counters(g + 1, cn);
#pragma unroll
    for (int q = 0; q < PJ_R; q++) if (c[q] != 0xFFFFFFFFu) atomicAdd(&pj_lds[c[q] >> 1], 1u << ((c[q] & 1) * 16));      // <= 8192 per counter: no carry
  }
  __syncthreads();
  for (int p = threadIdx.x; p < (int)P; p += PJ_NT) {
    const uint4* s = (const uint4*)(pj_lds + p * (PJ_G / 2)); const uint4 a = s[0], b = s[1];
    const uint32_t w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w }; uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (t0 + 2 * j < ntiles) pre[(t0 + 2 * j) * (int64_t)P + p] = sum;
      sum += w[j] & 0xFFFFu;
      if (t0 + 2 * j + 1 < ntiles) pre[(t0 + 2 * j + 1) * (int64_t)P + p] = sum;
      sum += w[j] >> 16;
    }
    tot[(int64_t)blockIdx.x * P + p] = sum;
  }
}
// within-partition exclusive prefix of the chunk totals, in place (chunk-major storage), and the partition totals: 64 partitions per workgroup, one wave per range of chunks
__global__ void __launch_bounds__(PJ_NT) k_pj_chunk_prefix(uint32_t* tot /*[nchunks][P]*/, int64_t nchunks, uint32_t P, uint32_t* ptot /*[P]*/) {
  __shared__ uint32_t part[PJ_NW][WAVE];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63; const uint32_t p = blockIdx.x * 64 + lane;
  const int64_t per = (nchunks + PJ_NW - 1) / PJ_NW, c0 = wave * per, c1 = c0 + per < nchunks ? c0 + per : nchunks;
  uint32_t s = 0;
  if (p < P) for (int64_t c = c0; c < c1; c++) s += tot[c * P + p];
  part[wave][lane] = s;
  __syncthreads();
  uint32_t run = 0; for (int w = 0; w < wave; w++) run += part[w][lane];
  if (wave == PJ_NW - 1 && p < P) ptot[p] = run + s;
  if (p < P) for (int64_t c = c0; c < c1; c++) { const uint32_t v = tot[c * P + p]; tot[c * P + p] = run; run += v; }
}
// pstart[p] = exclusive scan of ptot (P <= 2048, one workgroup), pstart[P] = rows moved (also -> *d_total)
template <int MP>
__global__ void __launch_bounds__(PJ_NT) k_pj_pstart(const uint32_t* ptot, uint32_t P, uint32_t* pstart, uint64_t* d_total) {
  __shared__ uint32_t wsum[PJ_NW];
  constexpr int PER = MP / PJ_NT; uint32_t v[PER]; uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) { const uint32_t p = threadIdx.x * PER + j; v[j] = p < P ? ptot[p] : 0; s += v[j]; }
  const uint32_t inc = wave_inclusive_sum(s);
  if (lane_id() == 63) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint32_t run = inc - s, tot = 0; for (int w = 0; w < PJ_NW; w++) { if (w < (int)(threadIdx.x >> 6)) run += wsum[w]; tot += wsum[w]; }
#pragma unroll
  for (int j = 0; j < PER; j++) { const uint32_t p = threadIdx.x * PER + j; if (p < P) pstart[p] = run; run += v[j]; }
  if (threadIdx.x == 0) { pstart[P] = tot; if (d_total) *d_total = tot; }
}
// LDS-staged scatter of (key, row) records; staging in two rounds of half the sorted tile: 8 P + 6 TILE bytes of LDS, two workgroups per CU.
// blockIdx -> tile: XCD x (= blockIdx & 7 under round-robin placement; speed only) works on a contiguous range of tiles, so the tiles in flight on one
// XCD are neighbours and their runs of a partition complete each other's cache lines inside that XCD's L2.
template <typename H, int MP = (int)PJ_MAX_P>
__global__ void __launch_bounds__(PJ_NT, 8) k_pj_scatter(H hs, int64_t n, uint32_t P, int64_t ntiles, const uint32_t* pre, const uint32_t* cpre, const uint32_t* pstart, RpRec12* out) {
  extern __shared__ uint32_t pj_lds[];
  constexpr int PIECE = PJ_TILE / 2;
  uint32_t* cnt = pj_lds; uint32_t* delta = pj_lds + P; uint16_t* spid = (uint16_t*)(pj_lds + 2 * P); uint16_t* slidx = spid + PIECE;
  uint64_t* skey = (uint64_t*)(((uintptr_t)(slidx + PIECE) + 7) & ~(uintptr_t)7);
  __shared__ uint32_t wsum[PJ_NW]; __shared__ uint32_t moved_sh;
  constexpr int PER = MP / PJ_NT;
  const int64_t per = (ntiles + 7) / 8, t = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (t >= ntiles || (int64_t)(blockIdx.x >> 3) >= per) return;
  const int64_t base = t * (int64_t)PJ_TILE, chunk = t / PJ_G;
  uint64_t k[PJ_R]; uint32_t pid[PJ_R], rk[PJ_R]; bool on[PJ_R]; uint32_t gc[PER];
#pragma unroll
  for (int q = 0; q < PJ_R; q++) { const int64_t i = base + (int64_t)q * PJ_NT + threadIdx.x; pid[q] = 0; k[q] = 0; on[q] = i < n && hs(i, P, &pid[q], &k[q]); }
#pragma unroll
  for (int j = 0; j < PER; j++) { const int p = threadIdx.x * PER + j; gc[j] = p < (int)P ? pre[t * (int64_t)P + p] + cpre[chunk * (int64_t)P + p] + pstart[p] : 0; }
  for (int p = threadIdx.x; p < (int)P; p += PJ_NT) cnt[p] = 0;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PJ_R; q++) rk[q] = on[q] ? atomicAdd(&cnt[pid[q]], 1u) : 0;
  __syncthreads();
  {
    uint32_t loc[PER]; uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) { const int p = threadIdx.x * PER + j; loc[j] = p < (int)P ? cnt[p] : 0; s += loc[j]; }
    const uint32_t inc = wave_inclusive_sum(s);
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t run = inc - s; for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += wsum[w];
#pragma unroll
    for (int j = 0; j < PER; j++) { const int p = threadIdx.x * PER + j; if (p < (int)P) { cnt[p] = run; delta[p] = gc[j] - run; run += loc[j]; } }      // mod 2^32: slot = delta + sorted position
    if (threadIdx.x == PJ_NT - 1) moved_sh = run;
  }
  __syncthreads();
  const uint32_t moved = moved_sh;
  uint32_t spos[PJ_R];
#pragma unroll
  for (int q = 0; q < PJ_R; q++) spos[q] = on[q] ? cnt[pid[q]] + rk[q] : 0xFFFFFFFFu;
#pragma unroll 1
  for (int h = 0; h < 2; h++) {
    const uint32_t lo = (uint32_t)h * PIECE;
    if (lo >= moved) break;
    if (h) __syncthreads();
#pragma unroll
    for (int q = 0; q < PJ_R; q++) { const uint32_t s = spos[q] - lo; if (s < (uint32_t)PIECE) { spid[s] = (uint16_t)pid[q]; slidx[s] = (uint16_t)(q * PJ_NT + threadIdx.x); skey[s] = k[q]; } }
    __syncthreads();
    const uint32_t m = moved - lo < (uint32_t)PIECE ? moved - lo : (uint32_t)PIECE;
    for (uint32_t i = threadIdx.x; i < m; i += PJ_NT) {
      const uint32_t pos = delta[spid[i]] + lo + i; const uint64_t v = skey[i];
      out[pos] = RpRec12{ (uint32_t)v, (uint32_t)(v >> 32), (uint32_t)(base + slidx[i]) };
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- LDS tables
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
// Inserts record j of `rec` (nb <= PJ_MAX_PART_ROWS); *found_slot (optional): the slot that holds the key afterwards, *inserted: this call put it there.
__device__ inline void pj_insert(uint32_t* tab, uint32_t M, int sbits, const RpRec12* rec, uint32_t j, const RpRec12 kj, uint32_t* found_slot, bool* inserted) {
  uint32_t s, tagsh; pj_hash(kj.lo, kj.hi, M, sbits, &s, &tagsh); const uint32_t ent = tagsh | j;
  for (;;) {
    const uint32_t old = atomicCAS(&tab[s], PJ_EMPTY, ent);
    if (old == PJ_EMPTY) { *found_slot = s; *inserted = true; return; }
    if ((old & ~PJ_IDX_MASK) == tagsh) { const RpRec12 o = rec[old & PJ_IDX_MASK]; if (o.lo == kj.lo && o.hi == kj.hi) { *found_slot = s; *inserted = false; return; } }     // the key is in the table already
    s = (s + 1) & M;
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

// ---- build side check: one workgroup per partition inserts the partition's keys into the LDS table exactly as the probe kernel will.
// flags: [0] largest partition, [1] a key repeats, [2] a partition is beyond the table's capacity, [4] most distinct keys in one partition; ndist[p] = distinct keys of partition p
// max_sbits: the table the launch has LDS for; a partition that needs a larger one raises flags[6] (the host launches again with 128 KB)
__global__ void __launch_bounds__(PJ_NT) k_pj_check(const RpRec12* brec, const uint32_t* bstart, int max_sbits, unsigned long long* flags, uint32_t* ndist) {
  extern __shared__ uint32_t pj_lds[];
  __shared__ uint32_t nd_sh;
  const uint32_t b0 = bstart[blockIdx.x], nb = bstart[blockIdx.x + 1] - b0;
  if (threadIdx.x == 0) { atomicMax(&flags[0], (unsigned long long)nb); nd_sh = 0; }
  if (nb > PJ_MAX_PART_ROWS) { if (threadIdx.x == 0) { flags[2] = 1ull; ndist[blockIdx.x] = nb; } return; }
  int sbits = 6; while ((1u << sbits) < 2 * nb && sbits < PJ_MAX_SBITS) sbits++;
  if ((1u << sbits) < nb + nb / 8 + 4) { if (threadIdx.x == 0) { flags[2] = 1ull; ndist[blockIdx.x] = nb; } return; }
  if (sbits > max_sbits) { if (threadIdx.x == 0) flags[6] = 1ull; return; }
  const uint32_t S = 1u << sbits;
  for (uint32_t s = threadIdx.x; s < S; s += PJ_NT) pj_lds[s] = PJ_EMPTY;
  __syncthreads();
  uint32_t mine = 0;
  for (uint32_t j = threadIdx.x; j < nb; j += PJ_NT) { uint32_t fs; bool ins; pj_insert(pj_lds, S - 1, sbits, brec + b0, j, brec[b0 + j], &fs, &ins); mine += ins; }
  mine = wave_sum(mine);
  if (lane_id() == 0 && mine) atomicAdd(&nd_sh, mine);
  __syncthreads();
  if (threadIdx.x == 0) { const uint32_t nd = nd_sh; ndist[blockIdx.x] = nd; if (nd != nb) flags[1] = 1ull; atomicMax(&flags[4], (unsigned long long)nd); }
}

// ---- repeated build keys: the distinct keys of a partition become groups (numbered in slot order), the partition's build rows a CSR.
// grec[gstart_p + g] = (key, global group number) -- what the LDS tables of the probe are built from; grp_start / grp_cnt index csr_rows
__global__ void __launch_bounds__(PJ_NT) k_pj_groups(const RpRec12* brec, const uint32_t* bstart, const uint32_t* gbase /*[P + 1] exclusive scan of ndist*/, RpRec12* grec, uint32_t* grp_start, uint32_t* grp_cnt,
                                                     uint32_t* csr_rows, unsigned long long* flags) {
  extern __shared__ uint32_t pj_lds[];
  constexpr int SB = 14; constexpr uint32_t S = 1u << SB, M = S - 1;
  uint32_t* tab = pj_lds; uint16_t* scnt = (uint16_t*)(pj_lds + S);       // rows of the key in slot s, then its CSR start inside the partition
  __shared__ uint32_t wsum[2][PJ_NW];
  const int p = blockIdx.x; const uint32_t b0 = bstart[p], nb = bstart[p + 1] - b0;       // nb <= PJ_DUP_MAX_ROWS (host checked)
  for (uint32_t s = threadIdx.x; s < S; s += PJ_NT) tab[s] = PJ_EMPTY;
  for (uint32_t s = threadIdx.x; s < S / 2; s += PJ_NT) ((uint32_t*)scnt)[s] = 0;
  __syncthreads();
  constexpr int RPT = (int)PJ_DUP_MAX_ROWS / PJ_NT;
  uint16_t slot[RPT], ord[RPT]; uint32_t row[RPT];
#pragma unroll
  for (int r = 0; r < RPT; r++) {
    const uint32_t j = threadIdx.x + r * PJ_NT; slot[r] = 0; ord[r] = 0; row[r] = 0;
    if (j < nb) {
      const RpRec12 kj = brec[b0 + j]; uint32_t fs; bool ins; pj_insert(tab, M, SB, brec + b0, j, kj, &fs, &ins);
      slot[r] = (uint16_t)fs; row[r] = kj.row;
      // ordinal inside the group: a 16-bit counter of a 32-bit LDS word
      const uint32_t old = atomicAdd((uint32_t*)scnt + (fs >> 1), 1u << ((fs & 1) * 16)); ord[r] = (uint16_t)(old >> ((fs & 1) * 16));
    }
  }
  __syncthreads();
  // slot order -> group numbers and CSR starts: thread t owns slots [16 t, 16 t + 16)
  constexpr int SPT = (int)(S / PJ_NT); uint32_t c[SPT]; uint32_t ng = 0, nr = 0;
#pragma unroll
  for (int x = 0; x < SPT; x++) { const uint32_t s = threadIdx.x * SPT + x; c[x] = tab[s] != PJ_EMPTY ? scnt[s] : 0; ng += c[x] != 0; nr += c[x]; }
  const uint32_t ig = wave_inclusive_sum(ng), ir = wave_inclusive_sum(nr);
  if (lane_id() == 63) { wsum[0][threadIdx.x >> 6] = ig; wsum[1][threadIdx.x >> 6] = ir; }
  __syncthreads();
  uint32_t g = ig - ng, st = ir - nr; for (int w = 0; w < (int)(threadIdx.x >> 6); w++) { g += wsum[0][w]; st += wsum[1][w]; }
  const uint32_t g0 = gbase[p]; uint32_t big = 0;
#pragma unroll
  for (int x = 0; x < SPT; x++) if (c[x]) {
    const uint32_t s = threadIdx.x * SPT + x; const RpRec12 rep = brec[b0 + (tab[s] & PJ_IDX_MASK)];
    grec[g0 + g] = RpRec12{ rep.lo, rep.hi, g0 + g }; grp_start[g0 + g] = b0 + st; grp_cnt[g0 + g] = c[x];
    scnt[s] = (uint16_t)st; big |= c[x] > PJ_MAX_GROUP;
    g++; st += c[x];
  }
  if (big) flags[5] = 1ull;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RPT; r++) { const uint32_t j = threadIdx.x + r * PJ_NT; if (j < nb) csr_rows[b0 + scnt[slot[r]] + ord[r]] = row[r]; }
}
// rows of a group in ascending order (the LDS atomics above hand the ordinals out in arrival order): one thread per group, insertion sort
__global__ void __launch_bounds__(BLOCK) k_pj_sort_groups(const uint32_t* grp_start, const uint32_t* grp_cnt, int64_t ngroups, uint32_t* csr_rows) {
  const int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (g >= ngroups) return;
  const uint32_t c = grp_cnt[g]; if (c < 2 || c > PJ_MAX_GROUP) return;
  uint32_t* r = csr_rows + grp_start[g];
  for (uint32_t i = 1; i < c; i++) { const uint32_t v = r[i]; uint32_t j = i; while (j && r[j - 1] > v) { r[j] = r[j - 1]; j--; } r[j] = v; }
}

// ---------------------------------------------------------------------------------------------------------------- probe
// One workgroup per partition: LDS table from the partition's build records (or group records), then every wave streams its slice of the partition's
// probe records: wave w owns the chunks [w K, (w + 1) K) of 2^PJ_CHS probe rows; tiles lie in row order inside a partition and chunk bounds are tile
// bounds, so the slice is the contiguous record range between two offsets of the partition pass.  Hits leave in record order behind the slice's first
// record (hits <= records): hits[pstart[p] + o] = (probe row << 32 | ref), hstart[p][c] = o of chunk c's first hit, send[p][w] = end of wave w's hits.
__global__ void __launch_bounds__(PJ_NT, 8) k_pj_join(const RpRec12* brec, const uint32_t* bstart, const RpRec12* prec, const uint32_t* pstart, int sbits, int NC, uint32_t P,
                                                     const uint32_t* pre, const uint32_t* cpre, int64_t ntiles, uint64_t* hits, uint32_t* hstart /*[P][NC]*/, uint32_t* send /*[P][PJ_NW]*/) {
  extern __shared__ uint4 pj_tab4[];                   // 16-byte aligned: one ds_read_b128 per group
  uint32_t* const tab = (uint32_t*)pj_tab4;
  const uint32_t S = 1u << sbits, M = S - 1;
  const int p = blockIdx.x; const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (uint32_t s = threadIdx.x; s < S; s += PJ_NT) tab[s] = PJ_EMPTY;
  __syncthreads();
  const uint32_t b0 = bstart[p], nb = bstart[p + 1] - b0;
  const RpRec12* br = brec + b0;
  for (uint32_t j = threadIdx.x; j < nb; j += PJ_NT) { uint32_t fs; bool ins; pj_insert(tab, M, sbits, br, j, br[j], &fs, &ins); }
  __syncthreads();
  const uint32_t q0 = pstart[p], q1 = pstart[p + 1];
  const int K = (NC + PJ_NW - 1) / PJ_NW, ca = wave * K, cb = ca + K < NC ? ca + K : NC;
  uint32_t* const hs = hstart + (size_t)p * NC;
  if (ca >= NC) { if (lane == 0) send[(size_t)p * PJ_NW + wave] = q1 - q0; return; }
  // first record of chunk c in this partition = offset of tile c << (PJ_CHS - 13)
  auto rec_of_chunk = [&](int c) -> uint32_t { const int64_t t = (int64_t)c << (PJ_CHS - 13); if (t >= ntiles) return q1; return q0 + pre[t * (int64_t)P + p] + cpre[(t / PJ_G) * (int64_t)P + p]; };
  const uint32_t r0 = rec_of_chunk(ca), r1 = cb < NC ? rec_of_chunk(cb) : q1;
  uint32_t run = r0 - q0;
  int clast = ca - 1;           // wave-uniform: the last chunk whose start has been written
  for (uint32_t i0 = r0 + lane; i0 - lane < r1; i0 += WAVE * PJ_U) {
    RpRec12 rc[PJ_U]; bool on[PJ_U];
#pragma unroll
    for (int u = 0; u < PJ_U; u++) { const uint32_t i = i0 + u * WAVE; on[u] = i < r1; rc[u] = prec[on[u] ? i : r1 - 1]; }
    // first group of every row: PJ_U independent 16-byte LDS reads, straight-line; the few rows whose first group is full of other keys go on in the loop below
    uint32_t s[PJ_U], tagsh[PJ_U], cand[PJ_U], pos[PJ_U]; uint4 v[PJ_U]; bool walking[PJ_U]; bool more = false;
#pragma unroll
    for (int u = 0; u < PJ_U; u++) { pj_hash(rc[u].lo, rc[u].hi, M, sbits, &s[u], &tagsh[u]); v[u] = pj_tab4[s[u] >> 2]; }
#pragma unroll
    for (int u = 0; u < PJ_U; u++) { const bool done = pj_group(v[u], tagsh[u], s[u], &cand[u], &pos[u]); walking[u] = on[u] & !done; cand[u] = on[u] ? cand[u] : PJ_EMPTY; more |= walking[u]; }
    while (__ballot(more)) {
      more = false;
#pragma unroll
      for (int u = 0; u < PJ_U; u++) if (walking[u]) { s[u] = (s[u] + 4) & M; walking[u] = !pj_group(pj_tab4[s[u] >> 2], tagsh[u], s[u], &cand[u], &pos[u]); more |= walking[u]; }
    }
    RpRec12 vb[PJ_U];
#pragma unroll
    for (int u = 0; u < PJ_U; u++) { vb[u] = RpRec12{0, 0, 0}; if (cand[u] != PJ_EMPTY) vb[u] = br[cand[u] & PJ_IDX_MASK]; }
    uint32_t href[PJ_U]; bool hit[PJ_U];
#pragma unroll
    for (int u = 0; u < PJ_U; u++) {
      hit[u] = false; href[u] = 0;
      if (cand[u] == PJ_EMPTY) continue;
      if (vb[u].lo == rc[u].lo && vb[u].hi == rc[u].hi) { hit[u] = true; href[u] = vb[u].row; continue; }
      // a different key with the same 17-bit tag (2^-17 per occupied slot passed): keep walking, verifying every tag match
      uint32_t s1 = (pos[u] + 1) & M, c = tab[s1];
      while (c != PJ_EMPTY) {
        if ((c & ~PJ_IDX_MASK) == tagsh[u]) { const RpRec12 w = br[c & PJ_IDX_MASK]; if (w.lo == rc[u].lo && w.hi == rc[u].hi) { hit[u] = true; href[u] = w.row; break; } }
        s1 = (s1 + 1) & M; c = tab[s1];
      }
    }
#pragma unroll
    for (int u = 0; u < PJ_U; u++) {
      const uint64_t b = __ballot(hit[u]);
      const uint32_t o = run + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
      if (hit[u]) hits[(size_t)q0 + o] = ((uint64_t)rc[u].row << 32) | href[u];
      // the first record of a chunk writes the start of every chunk since the previous record's
      const int c = on[u] ? (int)(rc[u].row >> PJ_CHS) : cb;
      int cp = __shfl_up(c, 1, 64); if (lane == 0) cp = clast;
      if (on[u] && c != cp) for (int x = cp + 1; x <= c; x++) hs[x] = o;
      clast = __shfl(c, 63, 64);
      if (clast >= cb) { const uint64_t onb = __ballot(on[u]); clast = onb ? __shfl(c, 63 - __clzll((long long)onb), 64) : cp; clast = __shfl(clast, 0, 64); }
      run += (uint32_t)__popcll(b);
    }
  }
  if (lane == 0) { for (int x = clast + 1; x < cb; x++) hs[x] = run; send[(size_t)p * PJ_NW + wave] = run; }
}
// hstart [P][NC] -> hT [NC][P] (starts) + lT [NC][P] (lengths: next start, or the owning wave's end, minus start); hits per chunk by atomics
__global__ void __launch_bounds__(PJ_NT) k_pj_transpose(const uint32_t* hstart, const uint32_t* send, int P, int NC, uint32_t* hT, uint16_t* lT, uint32_t* ctot) {
  __shared__ uint32_t ts[32][33], tl[32][33];
  const int c0 = blockIdx.x * 32, p0 = blockIdx.y * 32; const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int K = (NC + PJ_NW - 1) / PJ_NW;
  { const int p = p0 + ty, c = c0 + tx;
    if (p < P && c < NC) { const uint32_t a = hstart[(size_t)p * NC + c]; const bool last = (c + 1) % K == 0 || c + 1 == NC; const uint32_t b = last ? send[(size_t)p * PJ_NW + c / K] : hstart[(size_t)p * NC + c + 1]; ts[ty][tx] = a; tl[ty][tx] = b - a; }
    else { ts[ty][tx] = 0; tl[ty][tx] = 0; } }
  __syncthreads();
  { const int c = c0 + ty, p = p0 + tx;
    uint32_t l = tl[tx][ty];
    if (c < NC && p < P) { hT[(size_t)c * P + p] = ts[tx][ty]; lT[(size_t)c * P + p] = (uint16_t)l; }       // l <= 2^PJ_CHS
#pragma unroll
    for (int d = 16; d > 0; d >>= 1) l += __shfl_xor(l, d, 64);
    if (tx == 0 && c < NC && l) atomicAdd(&ctot[c], l); }
}
// restore probe order: one workgroup per chunk of 2^PJ_CHS probe rows.  Hit i of the chunk is found in its partition's list by a search over the prefix of
// the 2048 run lengths; every probe row has at most one hit, so its rank among the chunk's hits = set bits below it in a bitmap of the chunk's rows.  The
// hits leave through an LDS window of WIN ranks (a chunk usually fits one window: its hits then stay in registers between the two passes).
constexpr int PJ_K6 = 6, PJ_WIN = PJ_NT * PJ_K6;      // tried: 512 threads x 8 hits (32 KB window, three workgroups per CU instead of two): 0.393 against 0.396 ms -- not the limiter
template <bool GROUPS, int MP = (int)PJ_MAX_P>
__global__ void __launch_bounds__(PJ_NT) k_pj_restore(const uint64_t* hits, const uint32_t* pstart, const uint32_t* hT, const uint16_t* lT, int P, int NC, const uint32_t* coff,
                                                     uint32_t* out_probe, uint64_t* out_build, uint32_t* out_ref) {
  constexpr int NWORDS = 1 << (PJ_CHS - 5);
  __shared__ uint32_t bits[NWORDS]; __shared__ uint32_t pref[NWORDS]; __shared__ uint32_t rsrc[MP]; __shared__ uint32_t roff[MP + 1]; __shared__ uint32_t wsum[PJ_NW]; __shared__ uint64_t stage[PJ_WIN];
  const int64_t per = (NC + 7) / 8, cc = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (cc >= NC || (int64_t)(blockIdx.x >> 3) >= per) return;
  const int c = (int)cc; const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int x = threadIdx.x; x < NWORDS; x += PJ_NT) bits[x] = 0;
  uint32_t tot;
  { constexpr int PER = MP / PJ_NT; uint32_t len[PER]; uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) { const int p = threadIdx.x * PER + j; len[j] = 0; if (p < P) { len[j] = lT[(size_t)c * P + p]; rsrc[p] = pstart[p] + hT[(size_t)c * P + p]; } s += len[j]; }
    const uint32_t inc = wave_inclusive_sum(s);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - s; tot = 0; for (int w = 0; w < PJ_NW; w++) { if (w < wave) run += wsum[w]; tot += wsum[w]; }
#pragma unroll
    for (int j = 0; j < PER; j++) { const int p = threadIdx.x * PER + j; if (p < P) roff[p] = run; run += len[j]; }
    if (threadIdx.x == 0) roff[P] = tot; }
  __syncthreads();
  auto src_of = [&](uint32_t i) -> size_t { int lo = 0, hi = P;      // last run with roff <= i (empty runs share their successor's offset)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (roff[mid] <= i) lo = mid; else hi = mid; }
    return (size_t)rsrc[lo] + (i - roff[lo]); };
  constexpr uint32_t rmask = (1u << PJ_CHS) - 1u;
  const bool one = tot <= (uint32_t)PJ_WIN;
  uint64_t hv[PJ_K6];
  for (uint32_t i0 = 0; i0 < tot; i0 += PJ_WIN) {
    size_t a[PJ_K6];
#pragma unroll
    for (int k = 0; k < PJ_K6; k++) { const uint32_t i = i0 + k * PJ_NT + threadIdx.x; a[k] = i < tot ? src_of(i) : (size_t)0; }
#pragma unroll
    for (int k = 0; k < PJ_K6; k++) { const uint32_t i = i0 + k * PJ_NT + threadIdx.x; hv[k] = i < tot ? hits[a[k]] : ~0ull; }
#pragma unroll
    for (int k = 0; k < PJ_K6; k++) if (hv[k] != ~0ull) { const uint32_t rr = (uint32_t)(hv[k] >> 32) & rmask; atomicOr(&bits[rr >> 5], 1u << (rr & 31)); }
  }
  __syncthreads();
  { const uint32_t cw = threadIdx.x < (unsigned)NWORDS ? __popc(bits[threadIdx.x]) : 0; const uint32_t inc2 = wave_inclusive_sum(cw);
    __syncthreads();
    if (lane == 63) wsum[wave] = inc2;
    __syncthreads();
    uint32_t r2 = inc2 - cw; for (int w = 0; w < wave; w++) r2 += wsum[w];
    if (threadIdx.x < (unsigned)NWORDS) pref[threadIdx.x] = r2; }
  __syncthreads();
  const uint32_t o0 = coff[c];
  for (uint32_t lo = 0; lo < tot; lo += PJ_WIN) {            // window of ranks [lo, lo + WIN)
    if (lo) __syncthreads();
    if (one) {
#pragma unroll
      for (int k = 0; k < PJ_K6; k++) if (hv[k] != ~0ull) { const uint32_t rr = (uint32_t)(hv[k] >> 32) & rmask; stage[pref[rr >> 5] + __popc(bits[rr >> 5] & ((1u << (rr & 31)) - 1u))] = hv[k]; }
    } else {
      for (uint32_t i0 = 0; i0 < tot; i0 += PJ_WIN) {
        size_t a[PJ_K6]; uint64_t h2[PJ_K6];
#pragma unroll
        for (int k = 0; k < PJ_K6; k++) { const uint32_t i = i0 + k * PJ_NT + threadIdx.x; a[k] = i < tot ? src_of(i) : (size_t)0; }
#pragma unroll
        for (int k = 0; k < PJ_K6; k++) { const uint32_t i = i0 + k * PJ_NT + threadIdx.x; h2[k] = i < tot ? hits[a[k]] : ~0ull; }
#pragma unroll
        for (int k = 0; k < PJ_K6; k++) if (h2[k] != ~0ull) { const uint32_t rr = (uint32_t)(h2[k] >> 32) & rmask; const uint32_t rk = pref[rr >> 5] + __popc(bits[rr >> 5] & ((1u << (rr & 31)) - 1u)) - lo; if (rk < (uint32_t)PJ_WIN) stage[rk] = h2[k]; }
      }
    }
    __syncthreads();
    const uint32_t m = tot - lo < (uint32_t)PJ_WIN ? tot - lo : (uint32_t)PJ_WIN;
    for (uint32_t i = threadIdx.x; i < m; i += PJ_NT) {
      const uint64_t v = stage[i]; out_probe[o0 + lo + i] = (uint32_t)(v >> 32);
      if (GROUPS) out_ref[o0 + lo + i] = (uint32_t)v; else out_build[o0 + lo + i] = (uint32_t)v;
    }
  }
}
// repeated build keys: every matched probe row emits its group's rows in build order (the general path's k_probe_expand over the partitioned CSR)
__global__ void __launch_bounds__(BLOCK) k_pj_group_counts(const uint32_t* ref, int64_t m, const uint32_t* grp_cnt, uint32_t* cnt) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (i < m) cnt[i] = grp_cnt[ref[i]];
}
// Load-balanced: a wave owns 64 consecutive matches; its pairs [offsets[first], offsets[last] + cnt[last]) are consecutive in the output, so the lanes walk that range 64 pairs
// at a time (consecutive lanes -> consecutive output slots) and find their match by a 6-step search over the wave's 64 offsets held in registers (shuffles).  One thread per match
// with a loop over its group wrote `count`-strided runs: 0.8 ms for 30 M pairs of 5-row groups against 0.25 ms.
__global__ void __launch_bounds__(BLOCK) k_pj_expand(const uint32_t* rows, const uint32_t* ref, const uint64_t* offsets, int64_t m, const uint32_t* grp_start, const uint32_t* grp_cnt, const uint32_t* csr_rows,
                                                    uint64_t* out_build, uint32_t* out_probe) {
  const int lane = lane_id(); const int64_t w0 = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) - lane;      // first match of this wave
  if (w0 >= m) return;
  const int64_t i = w0 + lane; const bool on = i < m;
  const uint32_t g = on ? ref[i] : 0u, c = on ? grp_cnt[g] : 0u, st = on ? grp_start[g] : 0u, j = on ? rows[i] : 0u;
  const uint64_t o = on ? offsets[i] : 0ull;
  const uint64_t base = __shfl((long long)o, 0, 64);                                   // offsets are non-decreasing: lane 0 holds the wave's first pair
  const uint32_t rel = (uint32_t)(o - base);                                            // a wave's pairs fit 32 bits (64 matches x group size)
  const int last = 63 - __clzll((long long)ballot64(on));
  const uint32_t total = (uint32_t)__shfl((int)(rel + c), last, 64);
  for (uint32_t k0 = 0; k0 < total; k0 += WAVE) {
    const uint32_t k = k0 + lane;
    // the match whose pair range holds k: the last lane with rel <= k (lanes past `last` hold rel = 0 and c = 0: clamp the search to [0, last])
    int lo = 0, hi = last;
#pragma unroll
    for (int step = 0; step < 6; step++) { const int mid = (lo + hi + 1) >> 1; const uint32_t r = (uint32_t)__shfl((int)rel, mid, 64); if (r <= k) lo = mid; else hi = mid - 1; }
    // groups of zero rows cannot occur (a match has >= 1 row), so rel is strictly increasing over the active lanes and lo is exact
    const uint32_t r0 = (uint32_t)__shfl((int)rel, lo, 64), s0 = (uint32_t)__shfl((int)st, lo, 64), p0 = (uint32_t)__shfl((int)j, lo, 64);
    if (k < total) { out_build[base + k] = csr_rows[s0 + (k - r0)]; out_probe[base + k] = p0; }
  }
}

// ---------------------------------------------------------------------------------------------------------------- host
static void pj_set_lds(const void* fn, size_t bytes) { if (bytes > 48 * 1024) HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)); }     // per device, per call: cheap and free of process-wide state

static bool pj_key_type_ok(const dfgpu_array* a) {
  switch (a->type) {
    case DFGPU_INT8: case DFGPU_INT16: case DFGPU_INT32: case DFGPU_INT64: case DFGPU_DATE32:
    case DFGPU_UINT8: case DFGPU_UINT16: case DFGPU_UINT32: case DFGPU_UINT64: return true;
    default: return false;
  }
}

// partition the selected, non-NULL rows of an integer key column by rp_pid(mix64(widened key)) into 12-byte (key, row) records
template <typename H>
static PjOffsets pj_partition_t(dfgpu_ctx* ctx, H hs, int64_t n, uint32_t P, RpRec12* recs, uint64_t* d_total, const char* th, const char* ts, const char* tw) {
  PjOffsets o; o.ntiles = n ? (n + PJ_TILE - 1) / PJ_TILE : 1; o.nchunks = (o.ntiles + PJ_G - 1) / PJ_G;
  o.pre = alloc_buffer(ctx, (size_t)o.ntiles * P * 4); o.cpre = alloc_buffer(ctx, (size_t)o.nchunks * P * 4); o.pstart = alloc_buffer(ctx, (size_t)(P + 1) * 4);
  BufferPtr ptot = alloc_buffer(ctx, (size_t)P * 4);
  { KernelTimer kt_(ctx, th);
    const size_t lds = (size_t)P * PJ_G * 2;
    pj_set_lds((const void*)k_pj_hist<H>, lds);
    hipLaunchKernelGGL((k_pj_hist<H>), dim3((unsigned)o.nchunks), dim3(PJ_NT), lds, ctx->stream, hs, n, P, o.ntiles, (uint32_t*)o.pre->ptr, (uint32_t*)o.cpre->ptr);
    KERNEL_CHECK(); }
  { KernelTimer kt_(ctx, ts);
    hipLaunchKernelGGL(k_pj_chunk_prefix, dim3((P + 63) / 64), dim3(PJ_NT), 0, ctx->stream, (uint32_t*)o.cpre->ptr, o.nchunks, P, (uint32_t*)ptot->ptr);
    if (P > PJ_MAX_P) hipLaunchKernelGGL((k_pj_pstart<(int)PJ_MAX_P_BIG>), dim3(1), dim3(PJ_NT), 0, ctx->stream, (const uint32_t*)ptot->ptr, P, (uint32_t*)o.pstart->ptr, d_total);
    else hipLaunchKernelGGL((k_pj_pstart<(int)PJ_MAX_P>), dim3(1), dim3(PJ_NT), 0, ctx->stream, (const uint32_t*)ptot->ptr, P, (uint32_t*)o.pstart->ptr, d_total);
    KERNEL_CHECK(); }
  if (n) { KernelTimer kt_(ctx, tw);
    const size_t lds = (size_t)P * 8 + (size_t)PJ_TILE / 2 * 12 + 16;
    if (P > PJ_MAX_P) { pj_set_lds((const void*)k_pj_scatter<H, (int)PJ_MAX_P_BIG>, lds);
      hipLaunchKernelGGL((k_pj_scatter<H, (int)PJ_MAX_P_BIG>), dim3((unsigned)(((o.ntiles + 7) / 8) * 8)), dim3(PJ_NT), lds, ctx->stream, hs, n, P, o.ntiles, (const uint32_t*)o.pre->ptr, (const uint32_t*)o.cpre->ptr,
                         (const uint32_t*)o.pstart->ptr, recs); }
    else { pj_set_lds((const void*)k_pj_scatter<H>, lds);
    hipLaunchKernelGGL((k_pj_scatter<H>), dim3((unsigned)(((o.ntiles + 7) / 8) * 8)), dim3(PJ_NT), lds, ctx->stream, hs, n, P, o.ntiles, (const uint32_t*)o.pre->ptr, (const uint32_t*)o.cpre->ptr,
                       (const uint32_t*)o.pstart->ptr, recs); }
    KERNEL_CHECK(); }
  return o;
}
static PjOffsets pj_partition(dfgpu_ctx* ctx, const dfgpu_array* key, const uint64_t* mask, uint32_t P, RpRec12* recs, uint64_t* d_total, const char* th, const char* ts, const char* tw) {
  const uint64_t* valid = key->validity ? (const uint64_t*)key->validity->ptr : nullptr; const int64_t n = key->length;
#define PJ_PART(T) return pj_partition_t(ctx, RpHashInt<T>{ (const T*)key->values->ptr, valid, mask }, n, P, recs, d_total, th, ts, tw)
  switch (key->type) {
    case DFGPU_INT8: PJ_PART(int8_t); case DFGPU_INT16: PJ_PART(int16_t); case DFGPU_INT32: case DFGPU_DATE32: PJ_PART(int32_t);
    case DFGPU_UINT8: PJ_PART(uint8_t); case DFGPU_UINT16: PJ_PART(uint16_t); case DFGPU_UINT32: PJ_PART(uint32_t);
    default: PJ_PART(int64_t);          // INT64 / UINT64: the same 64-bit pattern
  }
#undef PJ_PART
}

// Hashed mode: the key columns are anything keyset_hash takes (several columns, Utf8, dictionaries, NULLs that match under null_equals_null).  The record key is the
// 64-bit keyset hash; everything downstream -- partition, LDS tables, groups of repeated keys -- compares record keys, so two different key values with one hash are simply
// rows of one "key" there.  That is sorted out at the end: k_pj_verify compares every emitted pair in the columns themselves and the pairs that fail are dropped.
struct PjHashKeys {
  KeySet ks; const uint64_t* mask; int null_eq; uint64_t hmask;        // hmask: all ones; fewer bits (option "join_partitioned_hash_mask", tests) make different keys share a hash
  __device__ inline bool operator()(int64_t i, uint32_t P, uint32_t* pid, uint64_t* key) const {
    bool an; const uint64_t h = keyset_hash(ks, i, 0, &an) & hmask;
    *key = h; *pid = rp_pid(mix64(h), P);
    return (!mask || bit_get(mask, i)) && (!an || null_eq);
  }
};
__global__ void __launch_bounds__(BLOCK) k_pj_verify(KeySet bks, KeySet pks, const uint64_t* __restrict__ bidx, const uint32_t* __restrict__ pidx, int64_t m, int null_eq, uint64_t* __restrict__ keep, unsigned long long* nfail) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const bool ok = i < m && keyset_equal(bks, (int64_t)bidx[i], pks, (int64_t)pidx[i], null_eq != 0);
  const uint64_t w = ballot64(ok);
  if (lane_id() == 0 && i < m) { keep[i >> 6] = w; const int live = (int)min((int64_t)64, m - i); const int bad = live - __popcll(w); if (bad) atomicAdd(nfail, (unsigned long long)bad); }
}
static PjOffsets pj_partition_keys(dfgpu_ctx* ctx, const KeySet& ks, int64_t n, const uint64_t* mask, bool null_eq, uint32_t P, RpRec12* recs, uint64_t* d_total, const char* th, const char* ts, const char* tw) {
  return pj_partition_t(ctx, PjHashKeys{ ks, mask, null_eq ? 1 : 0, ctx->join_partitioned_hash_mask }, n, P, recs, d_total, th, ts, tw);
}
bool pj_hashed_candidate(dfgpu_ctx* ctx, const dfgpu_join_table* t) {
  if (!ctx->join_partitioned || !ctx->join_partitioned_hashed || ctx->force_hash_collisions || t->n_build < ctx->join_partitioned_min_build) return false;
  return !(t->nkeys == 1 && !t->null_equals_null && pj_key_type_ok(t->keys[0]));      // those are the integer mode's (which asks pj_domain_is_sparse first)
}

bool pj_build(dfgpu_ctx* ctx, dfgpu_join_table* t) {
  const int64_t n = t->n_build;
  if (!ctx->join_partitioned || ctx->force_hash_collisions) return false;
  if (n < ctx->join_partitioned_min_build || n > 0xFFFFFFF0ll) return false;
  const dfgpu_array* key0 = t->keys[0];
  const bool hashed = !(t->nkeys == 1 && !t->null_equals_null && pj_key_type_ok(key0));
  if (hashed && !ctx->join_partitioned_hashed) return false;
  int64_t cap = ctx->join_partition_rows; if (cap < 16) cap = 16; if (cap > 14000) cap = 14000;
  const int64_t per = std::min<int64_t>(cap, 7300);      // <= 8192 keys: a 64 KB table at load <= 1/2, two workgroups per CU
  int64_t P64 = (n + per - 1) / per; if (P64 < 1) P64 = 1;
  if (P64 > PJ_MAX_P) {       // up to `cap` rows per partition: 128 KB tables, one workgroup per CU; beyond 2048 x cap rows up to 4096 partitions (the MP = 4096 instantiations)
    if ((n + PJ_MAX_P - 1) / PJ_MAX_P <= cap) P64 = PJ_MAX_P;
    else { P64 = (n + cap - 1) / cap; P64 = (P64 + 255) / 256 * 256; if (P64 > PJ_MAX_P_BIG || !ctx->join_partitioned_big) return false; }
  }
  else if (P64 > 2 * ctx->num_cus) P64 = std::min<int64_t>(PJ_MAX_P, (P64 + 2 * ctx->num_cus - 1) / (2 * ctx->num_cus) * (2 * ctx->num_cus));      // whole rounds of two workgroups per CU
  auto part = std::make_unique<PartitionedBuild>();
  part->P = (uint32_t)P64;
  BufferPtr recs = alloc_buffer(ctx, (size_t)(n + 1) * 12), ndist = alloc_buffer(ctx, (size_t)(P64 + 1) * 4);
  zero_scratch(ctx);
  const uint64_t* bmask = t->build_mask ? (const uint64_t*)t->build_mask->ptr : nullptr;
  part->hashed = hashed;
  PjOffsets off = hashed ? pj_partition_keys(ctx, t->ks, n, bmask, t->null_equals_null, part->P, (RpRec12*)recs->ptr, ctx->d_scratch64 + 3, "pj_build_hist", "pj_build_scan", "pj_build_scatter")
                         : pj_partition(ctx, key0, bmask, part->P, (RpRec12*)recs->ptr, ctx->d_scratch64 + 3, "pj_build_hist", "pj_build_scan", "pj_build_scatter");
  uint64_t max_rows = 0, dup = 0, over = 0, moved = 0, max_dist = 0;
  for (int max_sbits = (n + P64 - 1) / P64 <= 7500 ? 14 : PJ_MAX_SBITS;; max_sbits = PJ_MAX_SBITS) {      // 64 KB of LDS (two workgroups per CU) when the average partition leaves room for its spread
    { KernelTimer kt_(ctx, "pj_build_check");
      const size_t lds = (size_t)(1u << max_sbits) * 4;
      pj_set_lds((const void*)k_pj_check, lds);
      hipLaunchKernelGGL(k_pj_check, dim3(part->P), dim3(PJ_NT), lds, ctx->stream, (const RpRec12*)recs->ptr, (const uint32_t*)off.pstart->ptr, max_sbits, (unsigned long long*)ctx->d_scratch64, (uint32_t*)ndist->ptr);
      KERNEL_CHECK(); }
    const uint64_t* h = read_scratch_range(ctx, 0, 7);
    ctx->count_sync("sync:pj_build_check");
    max_rows = h[0]; dup = h[1]; over = h[2]; moved = h[3]; max_dist = h[4];
    if (!h[6] || max_sbits == PJ_MAX_SBITS) break;
    HIP_CHECK(hipMemsetAsync(ctx->d_scratch64, 0, 24, ctx->stream)); HIP_CHECK(hipMemsetAsync(ctx->d_scratch64 + 4, 0, 24, ctx->stream));       // a partition beyond 8192 rows: once more with the 128 KB table (slot 3, the rows moved, stays)
  }
  if (over) return false;
  part->rows = (int64_t)moved; part->starts = off.pstart; part->recs = recs;
  uint64_t max_keys = max_rows;
  if (dup) {
    if (max_rows > PJ_DUP_MAX_ROWS) return false;        // the CSR kernel keeps a partition's records, counters and cursors in LDS
    KernelTimer kt_(ctx, "pj_build_groups");
    exclusive_scan_u32_inplace32(ctx, (uint32_t*)ndist->ptr, (int64_t)P64 + 1, ctx->d_scratch64 + 6);     // ndist[P] is scratch: its exclusive prefix is the total
    BufferPtr grec = alloc_buffer(ctx, (size_t)(moved + 1) * 12), gstart = alloc_buffer(ctx, (size_t)(moved + 1) * 4), gcnt = alloc_buffer(ctx, (size_t)(moved + 1) * 4), csr = alloc_buffer(ctx, (size_t)(moved + 1) * 4);
    const size_t lds = (size_t)(1u << 14) * 4 + (size_t)(1u << 14) * 2;
    pj_set_lds((const void*)k_pj_groups, lds);
    hipLaunchKernelGGL(k_pj_groups, dim3(part->P), dim3(PJ_NT), lds, ctx->stream, (const RpRec12*)recs->ptr, (const uint32_t*)off.pstart->ptr, (const uint32_t*)ndist->ptr, (RpRec12*)grec->ptr, (uint32_t*)gstart->ptr,
                       (uint32_t*)gcnt->ptr, (uint32_t*)csr->ptr, (unsigned long long*)ctx->d_scratch64);
    KERNEL_CHECK();
    // total groups = ndist[P] after the scan; read it together with the big-group flag
    HIP_CHECK(hipMemcpyAsync(ctx->h_pinned + 8, (const uint32_t*)ndist->ptr + P64, 4, hipMemcpyDeviceToHost, ctx->stream));
    const uint64_t big = read_scratch(ctx, 5);
    const int64_t ng = (int64_t)(uint32_t)ctx->h_pinned[8];
    if (big) return false;                               // a key with more rows than one thread sorts: the general path's CSR
    if (ng) hipLaunchKernelGGL(k_pj_sort_groups, dim3(grid_for(ng, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)gstart->ptr, (const uint32_t*)gcnt->ptr, ng, (uint32_t*)csr->ptr);
    KERNEL_CHECK();
    part->dups = true; part->recs = grec; part->grp_start = gstart; part->grp_cnt = gcnt; part->csr_rows = csr; part->starts = ndist;       // ndist = group offsets of the partitions
    max_keys = max_dist;
    t->mem += (int64_t)moved * 24;
  }
  int sbits = 6; while ((1ull << sbits) < 2 * max_keys && sbits < PJ_MAX_SBITS) sbits++;
  part->sbits = sbits;
  t->part = std::move(part);
  t->unique = !dup;
  t->mem += (int64_t)n * 12 + (int64_t)(P64 + 1) * 4;
  return true;
}

bool pj_probe_eligible(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* const* probe_keys, int32_t nkeys, int64_t n) {
  if (!t->part || !ctx->join_partitioned || n < ctx->join_partitioned_min_probe || n > 0xFFFF0000ll) return false;
  if (t->part->hashed) return nkeys == t->nkeys;       // the caller has checked the column types against the build's; dictionaries hash by value
  return nkeys == 1 && probe_keys[0]->type == t->keys[0]->type;          // same physical integer type, no dictionary
}

// hashed mode: drop the pairs whose columns differ (two key values with one 64-bit hash; none on ordinary data -- the count decides whether anything is compacted)
static void pj_verify(dfgpu_ctx* ctx, const dfgpu_join_table* t, const KeySet& pks, ArrayHolder& ob, ArrayHolder& op) {
  const int64_t m = ob.get()->length; if (!m) return;
  KernelTimer kt_(ctx, "pj_verify");
  BufferPtr keep = alloc_buffer(ctx, bitmap_bytes(m));
  HIP_CHECK(hipMemsetAsync(ctx->d_scratch64 + 11, 0, 8, ctx->stream));
  hipLaunchKernelGGL(k_pj_verify, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, t->ks, pks, (const uint64_t*)ob.get()->values->ptr, (const uint32_t*)op.get()->values->ptr, m, t->null_equals_null ? 1 : 0,
                     (uint64_t*)keep->ptr, (unsigned long long*)(ctx->d_scratch64 + 11));
  KERNEL_CHECK();
  const uint64_t bad = read_scratch(ctx, 11);
  ctx->count_sync("sync:pj_verify");
  if (!bad) return;
  ArrayHolder idx(mask_to_indices_impl(ctx, (const uint64_t*)keep->ptr, m));
  ArrayHolder nb(take_impl(ctx, ob.get(), idx.get()->values->ptr, 4, nullptr, idx.get()->length)), np(take_impl(ctx, op.get(), idx.get()->values->ptr, 4, nullptr, idx.get()->length));
  dfgpu_array_release(ob.a); ob.a = nb.release(); dfgpu_array_release(op.a); op.a = np.release();
}
void pj_probe(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* const* probe_keys, int32_t nkeys, const uint64_t* mask, dfgpu_array** out_build, dfgpu_array** out_probe) {
  const PartitionedBuild& part = *t->part;
  const dfgpu_array* pk = probe_keys[0];
  const KeySet pks = part.hashed ? make_keyset(probe_keys, nkeys) : KeySet{};
  const int64_t n = pk->length; const uint32_t P = part.P;
  const int NC = (int)((n + (1ll << PJ_CHS) - 1) >> PJ_CHS);
  BufferPtr recs = alloc_buffer(ctx, (size_t)(n + 1) * 12);
  PjOffsets off = part.hashed ? pj_partition_keys(ctx, pks, n, mask, t->null_equals_null, P, (RpRec12*)recs->ptr, ctx->d_scratch64 + 9, "pj_probe_hist", "pj_probe_scan", "pj_probe_scatter")
                              : pj_partition(ctx, pk, mask, P, (RpRec12*)recs->ptr, ctx->d_scratch64 + 9, "pj_probe_hist", "pj_probe_scan", "pj_probe_scatter");
  BufferPtr hits = alloc_buffer(ctx, (size_t)(n + 1) * 8), hstart = alloc_buffer(ctx, (size_t)P * NC * 4 + 4), send = alloc_buffer(ctx, (size_t)P * PJ_NW * 4);
  { KernelTimer kt_(ctx, "pj_join");
    const size_t lds = (size_t)(1u << part.sbits) * 4;
    pj_set_lds((const void*)k_pj_join, lds);
    hipLaunchKernelGGL(k_pj_join, dim3(P), dim3(PJ_NT), lds, ctx->stream, (const RpRec12*)part.recs->ptr, (const uint32_t*)part.starts->ptr, (const RpRec12*)recs->ptr, (const uint32_t*)off.pstart->ptr,
                       part.sbits, NC, P, (const uint32_t*)off.pre->ptr, (const uint32_t*)off.cpre->ptr, off.ntiles, (uint64_t*)hits->ptr, (uint32_t*)hstart->ptr, (uint32_t*)send->ptr);
    KERNEL_CHECK(); }
  recs.reset(); off.pre.reset(); off.cpre.reset();
  std::unique_ptr<KernelTimer> kt_(new KernelTimer(ctx, "pj_compact"));       // ends before the verification of the hashed mode, which has a timer of its own
  BufferPtr hT = alloc_buffer(ctx, (size_t)P * NC * 4 + 4), lT = alloc_buffer(ctx, (size_t)P * NC * 2 + 4), coff = alloc_buffer(ctx, (size_t)(NC + 1) * 4, true);
  hipLaunchKernelGGL(k_pj_transpose, dim3((NC + 31) / 32, (P + 31) / 32), dim3(PJ_NT), 0, ctx->stream, (const uint32_t*)hstart->ptr, (const uint32_t*)send->ptr, (int)P, NC, (uint32_t*)hT->ptr, (uint16_t*)lT->ptr, (uint32_t*)coff->ptr);
  exclusive_scan_u32_inplace32(ctx, (uint32_t*)coff->ptr, (int64_t)NC + 1, ctx->d_scratch64 + 10);
  KERNEL_CHECK();
  hstart.reset(); send.reset();
  const int64_t m = (int64_t)read_scratch(ctx, 10);        // matched probe rows
  const unsigned rgrid = (unsigned)(((NC + 7) / 8) * 8);
  if (!part.dups) {
    ArrayHolder ob(new_fixed(ctx, DFGPU_UINT64, m)), op(new_fixed(ctx, DFGPU_UINT32, m));
    if (m && P > PJ_MAX_P) hipLaunchKernelGGL((k_pj_restore<false, (int)PJ_MAX_P_BIG>), dim3(rgrid), dim3(PJ_NT), 0, ctx->stream, (const uint64_t*)hits->ptr, (const uint32_t*)off.pstart->ptr, (const uint32_t*)hT->ptr, (const uint16_t*)lT->ptr, (int)P, NC,
                                  (const uint32_t*)coff->ptr, (uint32_t*)op.get()->values->ptr, (uint64_t*)ob.get()->values->ptr, (uint32_t*)nullptr);
    else if (m) hipLaunchKernelGGL((k_pj_restore<false>), dim3(rgrid), dim3(PJ_NT), 0, ctx->stream, (const uint64_t*)hits->ptr, (const uint32_t*)off.pstart->ptr, (const uint32_t*)hT->ptr, (const uint16_t*)lT->ptr, (int)P, NC,
                                  (const uint32_t*)coff->ptr, (uint32_t*)op.get()->values->ptr, (uint64_t*)ob.get()->values->ptr, (uint32_t*)nullptr);
    KERNEL_CHECK();
    kt_.reset();
    if (part.hashed) pj_verify(ctx, t, pks, ob, op);
    op.get()->identity = op.get()->length == n;
    *out_build = ob.release(); *out_probe = op.release();
    return;
  }
  // repeated build keys: restore the order of the (probe row, group) matches, then every match emits its group
  BufferPtr rows = alloc_buffer(ctx, (size_t)(m + 1) * 4), ref = alloc_buffer(ctx, (size_t)(m + 1) * 4), cnt = alloc_buffer(ctx, (size_t)(m + 1) * 4), offs = alloc_buffer(ctx, (size_t)(m + 1) * 8);
  int64_t total = 0;
  if (m) {
    if (P > PJ_MAX_P) hipLaunchKernelGGL((k_pj_restore<true, (int)PJ_MAX_P_BIG>), dim3(rgrid), dim3(PJ_NT), 0, ctx->stream, (const uint64_t*)hits->ptr, (const uint32_t*)off.pstart->ptr, (const uint32_t*)hT->ptr, (const uint16_t*)lT->ptr, (int)P, NC,
                       (const uint32_t*)coff->ptr, (uint32_t*)rows->ptr, (uint64_t*)nullptr, (uint32_t*)ref->ptr);
    else hipLaunchKernelGGL((k_pj_restore<true>), dim3(rgrid), dim3(PJ_NT), 0, ctx->stream, (const uint64_t*)hits->ptr, (const uint32_t*)off.pstart->ptr, (const uint32_t*)hT->ptr, (const uint16_t*)lT->ptr, (int)P, NC,
                       (const uint32_t*)coff->ptr, (uint32_t*)rows->ptr, (uint64_t*)nullptr, (uint32_t*)ref->ptr);
    hipLaunchKernelGGL(k_pj_group_counts, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)ref->ptr, m, (const uint32_t*)part.grp_cnt->ptr, (uint32_t*)cnt->ptr);
    KERNEL_CHECK();
    exclusive_scan_u32(ctx, (const uint32_t*)cnt->ptr, (uint64_t*)offs->ptr, m, ctx->d_scratch64 + 8);
    total = (int64_t)read_scratch(ctx, 8);
  }
  if (total > 0xFFFFFFF0ll) fail(DFGPU_RESOURCES_EXHAUSTED, "join output of %lld rows for one probe batch; split the probe batch", (long long)total);
  ArrayHolder ob(new_fixed(ctx, DFGPU_UINT64, total)), op(new_fixed(ctx, DFGPU_UINT32, total));
  if (total) hipLaunchKernelGGL(k_pj_expand, dim3(grid_for(m, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)rows->ptr, (const uint32_t*)ref->ptr, (const uint64_t*)offs->ptr, m, (const uint32_t*)part.grp_start->ptr,
                                (const uint32_t*)part.grp_cnt->ptr, (const uint32_t*)part.csr_rows->ptr, (uint64_t*)ob.get()->values->ptr, (uint32_t*)op.get()->values->ptr);
  KERNEL_CHECK();
  kt_.reset();
  if (part.hashed) pj_verify(ctx, t, pks, ob, op);
  *out_build = ob.release(); *out_probe = op.release();
}

// ---------------------------------------------------------------- membership bitmap, probed by key range
// A dense key domain gives the build a membership bitmap (join.hip); keys that arrive in key order walk it like a stream.  Keys in NO order (a probe side behind a hash
// repartition, a table never written in key order) each touch a line of their own: 280 M probes into a 75 MB bitmap moved 36 GB of lines for 2.3 GB of keys (TPC-H Q3 at
// SF100 over shuffled tables, 6.8 ms).  Such a batch is first split by key RANGE -- record = (key - key_min) << 32 | probe row, partition = the bits above a 2^21-key slice,
// one pass of radix_partition.h -- so that the records of a partition test a 256 KB slice of the bitmap that its XCD's L2 holds; a match sets its row's bit with an atomic OR
// (the records of a partition are in no particular row order).  The sample below decides: unclustered keys, and few enough matches that their atomics stay small.
template <typename T> struct RpHashKeyRange {
  const T* keys; const uint64_t* valid; const uint64_t* mask; int64_t kmin; uint64_t range; int shift;
  __device__ inline bool operator()(int64_t i, uint32_t, uint32_t* pid, uint64_t* key) const {
    const uint64_t d = (uint64_t)((int64_t)keys[i] - kmin);
    const bool ok = d < range && (!mask || bit_get(mask, i)) && (!valid || bit_get(valid, i));
    *pid = ok ? (uint32_t)(d >> shift) : 0u; *key = (d << 32) | (uint64_t)(uint32_t)i; return ok;
  }
};
// 2048 evenly spread rows i: [0] += rows whose successor's key lies in the same 4 KB of bitmap (clustered input: nearly all), [1] += selected rows, [2] += selected rows that match
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_bp_sample(const T* keys, const uint64_t* valid, const uint64_t* mask, int64_t n, int64_t kmin, uint64_t range, const uint64_t* bitmap, unsigned long long* out) {
  constexpr int S = 2048;
  uint32_t near = 0, sel = 0, hit = 0;
  const int64_t step = (n - 1) / S;                     // n >= 2^20: step >= 512
  for (int s0 = threadIdx.x; s0 < S; s0 += BLOCK) {
    const int64_t i = (int64_t)s0 * step;
    const uint64_t d = (uint64_t)((int64_t)keys[i] - kmin), d1 = (uint64_t)((int64_t)keys[i + 1] - kmin);
    near += (d >> 15) == (d1 >> 15);
    const bool ok = d < range && (!mask || bit_get(mask, i)) && (!valid || bit_get(valid, i));
    sel += ok; hit += ok && ((bitmap[d >> 6] >> (d & 63)) & 1ull);
  }
  near = wave_sum(near); sel = wave_sum(sel); hit = wave_sum(hit);
  if (lane_id() == 0) { atomicAdd(&out[0], (unsigned long long)near); atomicAdd(&out[1], (unsigned long long)sel); atomicAdd(&out[2], (unsigned long long)hit); }
}
constexpr int BP_ROWS = 4;
__global__ void __launch_bounds__(BLOCK) k_bp_probe(const uint64_t* recs, const unsigned long long* d_total, const uint64_t* bitmap, unsigned long long* match_bits) {
  const int64_t total = (int64_t)*d_total, base = (int64_t)blockIdx.x * BLOCK * BP_ROWS + threadIdx.x;
  if (base - threadIdx.x >= total) return;
  uint64_t r[BP_ROWS], w[BP_ROWS];
#pragma unroll
  for (int q = 0; q < BP_ROWS; q++) { const int64_t i = base + (int64_t)q * BLOCK; r[q] = recs[i < total ? i : total - 1]; }
#pragma unroll
  for (int q = 0; q < BP_ROWS; q++) w[q] = bitmap[r[q] >> 38];                 // d >> 6, d = r >> 32
#pragma unroll
  for (int q = 0; q < BP_ROWS; q++) {
    const int64_t i = base + (int64_t)q * BLOCK; const uint32_t d = (uint32_t)(r[q] >> 32), row = (uint32_t)r[q];
    if (i < total && ((w[q] >> (d & 63)) & 1ull)) atomicOr(&match_bits[row >> 6], 1ull << (row & 63));
  }
}
template <typename T>
static bool bp_probe_typed(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* pk, const uint64_t* mask, int64_t n, uint64_t* match_bits) {
  const T* keys = (const T*)pk->values->ptr; const uint64_t* valid = pk->validity ? (const uint64_t*)pk->validity->ptr : nullptr;
  // a column known to be sorted streams the bitmap: no sample, no read-back.  A table column is looked at once (the statistics stay with it); a computed column is sampled.
  auto st = order_stats_get(pk);
  if (!st && pk->base_column && !valid) st = order_stats_measure(ctx, pk);
  if (st && st->sorted) return false;
  HIP_CHECK(hipMemsetAsync(ctx->d_scratch64 + 10, 0, 32, ctx->stream));
  hipLaunchKernelGGL((k_bp_sample<T>), dim3(1), dim3(BLOCK), 0, ctx->stream, keys, valid, mask, n, t->key_min, t->range, (const uint64_t*)t->bitmap->ptr, (unsigned long long*)(ctx->d_scratch64 + 10));
  KERNEL_CHECK();
  const uint64_t* sm = read_scratch_range(ctx, 10, 3);
  const uint64_t near = sm[0], sel = sm[1], hit = sm[2];
  if (near * 2 >= 2048 || hit * 4 > sel) return false;                       // clustered keys stream the bitmap as it is; many matches would pay an atomic each
  int shift = 21; while ((t->range >> shift) >= RP_MAX_P) shift++;            // 2^21 keys = 256 KB of bitmap per partition
  const uint32_t P = (uint32_t)(((t->range - 1) >> shift) + 1);
  if (P < 8) return false;
  BufferPtr recs = alloc_buffer(ctx, (size_t)n * 8);
  RpCols cols{}; cols.n = 1; cols.c[0] = RpCol{ nullptr, recs->ptr, 8, RP_HASHKEY, 0 };
  rp_partition(ctx, RpHashKeyRange<T>{ keys, valid, mask, t->key_min, t->range, shift }, n, P, cols, false, ctx->d_scratch64 + 13, "bp_hist", "bp_scan", "bp_scatter", false);
  KernelTimer kt_(ctx, "bp_probe");
  HIP_CHECK(hipMemsetAsync(match_bits, 0, bitmap_bytes(n), ctx->stream));
  hipLaunchKernelGGL(k_bp_probe, dim3(grid_for(n, BLOCK * BP_ROWS)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)recs->ptr, (const unsigned long long*)(ctx->d_scratch64 + 13),
                     (const uint64_t*)t->bitmap->ptr, (unsigned long long*)match_bits);
  KERNEL_CHECK();
  return true;
}
// true = match_bits holds the answer of pass 1 (join.hip); false = not taken (clustered keys, a small batch or bitmap, many matches): the streaming probe runs
bool bp_probe(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* pk, const uint64_t* mask, int64_t n, uint64_t* match_bits) {
  if (!ctx->join_partitioned || !ctx->join_bitmap_partitioned || n < ctx->join_bitmap_partitioned_min_rows || n > 0xFFFFFFF0ll || !t->bitmap || t->range > (1ull << 32) || bitmap_bytes((int64_t)t->range) < ((size_t)8 << 20)) return false;          // a bitmap of a few MB is (mostly) L2-resident as it is
  switch (pk->type) {
    case DFGPU_INT64: return bp_probe_typed<int64_t>(ctx, t, pk, mask, n, match_bits);
    case DFGPU_INT32: case DFGPU_DATE32: return bp_probe_typed<int32_t>(ctx, t, pk, mask, n, match_bits);
    case DFGPU_UINT32: return bp_probe_typed<uint32_t>(ctx, t, pk, mask, n, match_bits);
    default: return false;
  }
}

}  // namespace dfgpu
