// join_table.h -- the build side of a HashJoinExec as the device sees it (shared by join.hip and pjoin.hip).
#pragma once
#include "device_utils.h"

namespace dfgpu {
// radix-partitioned build side (pjoin.hip): (key widened to 64 bits, original build row) grouped by partition of the key hash, sized so
// that one partition's open-addressing table fits 64 KB of LDS (two workgroups per CU; 128 KB at most)
struct PartitionedBuild {
  uint32_t P = 0; int sbits = 0;            // partitions; log2 of the LDS table slots
  int64_t rows = 0;                         // selected, non-NULL build rows
  BufferPtr recs;                           // RpRec12 {key lo, key hi, ref}[..], partition-major: ref = build row (unique keys) or group number (dups)
  BufferPtr starts;                         // u32[P + 1] over recs
  bool hashed = false;                      // record keys are keyset hashes of the key columns (any number, any type, NULLs under null_equals_null): a match is a candidate, verified against the columns
  bool dups = false;                        // a build key repeats: recs holds one record per DISTINCT key, the rows of a key are a CSR
  BufferPtr grp_start, grp_cnt, csr_rows;   // u32[groups] into csr_rows / rows of the group; u32[rows] build rows, ascending inside a group
};
}  // namespace dfgpu

struct dfgpu_join_table {
  dfgpu_ctx* ctx = nullptr;
  int64_t n_build = 0; int32_t nkeys = 0; bool null_equals_null = false;
  std::vector<dfgpu_array*> keys; dfgpu::KeySet ks{};
  uint64_t capacity = 0; int cap_bits = 0;
  dfgpu::BufferPtr slots;        // u64[capacity]
  dfgpu::BufferPtr slot_count;   // u32[capacity]   rows per key group
  dfgpu::BufferPtr slot_start;   // u32[capacity]   CSR start (non-unique only)
  dfgpu::BufferPtr csr_rows;     // u32[n_inserted] build rows ordered by (slot, row) (non-unique only)
  dfgpu::BufferPtr build_mask;   // effective opt_mask words or null
  dfgpu::BufferPtr visited;      // u64 words over n_build
  bool unique = true;
  // exact membership bitmap over [key_min, key_min + range) for single integer keys with a dense domain: the probe tests
  // one bit (L2 / Infinity Cache resident, perfectly local for clustered keys) and touches the hash table for matches only
  dfgpu::BufferPtr bitmap; int64_t key_min = 0; uint64_t range = 0;
  // a build side that is tiny against its key range (a few thousand order keys out of 600 M) gets no bitmap up front; a probe batch
  // of >= range / 16 rows builds it on arrival (clearing range / 8 bytes is then small against streaming the probe keys)
  bool lazy_bitmap = false; dfgpu::BufferPtr lazy_row_slot;
  // rank index (strictly increasing single integer key, the shape of every clustered primary key): no hash table at all.
  // The bitmap IS the table: build row = rank of the key's bit among the set bits (word prefix + popcount), mapped through
  // sel_rows when a build selection is fused; rank_identity = the keys are key_min + row, so the row is the key offset.
  bool rank_mode = false, rank_identity = false;
  // rank_runs: the keys are non-decreasing WITH repeats (a sorted foreign key): sel_rows[r] = first build row of the r-th distinct key,
  // its rows are the run up to sel_rows[r + 1] (or n_build) -- the CSR of the hash path without hashing or sorting
  bool rank_runs = false;
  // key packing: 2..4 integer key columns whose value ranges multiply to < 2^40 are packed into ONE Int64 key (sum of (k - min) * stride):
  // tuple equality == packed equality, and the single-key paths (rank index, bitmap prefilter) apply.  The table then holds the packed
  // column as its only key; probes pack their tuples with the same parameters (a component outside the build range = NULL = no match).
  int pack_n = 0; int32_t pack_types[dfgpu::MAX_KEYS] = {0}; int64_t pack_min[dfgpu::MAX_KEYS] = {0}; uint64_t pack_range[dfgpu::MAX_KEYS] = {0}, pack_stride[dfgpu::MAX_KEYS] = {0};
  dfgpu::BufferPtr rank_prefix;  // u32[range / 64]   set bits before each bitmap word
  bool have_minmax = false; long long sel_min = 0, sel_max = 0;      // min / max of the selected build keys, once some builder has computed them
  dfgpu_array* sel_rows = nullptr;   // u32[selected] ascending build rows (masked builds only)
  std::unique_ptr<dfgpu::PartitionedBuild> part;   // set = probes of large batches run partition by partition out of LDS (pjoin.hip)
  int64_t mem = 0;
  ~dfgpu_join_table() { for (auto* a : keys) dfgpu_array_release(a); if (sel_rows) dfgpu_array_release(sel_rows); }
};

namespace dfgpu {
// pjoin.hip
bool pj_build(dfgpu_ctx* ctx, dfgpu_join_table* t);      // false = shape not taken (nothing kept)
bool pj_probe_eligible(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* const* probe_keys, int32_t nkeys, int64_t n);
bool bp_probe(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* probe_key, const uint64_t* mask, int64_t n, uint64_t* match_bits);      // membership bitmap probed by key range (unclustered probe keys)
void pj_probe(dfgpu_ctx* ctx, const dfgpu_join_table* t, const dfgpu_array* const* probe_keys, int32_t nkeys, const uint64_t* mask, dfgpu_array** out_build, dfgpu_array** out_probe);
bool pj_hashed_candidate(dfgpu_ctx* ctx, const dfgpu_join_table* t);      // a build the integer mode does not take, large enough for the partitioned path
}  // namespace dfgpu
