/* dfo_join.c -- CPU oracle restatement of HashJoinExec (TEST INFRASTRUCTURE ONLY).
 *
 * Follows datafusion/physical-plan/src/joins/hash_join.rs and joins/utils.rs:
 *   collect_left_input        hash_join.rs:678-768  (batches hashed & concatenated in REVERSE order)
 *   update_hash               hash_join.rs:777-815  (fifo: rows of a batch inserted in reverse)
 *   JoinHashMap               joins/utils.rs:121-229 (hash -> head+1, next[] chain, 0 = end)
 *   get_matched_indices_with_limit_offset + chain_traverse  joins/utils.rs:147-187, :284-348
 *   lookup_join_hashmap / equal_rows_arr / eq_dyn_null      hash_join.rs:1024-1118
 *   process_probe_batch (visited bitmap, alignment range)   hash_join.rs:1238-1343
 *   adjust_indices_by_join_type & friends                   joins/utils.rs:1234-1364
 *   process_unmatched_build_batch / get_final_indices_from_bit_map  hash_join.rs:1348-1388, utils.rs:1119-1141
 */
#include "dfo_internal.h"

typedef struct { int64_t *p; int64_t n, cap; } vec64;
typedef struct { int32_t *p; int64_t n, cap; } vec32;
static void v64_push(vec64 *v, int64_t x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 64; v->p = (int64_t *)dfo_xrealloc(v->p, (size_t)v->cap * 8); }
  v->p[v->n++] = x;
}
static void v32_push(vec32 *v, int32_t x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 64; v->p = (int32_t *)dfo_xrealloc(v->p, (size_t)v->cap * 4); }
  v->p[v->n++] = x;
}

/* RawTable<(u64 hash, u64 head)> stand-in: open addressing on the hash value. */
typedef struct { uint64_t *hash; uint64_t *head; uint64_t mask; } hmap;
static void hmap_init(hmap *m, int64_t n) {
  uint64_t cap = 16; while (cap < (uint64_t)n * 2 + 2) cap <<= 1;
  m->hash = (uint64_t *)calloc(cap, 8); m->head = (uint64_t *)calloc(cap, 8); m->mask = cap - 1;
}
static uint64_t *hmap_get(hmap *m, uint64_t h, int insert) {
  uint64_t s = dfo_mix64(h) & m->mask;
  for (;;) {
    if (m->head[s] == 0) { if (!insert) return NULL; m->hash[s] = h; return &m->head[s]; }
    if (m->hash[s] == h) return &m->head[s];
    s = (s + 1) & m->mask;
  }
}

typedef struct {
  const dfo_array *const *build_keys; int nb;
  const dfo_array *const *probe_keys; int np; int nkeys;
  int64_t n_build; int32_t *cb; int64_t *cr;   /* concat idx -> (input batch, row) */
  hmap map; uint64_t *next;
} jstate;

/* eq_dyn_null (hash_join.rs:1067-1076): eq -> NULL if either side NULL (dropped by the filter),
 * not_distinct when null_equals_null. */
static int keys_equal(const jstate *s, int64_t bidx, int pb, int64_t prow, int null_equals_null) {
  for (int c = 0; c < s->nkeys; c++) {
    int64_t i = s->cr[bidx], j = prow;
    const dfo_array *a = dfo_resolve(s->build_keys[(int64_t)s->cb[bidx] * s->nkeys + c], &i);
    const dfo_array *b = dfo_resolve(s->probe_keys[(int64_t)pb * s->nkeys + c], &j);
    if (!a || !b) { if (null_equals_null && !a && !b) continue; return 0; }
    if (!dfo_cell_equal(a, i, b, j)) return 0;
  }
  return 1;
}

int dfo_hash_join(const dfo_array *const *build_keys, int nb, const dfo_array *const *probe_keys,
                  int np, int nkeys, int join_type, int null_equals_null, int64_t batch_size,
                  int force_collisions, dfo_join_filter_fn filter, void *filter_ud,
                  dfo_join_result *out) {
  memset(out, 0, sizeof *out);
  if (nkeys < 1 || batch_size < 1) { dfo_set_error("hash_join: bad arguments"); return 1; }
  jstate s; memset(&s, 0, sizeof s);
  s.build_keys = build_keys; s.nb = nb; s.probe_keys = probe_keys; s.np = np; s.nkeys = nkeys;
  for (int b = 0; b < nb; b++) s.n_build += build_keys[(int64_t)b * nkeys]->length;
  s.cb = (int32_t *)dfo_xrealloc(NULL, (size_t)(s.n_build + 1) * 4);
  s.cr = (int64_t *)dfo_xrealloc(NULL, (size_t)(s.n_build + 1) * 8);
  s.next = (uint64_t *)calloc((size_t)s.n_build + 1, 8);
  hmap_init(&s.map, s.n_build);

  /* collect_left_input: iterate batches in reverse, offset accumulates (hash_join.rs:746-761);
   * concat_batches over the same reversed iterator (:764). */
  int64_t offset = 0;
  for (int b = nb - 1; b >= 0; b--) {
    int64_t n = build_keys[(int64_t)b * nkeys]->length;
    uint64_t *hashes = (uint64_t *)dfo_xrealloc(NULL, (size_t)(n + 1) * 8);
    dfo_create_hashes(build_keys + (int64_t)b * nkeys, nkeys, n, 0, force_collisions, hashes);
    for (int64_t r = 0; r < n; r++) { s.cb[offset + r] = b; s.cr[offset + r] = r; }
    /* update_from_iter over hash_values_iter.rev() (fifo_hashmap = true) */
    for (int64_t r = n - 1; r >= 0; r--) {
      int64_t row = r + offset;
      uint64_t *head = hmap_get(&s.map, hashes[r], 1);
      if (*head != 0) { s.next[row] = *head; *head = (uint64_t)row + 1; }
      else *head = (uint64_t)row + 1;
    }
    free(hashes);
    offset += n;
  }

  int need_final = join_type == DFO_JOIN_LEFT || join_type == DFO_JOIN_LEFT_ANTI ||
                   join_type == DFO_JOIN_LEFT_SEMI || join_type == DFO_JOIN_FULL;
  uint8_t *visited = (uint8_t *)calloc((size_t)s.n_build / 8 + 8, 1);

  vec64 ob = {0}, op = {0}, offs = {0}; vec32 opb = {0};
  v64_push(&offs, 0);

  for (int pb = 0; pb < np; pb++) {
    int64_t n = probe_keys[(int64_t)pb * nkeys]->length;
    if (n == 0) continue;
    uint64_t *hashes = (uint64_t *)dfo_xrealloc(NULL, (size_t)(n + 1) * 8);
    dfo_create_hashes(probe_keys + (int64_t)pb * nkeys, nkeys, n, 0, force_collisions, hashes);

    /* ProcessProbeBatchState { offset: (0, None), joined_probe_idx: None } */
    int64_t off_idx = 0; int off_has_next = 0; uint64_t off_next = 0;
    int have_joined = 0; int64_t joined_probe_idx = 0;
    for (;;) {
      /* ---- get_matched_indices_with_limit_offset (joins/utils.rs:284-348) ---- */
      vec64 cb_ = {0}, cp_ = {0};
      int64_t remaining = batch_size;
      int next_some = 0; int64_t next_idx = 0; uint64_t next_chain = 0; int limit_hit = 0;
      int64_t to_skip;
#define CHAIN_TRAVERSE(INPUT_IDX, CHAIN_IDX)                                              \
      {                                                                                   \
        uint64_t i_ = (CHAIN_IDX) - 1;                                                    \
        for (;;) {                                                                        \
          v64_push(&cb_, (int64_t)i_); v64_push(&cp_, (INPUT_IDX));                       \
          remaining--;                                                                    \
          uint64_t nx_ = s.next[i_];                                                      \
          if (remaining == 0) {                                                           \
            if ((INPUT_IDX) == n - 1 && nx_ == 0) next_some = 0;                          \
            else { next_some = 1; next_idx = (INPUT_IDX); next_chain = nx_; }             \
            limit_hit = 1; break;                                                         \
          }                                                                               \
          if (nx_ == 0) break;                                                            \
          i_ = nx_ - 1;                                                                   \
        }                                                                                 \
      }
      if (!off_has_next) to_skip = off_idx;
      else if (off_next == 0) to_skip = off_idx + 1;
      else { CHAIN_TRAVERSE(off_idx, off_next); to_skip = off_idx + 1; }
      if (!limit_hit) {
        for (int64_t row = to_skip; row < n; row++) {
          uint64_t *head = hmap_get(&s.map, hashes[row], 0);
          if (head) { CHAIN_TRAVERSE(row, *head); if (limit_hit) break; }
        }
      }
#undef CHAIN_TRAVERSE
      /* ---- equal_rows_arr (hash_join.rs:1078-1118) ---- */
      vec64 mb = {0}, mp = {0};
      for (int64_t i = 0; i < cb_.n; i++)
        if (keys_equal(&s, cb_.p[i], pb, cp_.p[i], null_equals_null)) { v64_push(&mb, cb_.p[i]); v64_push(&mp, cp_.p[i]); }
      free(cb_.p); free(cp_.p);
      /* ---- apply_join_filter_to_indices (joins/utils.rs:1143-1176) ---- */
      if (filter && mb.n > 0) {
        uint8_t *keep = (uint8_t *)calloc((size_t)mb.n, 1);
        filter(filter_ud, pb, mb.p, mp.p, mb.n, keep);
        int64_t w = 0;
        for (int64_t i = 0; i < mb.n; i++) if (keep[i]) { mb.p[w] = mb.p[i]; mp.p[w] = mp.p[i]; w++; }
        mb.n = mp.n = w; free(keep);
      }
      /* mark visited (hash_join.rs:1274-1278) */
      if (need_final) for (int64_t i = 0; i < mb.n; i++) dfo_bit_set(visited, mb.p[i], 1);
      /* alignment range (hash_join.rs:1297-1309) */
      int last_some = mp.n > 0; int64_t last_joined = last_some ? mp.p[mp.n - 1] : 0;
      int64_t range_start = have_joined ? joined_probe_idx + 1 : 0;
      int64_t range_end = !next_some ? n : (last_some ? last_joined + 1 : 0);
      /* ---- adjust_indices_by_join_type (joins/utils.rs:1234-1279) ---- */
      switch (join_type) {
        case DFO_JOIN_INNER: case DFO_JOIN_LEFT:
          for (int64_t i = 0; i < mb.n; i++) { v64_push(&ob, mb.p[i]); v64_push(&op, mp.p[i]); v32_push(&opb, pb); }
          break;
        case DFO_JOIN_RIGHT: case DFO_JOIN_FULL: case DFO_JOIN_RIGHT_SEMI: case DFO_JOIN_RIGHT_ANTI: {
          /* get_anti_indices / get_semi_indices bitmap over the range (:1309-1364) */
          int64_t rl = range_end > range_start ? range_end - range_start : 0;
          uint8_t *bm = (uint8_t *)calloc((size_t)rl / 8 + 8, 1);
          for (int64_t i = 0; i < mp.n; i++)
            if (mp.p[i] >= range_start && mp.p[i] < range_end) dfo_bit_set(bm, mp.p[i] - range_start, 1);
          if (join_type == DFO_JOIN_RIGHT || join_type == DFO_JOIN_FULL) {
            for (int64_t i = 0; i < mb.n; i++) { v64_push(&ob, mb.p[i]); v64_push(&op, mp.p[i]); v32_push(&opb, pb); }
            for (int64_t i = 0; i < rl; i++) if (!dfo_bit(bm, i)) { v64_push(&ob, -1); v64_push(&op, range_start + i); v32_push(&opb, pb); } /* append_right_indices :1284-1306 */
          } else {
            int want = join_type == DFO_JOIN_RIGHT_SEMI;
            for (int64_t i = 0; i < rl; i++) if (dfo_bit(bm, i) == want) { v64_push(&ob, -1); v64_push(&op, range_start + i); v32_push(&opb, pb); }
          }
          free(bm);
          break;
        }
        default: break; /* LeftSemi / LeftAnti emit nothing per probe batch (:1270-1277) */
      }
      v64_push(&offs, ob.n);
      free(mb.p); free(mp.p);
      if (!next_some) break;
      /* state.advance(next_offset, last_joined_right_idx) (hash_join.rs:1335-1339, :920-925) */
      off_idx = next_idx; off_has_next = 1; off_next = next_chain;
      if (last_some) { have_joined = 1; joined_probe_idx = last_joined; }
    }
    free(hashes);
  }

  /* process_unmatched_build_batch (hash_join.rs:1348-1388) */
  if (need_final) {
    int want = join_type == DFO_JOIN_LEFT_SEMI;
    for (int64_t i = 0; i < s.n_build; i++)
      if (dfo_bit(visited, i) == want) { v64_push(&ob, i); v64_push(&op, -1); v32_push(&opb, -1); }
    v64_push(&offs, ob.n);
  }

  out->n = ob.n; out->build_idx = ob.p; out->probe_idx = op.p; out->probe_batch = opb.p;
  out->n_batches = offs.n - 1; out->batch_offsets = offs.p;
  free(visited); free(s.cb); free(s.cr); free(s.next); free(s.map.hash); free(s.map.head);
  return 0;
}

void dfo_join_result_free(dfo_join_result *r) {
  free(r->build_idx); free(r->probe_idx); free(r->probe_batch); free(r->batch_offsets);
  memset(r, 0, sizeof *r);
}
