/* dfo_sort.c -- CPU oracle: lexsort_to_indices and hash partitioning (TEST INFRASTRUCTURE ONLY).
 *
 * sort_batch (physical-plan/src/sorts/sort.rs:584-609) = lexsort_to_indices(sort columns, fetch) + take.
 * arrow-ord 50 semantics restated: per column SortOptions{descending, nulls_first}; NULLs are placed by
 * nulls_first independently of `descending`; floats use IEEE totalOrder (NaN above +inf, -0 < +0)
 * (pinned by sorts/sort.rs:1290-1392 test_lex_sort_by_float).  The reference's tie order is unspecified
 * (sort_unstable_by); the oracle is STABLE (ties keep input order), as is the HIP radix sort.
 *
 * BatchPartitioner::partition_iter (repartition/mod.rs:148-221): dest = hash % n with create_hashes over
 * the key columns and fixed seeds; per destination the row indices are kept in input order (:196-214).
 */
#include "dfo_internal.h"

typedef struct { const dfo_array *const *cols; int k; const uint8_t *desc; const uint8_t *nf; } sortctx;

static int row_cmp(const sortctx *c, uint32_t a, uint32_t b) {
  for (int j = 0; j < c->k; j++) {
    int64_t i = a, q = b;
    const dfo_array *x = dfo_resolve(c->cols[j], &i), *y = dfo_resolve(c->cols[j], &q);
    if (!x || !y) {
      if (!x && !y) continue;
      int r = !x ? -1 : 1;               /* null first */
      return c->nf[j] ? r : -r;
    }
    int r = dfo_cell_cmp(x, i, y, q);
    if (r) return c->desc[j] ? -r : r;
  }
  return 0;
}

static void msort(const sortctx *c, uint32_t *a, uint32_t *tmp, int64_t n) {
  if (n < 2) return;
  int64_t h = n / 2;
  msort(c, a, tmp, h); msort(c, a + h, tmp, n - h);
  int64_t i = 0, j = h, o = 0;
  while (i < h && j < n) tmp[o++] = row_cmp(c, a[j], a[i]) < 0 ? a[j++] : a[i++];
  while (i < h) tmp[o++] = a[i++];
  while (j < n) tmp[o++] = a[j++];
  memcpy(a, tmp, (size_t)n * 4);
}

int dfo_lexsort_to_indices(const dfo_array *const *cols, int k, const uint8_t *descending, const uint8_t *nulls_first,
                           int64_t n, int64_t fetch, uint32_t *out, int64_t *n_out) {
  if (k < 1) { dfo_set_error("Sort requires at least one column"); return 1; }
  sortctx c = { cols, k, descending, nulls_first };
  uint32_t *idx = (uint32_t *)dfo_xrealloc(NULL, (size_t)(n + 1) * 4), *tmp = (uint32_t *)dfo_xrealloc(NULL, (size_t)(n + 1) * 4);
  for (int64_t i = 0; i < n; i++) idx[i] = (uint32_t)i;
  msort(&c, idx, tmp, n);
  int64_t m = fetch >= 0 && fetch < n ? fetch : n;
  memcpy(out, idx, (size_t)m * 4); *n_out = m;
  free(idx); free(tmp);
  return 0;
}

int dfo_hash_partition(const dfo_array *const *cols, int k, int64_t n, int num_partitions, int force_collisions,
                       uint32_t *indices_out, int64_t *counts_out) {
  if (num_partitions < 1) { dfo_set_error("partition: n < 1"); return 1; }
  uint64_t *h = (uint64_t *)dfo_xrealloc(NULL, (size_t)(n + 1) * 8);
  dfo_create_hashes(cols, k, n, 0, force_collisions, h);
  for (int p = 0; p < num_partitions; p++) counts_out[p] = 0;
  for (int64_t i = 0; i < n; i++) counts_out[h[i] % (uint64_t)num_partitions]++;
  int64_t *start = (int64_t *)dfo_xrealloc(NULL, (size_t)num_partitions * 8), acc = 0;
  for (int p = 0; p < num_partitions; p++) { start[p] = acc; acc += counts_out[p]; }
  for (int64_t i = 0; i < n; i++) indices_out[start[h[i] % (uint64_t)num_partitions]++] = (uint32_t)i;
  free(start); free(h);
  return 0;
}
