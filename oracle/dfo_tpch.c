/* dfo_tpch.c -- CPU oracle: TPC-H Q3 through the restated DataFusion 36 CPU operators
 * (TEST INFRASTRUCTURE ONLY; also the "port" cpu_baseline timed by bench.py).
 *
 * Plan shape = the reference's own physical plan, datafusion/sqllogictest/test_files/tpch/q3.slt.part
 * (physical_plan, benchmark SQL benchmarks/queries/q3.sql has no LIMIT):
 *   FilterExec -> RepartitionExec Hash(key, P) -> HashJoinExec mode=Partitioned Inner (x2)
 *   -> ProjectionExec -> AggregateExec Partial -> RepartitionExec Hash(3 keys, P) -> AggregateExec
 *   FinalPartitioned -> SortExec [revenue DESC, o_orderdate ASC NULLS LAST] -> SortPreservingMergeExec.
 * One worker per partition (target_partitions, common/src/config.rs:230), batch_size rows per batch
 * (config.rs:215).  Per-batch flow follows the operators restated in dfo_join.c / dfo_agg.c:
 * create_hashes -> JoinHashMap chain lookup -> key equality -> gather; GroupValuesRows interning
 * (first-seen ids) + SUM(Decimal128) add_wrapping; Partial state merged by FinalPartitioned.
 * revenue = l_extendedprice * (1 - l_discount): Decimal128(15,2) * (Decimal128(20,0) - Decimal128(15,2))
 *         = Decimal128(15,2) * Decimal128(23,2) -> Decimal128(38,4)   (type_coercion/binary.rs:524-538,
 *         arrow-arith decimal rules restated in dfo_expr.c); SUM -> Decimal128(38,4).
 */
#include "dfo_internal.h"
#include <omp.h>

typedef struct { int64_t n, cap; int64_t *a, *b; int32_t *c, *d; i128 *e, *f; } buf;   /* generic column bag */
static void buf_reserve(buf *x, int64_t extra, int mask) {
  if (x->n + extra <= x->cap) return;
  int64_t nc = x->cap ? x->cap * 2 : 4096; while (nc < x->n + extra) nc *= 2;
  if (mask & 1) x->a = (int64_t *)dfo_xrealloc(x->a, (size_t)nc * 8);
  if (mask & 2) x->b = (int64_t *)dfo_xrealloc(x->b, (size_t)nc * 8);
  if (mask & 4) x->c = (int32_t *)dfo_xrealloc(x->c, (size_t)nc * 4);
  if (mask & 8) x->d = (int32_t *)dfo_xrealloc(x->d, (size_t)nc * 4);
  if (mask & 16) x->e = (i128 *)dfo_xrealloc(x->e, (size_t)nc * 16);
  if (mask & 32) x->f = (i128 *)dfo_xrealloc(x->f, (size_t)nc * 16);
  x->cap = nc;
}
static void buf_free(buf *x) { free(x->a); free(x->b); free(x->c); free(x->d); free(x->e); free(x->f); memset(x, 0, sizeof *x); }

/* JoinHashMap (joins/utils.rs:121-229) over the u64 hash; head/next store row+1, 0 = end */
typedef struct { uint64_t *hash, *head, *next; uint64_t mask; } jmap;
static void jmap_build(jmap *m, const int64_t *keys, int64_t n) {
  uint64_t cap = 16; while (cap < (uint64_t)n * 2 + 2) cap <<= 1;
  m->hash = (uint64_t *)calloc(cap, 8); m->head = (uint64_t *)calloc(cap, 8); m->next = (uint64_t *)calloc((size_t)n + 1, 8); m->mask = cap - 1;
  for (int64_t r = n - 1; r >= 0; r--) {                 /* update_hash fifo order (hash_join.rs:808-809) */
    uint64_t h = dfo_mix64((uint64_t)keys[r]), s = dfo_mix64(h) & m->mask;
    for (;;) { if (m->head[s] == 0) { m->hash[s] = h; m->head[s] = (uint64_t)r + 1; break; }
      if (m->hash[s] == h) { m->next[r] = m->head[s]; m->head[s] = (uint64_t)r + 1; break; }
      s = (s + 1) & m->mask; }
  }
}
static inline uint64_t jmap_get(const jmap *m, uint64_t h) {
  uint64_t s = dfo_mix64(h) & m->mask;
  for (;;) { if (m->head[s] == 0) return 0; if (m->hash[s] == h) return m->head[s]; s = (s + 1) & m->mask; }
}
static void jmap_free(jmap *m) { free(m->hash); free(m->head); free(m->next); memset(m, 0, sizeof *m); }

/* GroupValuesRows stand-in for the 3-column key (l_orderkey i64, o_orderdate i32, o_shippriority i32) */
typedef struct { int64_t n, cap; int64_t *k; int32_t *d, *p; i128 *sum; uint64_t *ghash; int64_t *slot; uint64_t mask; } gtab;
static void gtab_init(gtab *g) { memset(g, 0, sizeof *g); g->mask = 4095; g->slot = (int64_t *)dfo_xrealloc(NULL, 4096 * 8); memset(g->slot, 0xff, 4096 * 8); }
static inline uint64_t gkey_hash(int64_t k, int32_t d, int32_t p) {    /* create_hashes over 3 columns */
  uint64_t h = dfo_mix64((uint64_t)k);
  h = dfo_combine_hashes(dfo_mix64((uint64_t)(int64_t)d), h);
  return dfo_combine_hashes(dfo_mix64((uint64_t)(int64_t)p), h);
}
static int64_t gtab_intern(gtab *g, int64_t k, int32_t d, int32_t p) {
  if ((uint64_t)(g->n + 1) * 2 > g->mask + 1) {
    uint64_t nc = (g->mask + 1) * 2; free(g->slot); g->slot = (int64_t *)dfo_xrealloc(NULL, nc * 8); memset(g->slot, 0xff, nc * 8); g->mask = nc - 1;
    for (int64_t i = 0; i < g->n; i++) { uint64_t s = dfo_mix64(g->ghash[i]) & g->mask; while (g->slot[s] >= 0) s = (s + 1) & g->mask; g->slot[s] = i; }
  }
  uint64_t h = gkey_hash(k, d, p), s = dfo_mix64(h) & g->mask;
  for (;;) {
    int64_t id = g->slot[s];
    if (id < 0) break;
    if (g->ghash[id] == h && g->k[id] == k && g->d[id] == d && g->p[id] == p) return id;
    s = (s + 1) & g->mask;
  }
  if (g->n == g->cap) { g->cap = g->cap ? g->cap * 2 : 4096;
    g->k = (int64_t *)dfo_xrealloc(g->k, (size_t)g->cap * 8); g->d = (int32_t *)dfo_xrealloc(g->d, (size_t)g->cap * 4); g->p = (int32_t *)dfo_xrealloc(g->p, (size_t)g->cap * 4);
    g->sum = (i128 *)dfo_xrealloc(g->sum, (size_t)g->cap * 16); g->ghash = (uint64_t *)dfo_xrealloc(g->ghash, (size_t)g->cap * 8); }
  int64_t id = g->n++; g->k[id] = k; g->d[id] = d; g->p[id] = p; g->sum[id] = 0; g->ghash[id] = h; g->slot[s] = id;
  return id;
}
static void gtab_free(gtab *g) { free(g->k); free(g->d); free(g->p); free(g->sum); free(g->ghash); free(g->slot); memset(g, 0, sizeof *g); }

static inline i128 ld128(const __int128 *p, int64_t i) { i128 v; memcpy(&v, (const char *)p + 16 * i, 16); return v; }

typedef struct { int64_t k; i128 rev; int32_t d, p; } orow;
static int orow_cmp(const orow *x, const orow *y) {   /* revenue DESC, o_orderdate ASC */
  if (x->rev != y->rev) return x->rev > y->rev ? -1 : 1;
  if (x->d != y->d) return x->d < y->d ? -1 : 1;
  return 0;
}
static void orow_msort(orow *a, orow *tmp, int64_t n) {
  if (n < 2) return; int64_t h = n / 2; orow_msort(a, tmp, h); orow_msort(a + h, tmp, n - h);
  int64_t i = 0, j = h, o = 0;
  while (i < h && j < n) tmp[o++] = orow_cmp(&a[j], &a[i]) < 0 ? a[j++] : a[i++];
  while (i < h) tmp[o++] = a[i++]; while (j < n) tmp[o++] = a[j++];
  memcpy(a, tmp, (size_t)n * sizeof(orow));
}

int dfo_tpch_q3(const dfo_q3_input *in, int P, int64_t B, dfo_q3_output *out) {
  if (P < 1 || B < 1) { dfo_set_error("q3: bad arguments"); return 1; }
  memset(out, 0, sizeof *out);
  omp_set_num_threads(P);
  /* bufs[src*P + dst] : exchange buffers of one RepartitionExec */
  buf *xc = (buf *)calloc((size_t)P * P, sizeof(buf)), *xo = (buf *)calloc((size_t)P * P, sizeof(buf));
  buf *xj = (buf *)calloc((size_t)P * P, sizeof(buf)), *xl = (buf *)calloc((size_t)P * P, sizeof(buf));
  buf *xa = (buf *)calloc((size_t)P * P, sizeof(buf));
  gtab *fin = (gtab *)calloc((size_t)P, sizeof(gtab));

  /* ---- customer: FilterExec c_mktsegment = 'BUILDING' -> Projection(c_custkey) -> Repartition Hash(c_custkey) */
#pragma omp parallel for schedule(static, 1)
  for (int t = 0; t < P; t++) {
    int64_t lo = in->n_customer * t / P, hi = in->n_customer * (t + 1) / P;
    for (int64_t b0 = lo; b0 < hi; b0 += B) {
      int64_t b1 = b0 + B < hi ? b0 + B : hi;
      for (int64_t i = b0; i < b1; i++) if (in->c_mktsegment[i] == in->segment_code) {
        int64_t k = in->c_custkey[i]; buf *d = &xc[(int64_t)t * P + (int)(dfo_mix64((uint64_t)k) % (uint64_t)P)];
        buf_reserve(d, 1, 1); d->a[d->n++] = k;
      }
    }
  }
  /* ---- orders: FilterExec o_orderdate < cut -> Repartition Hash(o_custkey) */
#pragma omp parallel for schedule(static, 1)
  for (int t = 0; t < P; t++) {
    int64_t lo = in->n_orders * t / P, hi = in->n_orders * (t + 1) / P;
    for (int64_t i = lo; i < hi; i++) if (in->o_orderdate[i] < in->date_cut) {
      int64_t ck = in->o_custkey[i]; buf *d = &xo[(int64_t)t * P + (int)(dfo_mix64((uint64_t)ck) % (uint64_t)P)];
      buf_reserve(d, 1, 1 | 2 | 4 | 8); d->a[d->n] = in->o_orderkey[i]; d->b[d->n] = ck; d->c[d->n] = in->o_orderdate[i]; d->d[d->n] = in->o_shippriority[i]; d->n++;
    }
  }
  /* ---- HashJoinExec Partitioned Inner on (c_custkey = o_custkey) -> Projection -> Repartition Hash(o_orderkey) */
#pragma omp parallel for schedule(static, 1)
  for (int p = 0; p < P; p++) {
    buf build = {0};
    for (int s = 0; s < P; s++) { buf *x = &xc[(int64_t)s * P + p]; buf_reserve(&build, x->n, 1); memcpy(build.a + build.n, x->a, (size_t)x->n * 8); build.n += x->n; }
    jmap m; jmap_build(&m, build.a, build.n);
    uint64_t *hb = (uint64_t *)dfo_xrealloc(NULL, (size_t)B * 8);
    for (int s = 0; s < P; s++) {
      buf *x = &xo[(int64_t)s * P + p];
      for (int64_t b0 = 0; b0 < x->n; b0 += B) {                       /* one probe RecordBatch */
        int64_t nb = x->n - b0 < B ? x->n - b0 : B;
        for (int64_t i = 0; i < nb; i++) hb[i] = dfo_mix64((uint64_t)x->b[b0 + i]);     /* create_hashes */
        for (int64_t i = 0; i < nb; i++) {
          uint64_t c = jmap_get(&m, hb[i]);
          while (c) { int64_t br = (int64_t)c - 1;
            if (build.a[br] == x->b[b0 + i]) {                        /* equal_rows_arr */
              int64_t ok = x->a[b0 + i]; buf *d = &xj[(int64_t)p * P + (int)(dfo_mix64((uint64_t)ok) % (uint64_t)P)];
              buf_reserve(d, 1, 1 | 4 | 8); d->a[d->n] = ok; d->c[d->n] = x->c[b0 + i]; d->d[d->n] = x->d[b0 + i]; d->n++;
            }
            c = m.next[br]; }
        }
      }
    }
    free(hb); jmap_free(&m); buf_free(&build);
  }
  /* ---- lineitem: FilterExec l_shipdate > cut -> Projection -> Repartition Hash(l_orderkey) */
#pragma omp parallel for schedule(static, 1)
  for (int t = 0; t < P; t++) {
    int64_t lo = in->n_lineitem * t / P, hi = in->n_lineitem * (t + 1) / P;
    for (int64_t i = lo; i < hi; i++) if (in->l_shipdate[i] > in->date_cut) {
      int64_t k = in->l_orderkey[i]; buf *d = &xl[(int64_t)t * P + (int)(dfo_mix64((uint64_t)k) % (uint64_t)P)];
      buf_reserve(d, 1, 1 | 16 | 32); d->a[d->n] = k; d->e[d->n] = ld128(in->l_extendedprice, i); d->f[d->n] = ld128(in->l_discount, i); d->n++;
    }
  }
  /* ---- HashJoinExec (o_orderkey = l_orderkey) -> Projection -> AggregateExec Partial -> Repartition Hash(3 keys) */
#pragma omp parallel for schedule(static, 1)
  for (int p = 0; p < P; p++) {
    buf build = {0};
    for (int s = 0; s < P; s++) { buf *x = &xj[(int64_t)s * P + p]; buf_reserve(&build, x->n, 1 | 4 | 8);
      memcpy(build.a + build.n, x->a, (size_t)x->n * 8); memcpy(build.c + build.n, x->c, (size_t)x->n * 4); memcpy(build.d + build.n, x->d, (size_t)x->n * 4); build.n += x->n; }
    jmap m; jmap_build(&m, build.a, build.n);
    gtab g; gtab_init(&g);
    uint64_t *hb = (uint64_t *)dfo_xrealloc(NULL, (size_t)B * 8);
    for (int s = 0; s < P; s++) {
      buf *x = &xl[(int64_t)s * P + p];
      for (int64_t b0 = 0; b0 < x->n; b0 += B) {
        int64_t nb = x->n - b0 < B ? x->n - b0 : B;
        for (int64_t i = 0; i < nb; i++) hb[i] = dfo_mix64((uint64_t)x->a[b0 + i]);
        for (int64_t i = 0; i < nb; i++) {
          uint64_t c = jmap_get(&m, hb[i]);
          while (c) { int64_t br = (int64_t)c - 1;
            if (build.a[br] == x->a[b0 + i]) {
              /* (Decimal128(20,0) 1 -> scale 2 = 100) - l_discount, checked; then * l_extendedprice, checked */
              i128 one_minus = 100 - x->f[b0 + i], rev;
              if (__builtin_mul_overflow(x->e[b0 + i], one_minus, &rev)) { rev = 0; }
              int64_t id = gtab_intern(&g, x->a[b0 + i], build.c[br], build.d[br]);
              g.sum[id] = (i128)((u128)g.sum[id] + (u128)rev);       /* add_wrapping (sum.rs:137) */
            }
            c = m.next[br]; }
        }
      }
    }
    for (int64_t i = 0; i < g.n; i++) {                                /* emit Partial state -> hash repartition */
      buf *d = &xa[(int64_t)p * P + (int)(g.ghash[i] % (uint64_t)P)];
      buf_reserve(d, 1, 1 | 4 | 8 | 16); d->a[d->n] = g.k[i]; d->c[d->n] = g.d[i]; d->d[d->n] = g.p[i]; d->e[d->n] = g.sum[i]; d->n++;
    }
    free(hb); gtab_free(&g); jmap_free(&m); buf_free(&build);
  }
  /* ---- AggregateExec FinalPartitioned (merge_batch) -> SortExec per partition */
  orow **parts = (orow **)calloc((size_t)P, sizeof(orow *)); int64_t *pn = (int64_t *)calloc((size_t)P, 8);
#pragma omp parallel for schedule(static, 1)
  for (int p = 0; p < P; p++) {
    gtab *g = &fin[p]; gtab_init(g);
    for (int s = 0; s < P; s++) { buf *x = &xa[(int64_t)s * P + p];
      for (int64_t i = 0; i < x->n; i++) { int64_t id = gtab_intern(g, x->a[i], x->c[i], x->d[i]); g->sum[id] = (i128)((u128)g->sum[id] + (u128)x->e[i]); } }
    orow *r = (orow *)dfo_xrealloc(NULL, (size_t)(g->n + 1) * sizeof(orow)), *tmp = (orow *)dfo_xrealloc(NULL, (size_t)(g->n + 1) * sizeof(orow));
    for (int64_t i = 0; i < g->n; i++) { r[i].k = g->k[i]; r[i].rev = g->sum[i]; r[i].d = g->d[i]; r[i].p = g->p[i]; }
    orow_msort(r, tmp, g->n); free(tmp);
    parts[p] = r; pn[p] = g->n; gtab_free(g);
  }
  /* ---- SortPreservingMergeExec: k-way merge, lowest partition wins ties */
  int64_t total = 0; for (int p = 0; p < P; p++) total += pn[p];
  out->n = total;
  out->l_orderkey = (int64_t *)dfo_xrealloc(NULL, (size_t)(total + 1) * 8); out->revenue = (__int128 *)dfo_xrealloc(NULL, (size_t)(total + 1) * 16);
  out->o_orderdate = (int32_t *)dfo_xrealloc(NULL, (size_t)(total + 1) * 4); out->o_shippriority = (int32_t *)dfo_xrealloc(NULL, (size_t)(total + 1) * 4);
  int64_t *cur = (int64_t *)calloc((size_t)P, 8);
  for (int64_t o = 0; o < total; o++) {
    int best = -1;
    for (int p = 0; p < P; p++) if (cur[p] < pn[p] && (best < 0 || orow_cmp(&parts[p][cur[p]], &parts[best][cur[best]]) < 0)) best = p;
    orow *r = &parts[best][cur[best]++];
    out->l_orderkey[o] = r->k; memcpy((char *)out->revenue + 16 * o, &r->rev, 16); out->o_orderdate[o] = r->d; out->o_shippriority[o] = r->p;
  }
  free(cur);
  for (int p = 0; p < P; p++) free(parts[p]);
  free(parts); free(pn); free(fin);
  for (int64_t i = 0; i < (int64_t)P * P; i++) { buf_free(&xc[i]); buf_free(&xo[i]); buf_free(&xj[i]); buf_free(&xl[i]); buf_free(&xa[i]); }
  free(xc); free(xo); free(xj); free(xl); free(xa);
  return 0;
}

void dfo_q3_output_free(dfo_q3_output *o) {
  free(o->l_orderkey); free(o->revenue); free(o->o_orderdate); free(o->o_shippriority); memset(o, 0, sizeof *o);
}
