/* TPC-H Q3 through the drop-in boundary from plain C -- no Python, no PyTorch: host columns -> dfgpu_array_import_host -> the
 * ExecutionPlan mirror of include/dfgpu_exec.h (the reference's physical plan, sqllogictest/test_files/tpch/q3.slt.part, one
 * partition) -> dfgpu_plan_execute / dfgpu_stream_next -> dfgpu_array_export_host, checked row for row against a nested-loop
 * evaluation of the same query on the host.  This is the call sequence a Rust / Go / JVM binding makes.
 * Build: gcc -std=gnu11 -O2 -Iinclude examples/q3_native.c -Ldatafusion-upstream_amd -ldfgpu -Wl,-rpath,$PWD/datafusion-upstream_amd -o q3_native */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dfgpu.h"
#include "dfgpu_exec.h"

typedef __int128 i128;
#define CK(ctx, call) do { dfgpu_status st_ = (call); if (st_ != DFGPU_OK) { fprintf(stderr, "%s -> %d: %s / %s\n", #call, (int)st_, (ctx) ? dfgpu_last_error(ctx) : "", dfgpu_exec_last_error()); exit(1); } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static dfgpu_array *import_fixed(dfgpu_ctx *ctx, int32_t type, int32_t p, int32_t s, const void *values, int64_t n) {
  dfgpu_array_desc d; memset(&d, 0, sizeof d); d.type = type; d.precision = p; d.scale = s; d.length = n; d.values = values;
  dfgpu_array *a = NULL; CK(ctx, dfgpu_array_import_host(ctx, &d, &a)); return a;
}
/* Handles are reference counted inside the library (a parent keeps its children alive): every handle made here is recorded and
 * dropped at the end, as a binding's Drop impls would. */
static dfgpu_expr *exprs[128]; static int n_exprs; static dfgpu_plan *plans[64]; static int n_plans;
static dfgpu_expr *keep_e(dfgpu_expr *e) { exprs[n_exprs++] = e; return e; }
static dfgpu_plan *keep_p(dfgpu_plan *p) { plans[n_plans++] = p; return p; }
static dfgpu_expr *col(const char *name, int idx) { dfgpu_expr *e = NULL; CK(NULL, dfgpu_expr_column(name, idx, &e)); return keep_e(e); }
static dfgpu_expr *lit(dfgpu_array *scalar) { dfgpu_expr *e = NULL; CK(NULL, dfgpu_expr_literal(scalar, &e)); return keep_e(e); }
static dfgpu_expr *bin(dfgpu_expr *l, int op, dfgpu_expr *r) { dfgpu_expr *e = NULL; CK(NULL, dfgpu_expr_binary(l, op, r, &e)); return keep_e(e); }
static dfgpu_plan *memory(dfgpu_batch *b) { const dfgpu_batch *bs[1] = { b }; int32_t sizes[1] = { 1 }; dfgpu_plan *p = NULL; CK(NULL, dfgpu_plan_memory(bs, sizes, 1, &p)); return keep_p(p); }
static dfgpu_plan *filter(dfgpu_expr *pred, dfgpu_plan *in) { dfgpu_plan *f = NULL, *c = NULL; CK(NULL, dfgpu_plan_filter(pred, in, &f)); keep_p(f); CK(NULL, dfgpu_plan_coalesce_batches(f, 8192, &c)); return keep_p(c); }
static dfgpu_plan *project(dfgpu_plan *in, int n, const char **names, const int *idx) {
  dfgpu_expr *es[8]; for (int i = 0; i < n; i++) es[i] = col(names[i], idx[i]);
  dfgpu_plan *p = NULL; CK(NULL, dfgpu_plan_projection((const dfgpu_expr *const *)es, names, n, in, &p)); return keep_p(p);
}
static dfgpu_plan *join(dfgpu_plan *l, dfgpu_plan *r, dfgpu_expr *lk, dfgpu_expr *rk) {
  const dfgpu_expr *ls[1] = { lk }, *rs[1] = { rk }; int32_t z[1] = { 0 };
  dfgpu_plan *j = NULL, *c = NULL; CK(NULL, dfgpu_plan_hash_join(l, r, ls, rs, 1, NULL, z, z, 0, DFGPU_JOIN_INNER, 1 /* Partitioned */, 0, &j)); keep_p(j);
  CK(NULL, dfgpu_plan_coalesce_batches(j, 8192, &c)); return keep_p(c);
}

int main(void) {
  dfgpu_ctx *ctx = NULL;
  if (dfgpu_ctx_create(0, NULL, &ctx) != DFGPU_OK || !ctx) { fprintf(stderr, "no HIP device (there is no CPU fallback)\n"); return 2; }
  enum { NC = 1500, NO = 15000 };
  const int32_t DATE = 9204;                                     /* 1995-03-15 */
  /* ---- host tables (TPC-H shaped: every order has a customer, every lineitem an order, lineitems clustered by order key) */
  static int64_t c_custkey[NC]; static int32_t c_off[NC + 1]; static char c_seg[NC * 10];
  const char *segs[5] = { "AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY" };
  c_off[0] = 0;
  for (int i = 0; i < NC; i++) { c_custkey[i] = i + 1; const char *s = segs[rnd() % 5]; int L = (int)strlen(s); memcpy(c_seg + c_off[i], s, (size_t)L); c_off[i + 1] = c_off[i] + L; }
  static int64_t o_orderkey[NO], o_custkey[NO]; static int32_t o_orderdate[NO], o_shippriority[NO];
  static int64_t l_orderkey[NO * 7]; static i128 l_price[NO * 7], l_disc[NO * 7]; static int32_t l_shipdate[NO * 7];
  int64_t nl = 0;
  for (int i = 0; i < NO; i++) {
    o_orderkey[i] = (int64_t)(i / 8) * 32 + (i % 8) + 1; o_custkey[i] = (int64_t)(rnd() % NC) + 1; o_orderdate[i] = 8035 + (int32_t)(rnd() % 2406); o_shippriority[i] = 0;
    int lines = 1 + (int)(rnd() % 7);
    for (int k = 0; k < lines; k++, nl++) { l_orderkey[nl] = o_orderkey[i]; l_price[nl] = 90000 + (i128)(rnd() % 10404951); l_disc[nl] = (i128)(rnd() % 11); l_shipdate[nl] = o_orderdate[i] + 1 + (int32_t)(rnd() % 121); }
  }
  /* ---- device columns */
  dfgpu_array_desc sd; memset(&sd, 0, sizeof sd); sd.type = DFGPU_UTF8; sd.length = NC; sd.values = c_seg; sd.offsets = c_off; sd.values_bytes = c_off[NC];
  dfgpu_array *a_seg = NULL; CK(ctx, dfgpu_array_import_host(ctx, &sd, &a_seg));
  const dfgpu_array *cc[2] = { import_fixed(ctx, DFGPU_INT64, 0, 0, c_custkey, NC), a_seg }; const char *cn[2] = { "c_custkey", "c_mktsegment" };
  const dfgpu_array *oc[4] = { import_fixed(ctx, DFGPU_INT64, 0, 0, o_orderkey, NO), import_fixed(ctx, DFGPU_INT64, 0, 0, o_custkey, NO), import_fixed(ctx, DFGPU_DATE32, 0, 0, o_orderdate, NO),
                               import_fixed(ctx, DFGPU_INT32, 0, 0, o_shippriority, NO) };
  const char *on[4] = { "o_orderkey", "o_custkey", "o_orderdate", "o_shippriority" };
  const dfgpu_array *lc[4] = { import_fixed(ctx, DFGPU_INT64, 0, 0, l_orderkey, nl), import_fixed(ctx, DFGPU_DECIMAL128, 15, 2, l_price, nl), import_fixed(ctx, DFGPU_DECIMAL128, 15, 2, l_disc, nl),
                               import_fixed(ctx, DFGPU_DATE32, 0, 0, l_shipdate, nl) };
  const char *ln[4] = { "l_orderkey", "l_extendedprice", "l_discount", "l_shipdate" };
  dfgpu_batch *bc = NULL, *bo = NULL, *bl = NULL;
  CK(ctx, dfgpu_batch_new(cn, cc, 2, &bc)); CK(ctx, dfgpu_batch_new(on, oc, 4, &bo)); CK(ctx, dfgpu_batch_new(ln, lc, 4, &bl));
  /* ---- literals (1-row arrays, ≙ ScalarValue) */
  int32_t seg_off[2] = { 0, 8 }; dfgpu_array_desc ld; memset(&ld, 0, sizeof ld); ld.type = DFGPU_UTF8; ld.length = 1; ld.values = "BUILDING"; ld.offsets = seg_off; ld.values_bytes = 8;
  dfgpu_array *s_seg = NULL; CK(ctx, dfgpu_array_import_host(ctx, &ld, &s_seg));
  dfgpu_array *s_date = import_fixed(ctx, DFGPU_DATE32, 0, 0, &DATE, 1);
  i128 one = 1; dfgpu_array *s_one = import_fixed(ctx, DFGPU_DECIMAL128, 20, 0, &one, 1);
  /* ---- the plan (tpch.q3_plan of the Python builders, node for node) */
  const char *n1[1] = { "c_custkey" }; const int i1[1] = { 0 };
  dfgpu_plan *p_c = project(filter(bin(col("c_mktsegment", 1), DFGPU_OP_EQ, lit(s_seg)), memory(bc)), 1, n1, i1);
  dfgpu_plan *f_o = filter(bin(col("o_orderdate", 2), DFGPU_OP_LT, lit(s_date)), memory(bo));
  dfgpu_plan *j1 = join(p_c, f_o, col("c_custkey", 0), col("o_custkey", 1));            /* c_custkey, o_orderkey, o_custkey, o_orderdate, o_shippriority */
  const char *n2[3] = { "o_orderkey", "o_orderdate", "o_shippriority" }; const int i2[3] = { 1, 3, 4 };
  dfgpu_plan *p_j1 = project(j1, 3, n2, i2);
  const char *n3[3] = { "l_orderkey", "l_extendedprice", "l_discount" }; const int i3[3] = { 0, 1, 2 };
  dfgpu_plan *p_l = project(filter(bin(col("l_shipdate", 3), DFGPU_OP_GT, lit(s_date)), memory(bl)), 3, n3, i3);
  dfgpu_plan *j2 = join(p_j1, p_l, col("o_orderkey", 0), col("l_orderkey", 0));          /* o_orderkey, o_orderdate, o_shippriority, l_orderkey, l_extendedprice, l_discount */
  const char *n4[5] = { "o_orderdate", "o_shippriority", "l_orderkey", "l_extendedprice", "l_discount" }; const int i4[5] = { 1, 2, 3, 4, 5 };
  dfgpu_plan *p_j2 = project(j2, 5, n4, i4);
  dfgpu_expr *revenue = bin(col("l_extendedprice", 3), DFGPU_OP_MUL, bin(lit(s_one), DFGPU_OP_SUB, col("l_discount", 4)));
  const dfgpu_expr *gk[3] = { col("l_orderkey", 2), col("o_orderdate", 0), col("o_shippriority", 1) }; const char *gn[3] = { "l_orderkey", "o_orderdate", "o_shippriority" };
  int32_t kinds[1] = { DFGPU_AGG_SUM }; const dfgpu_expr *args[1] = { revenue }, *filts[1] = { NULL }; const char *an[1] = { "revenue" }; int32_t at[3] = { DFGPU_DECIMAL128, 38, 4 };
  dfgpu_plan *agg = NULL; CK(ctx, dfgpu_plan_aggregate(3 /* Single */, gk, gn, 3, kinds, args, filts, an, at, 1, p_j2, &agg)); keep_p(agg);
  const char *n5[4] = { "l_orderkey", "revenue", "o_orderdate", "o_shippriority" }; const int i5[4] = { 0, 3, 1, 2 };
  dfgpu_plan *proj = project(agg, 4, n5, i5);
  const dfgpu_expr *sk[2] = { col("revenue", 1), col("o_orderdate", 2) }; uint8_t desc[2] = { 1, 0 }, nf[2] = { 1, 0 };
  dfgpu_plan *plan = NULL; CK(ctx, dfgpu_plan_sort(sk, desc, nf, 2, -1, 0, proj, &plan)); keep_p(plan);
  /* ---- execute, pull the single output partition */
  dfgpu_stream *st = NULL; CK(ctx, dfgpu_plan_execute(plan, 0, ctx, 8192, &st));
  int64_t got_n = 0; int64_t *g_key = NULL; i128 *g_rev = NULL; int32_t *g_date = NULL;
  for (;;) {
    dfgpu_batch *b = NULL; CK(ctx, dfgpu_stream_next(st, &b)); if (!b) break;
    int64_t rows = 0; CK(ctx, dfgpu_batch_num_rows(ctx, b, &rows));
    g_key = realloc(g_key, (size_t)(got_n + rows) * 8); g_rev = realloc(g_rev, (size_t)(got_n + rows) * 16); g_date = realloc(g_date, (size_t)(got_n + rows) * 4);
    dfgpu_array *c0 = NULL, *c1 = NULL, *c2 = NULL;
    CK(ctx, dfgpu_batch_column(ctx, b, 0, &c0)); CK(ctx, dfgpu_batch_column(ctx, b, 1, &c1)); CK(ctx, dfgpu_batch_column(ctx, b, 2, &c2));
    dfgpu_array_desc d1; CK(ctx, dfgpu_array_describe(c1, &d1));
    if (d1.type != DFGPU_DECIMAL128 || d1.precision != 38 || d1.scale != 4) { fprintf(stderr, "revenue type (%d, %d, %d)\n", d1.type, d1.precision, d1.scale); return 1; }
    CK(ctx, dfgpu_array_export_host(ctx, c0, g_key + got_n, NULL, NULL)); CK(ctx, dfgpu_array_export_host(ctx, c1, g_rev + got_n, NULL, NULL)); CK(ctx, dfgpu_array_export_host(ctx, c2, g_date + got_n, NULL, NULL));
    dfgpu_array_release(c0); dfgpu_array_release(c1); dfgpu_array_release(c2); dfgpu_batch_free(b); got_n += rows;
  }
  dfgpu_stream_free(st);
  /* ---- the same query on the host: orders are unique per key, so one pass over the (clustered) lineitems per qualifying order */
  int64_t want_n = 0, li = 0, bad = 0; i128 total = 0;
  for (int i = 0; i < NO; i++) {
    i128 rev = 0; int any = 0;
    int seg_ok = (c_off[o_custkey[i]] - c_off[o_custkey[i] - 1] == 8) && memcmp(c_seg + c_off[o_custkey[i] - 1], "BUILDING", 8) == 0;
    for (; li < nl && l_orderkey[li] == o_orderkey[i]; li++)
      if (seg_ok && o_orderdate[i] < DATE && l_shipdate[li] > DATE) { rev += l_price[li] * (100 - l_disc[li]); any = 1; }      /* Decimal(15,2) * (Decimal(20,0) - Decimal(15,2)) -> scale 4 */
    if (!any) continue;
    want_n++; total += rev;
    int found = 0; for (int64_t k = 0; k < got_n; k++) if (g_key[k] == o_orderkey[i]) { found = 1; if (g_rev[k] != rev || g_date[k] != o_orderdate[i]) bad++; break; }
    if (!found) bad++;
  }
  for (int64_t k = 1; k < got_n; k++) if (g_rev[k - 1] < g_rev[k] || (g_rev[k - 1] == g_rev[k] && g_date[k - 1] > g_date[k])) bad++;   /* ORDER BY revenue DESC, o_orderdate */
  printf("q3_native: %lld result rows (host evaluation: %lld), %lld mismatches, revenue total %lld.%04lld\n", (long long)got_n, (long long)want_n, (long long)bad,
         (long long)(total / 10000), (long long)(total % 10000));
  for (int i = 0; i < n_plans; i++) dfgpu_plan_free(plans[i]);
  for (int i = 0; i < n_exprs; i++) dfgpu_expr_free(exprs[i]);
  dfgpu_batch_free(bc); dfgpu_batch_free(bo); dfgpu_batch_free(bl);
  for (int i = 0; i < 2; i++) dfgpu_array_release((dfgpu_array *)cc[i]);
  for (int i = 0; i < 4; i++) { dfgpu_array_release((dfgpu_array *)oc[i]); dfgpu_array_release((dfgpu_array *)lc[i]); }
  dfgpu_array_release(s_seg); dfgpu_array_release(s_date); dfgpu_array_release(s_one);
  free(g_key); free(g_rev); free(g_date);
  dfgpu_ctx_destroy(ctx);
  return (bad == 0 && got_n == want_n && got_n > 0) ? 0 : 1;
}
