/* dfo_agg.c -- CPU oracle restatement of GroupValues + GroupsAccumulator (TEST INFRASTRUCTURE ONLY).
 *
 * Follows:
 *   GroupValues::intern/emit        physical-plan/src/aggregates/group_values/primitive.rs:112-149,163-209
 *                                   row.rs:94-146,167-222; bytes.rs:44-73  (ids in FIRST-SEEN order; NULL is
 *                                   a key value of its own, primitive.rs:118-122)
 *   PrimitiveGroupsAccumulator      physical-expr/src/aggregate/groups_accumulator/prim_op.rs:36-140
 *   NullState::accumulate / build   groups_accumulator/accumulate.rs:126-233, :332-358
 *   SUM closure add_wrapping        aggregate/sum.rs:132-142
 *   MIN/MAX closures + start values aggregate/min_max.rs:102-139
 *   AvgGroupsAccumulator            aggregate/average.rs:392-568 (state = (count u64, sum))
 *   DecimalAverager::avg            aggregate/utils.rs:55-125
 *   CountGroupsAccumulator          aggregate/count.rs:93-192
 *   sum/avg return types            expr/src/type_coercion/aggregates.rs:397-416, :455-477
 */
#include "dfo_internal.h"
#include <math.h>

/* ------------------------------------------------------------------ GroupValues */
struct dfo_groups {
  int nkeys; dfo_builder **keys;   /* one builder per key column: the group's key values in id order */
  int64_t n;                       /* number of groups */
  uint64_t *slot_hash; int64_t *slot_gid; uint64_t mask; int64_t used;
  uint64_t *ghash;                 /* hash per group (for rehash) */
  int64_t ghash_cap;
};

dfo_groups *dfo_groups_new(int nkeys, const int32_t *types, const int32_t *precisions, const int32_t *scales) {
  dfo_groups *g = (dfo_groups *)calloc(1, sizeof *g);
  g->nkeys = nkeys; g->keys = (dfo_builder **)calloc((size_t)nkeys, sizeof(dfo_builder *));
  for (int c = 0; c < nkeys; c++) g->keys[c] = dfo_builder_new(types[c], precisions ? precisions[c] : 0, scales ? scales[c] : 0);
  g->mask = 1023; g->slot_hash = (uint64_t *)calloc(1024, 8);
  g->slot_gid = (int64_t *)dfo_xrealloc(NULL, 1024 * 8); memset(g->slot_gid, 0xff, 1024 * 8);
  return g;
}
void dfo_groups_free(dfo_groups *g) {
  if (!g) return;
  for (int c = 0; c < g->nkeys; c++) dfo_builder_free(g->keys[c]);
  free(g->keys); free(g->slot_hash); free(g->slot_gid); free(g->ghash); free(g);
}
int64_t dfo_groups_len(const dfo_groups *g) { return g->n; }
const dfo_array *dfo_groups_emit(dfo_groups *g, int col) { return dfo_builder_array(g->keys[col]); }

static void groups_grow(dfo_groups *g) {
  uint64_t ncap = (g->mask + 1) * 2;
  free(g->slot_hash); free(g->slot_gid);
  g->slot_hash = (uint64_t *)calloc(ncap, 8);
  g->slot_gid = (int64_t *)dfo_xrealloc(NULL, ncap * 8); memset(g->slot_gid, 0xff, ncap * 8);
  g->mask = ncap - 1;
  for (int64_t i = 0; i < g->n; i++) {
    uint64_t s = dfo_mix64(g->ghash[i]) & g->mask;
    while (g->slot_gid[s] >= 0) s = (s + 1) & g->mask;
    g->slot_gid[s] = i; g->slot_hash[s] = g->ghash[i];
  }
}

static int group_key_equal(dfo_groups *g, int64_t gid, const dfo_array *const *cols, int64_t row) {
  for (int c = 0; c < g->nkeys; c++) {
    int64_t i = row, j = gid;
    const dfo_array *a = dfo_resolve(cols[c], &i);
    const dfo_array *b = dfo_resolve(&g->keys[c]->arr, &j);
    if (!a || !b) { if (!a && !b) continue; return 0; }
    if (!dfo_cell_equal(a, i, b, j)) return 0;
  }
  return 1;
}

void dfo_groups_intern(dfo_groups *g, const dfo_array *const *cols, int64_t n, int64_t *out_ids) {
  uint64_t *hashes = (uint64_t *)dfo_xrealloc(NULL, (size_t)(n + 1) * 8);
  dfo_create_hashes(cols, g->nkeys, n, 0x5851f42d4c957f2dULL, 0, hashes);
  for (int64_t r = 0; r < n; r++) {
    if ((uint64_t)(g->n + 1) * 2 > g->mask + 1) groups_grow(g);
    uint64_t h = hashes[r], s = dfo_mix64(h) & g->mask;
    int64_t gid = -1;
    for (;;) {
      if (g->slot_gid[s] < 0) break;
      if (g->slot_hash[s] == h && group_key_equal(g, g->slot_gid[s], cols, r)) { gid = g->slot_gid[s]; break; }
      s = (s + 1) & g->mask;
    }
    if (gid < 0) {
      gid = g->n++;
      for (int c = 0; c < g->nkeys; c++) { dfo_builder_append_cell(g->keys[c], cols[c], r); dfo_builder_finish(g->keys[c]); }
      if (gid >= g->ghash_cap) { g->ghash_cap = g->ghash_cap ? g->ghash_cap * 2 : 1024; g->ghash = (uint64_t *)dfo_xrealloc(g->ghash, (size_t)g->ghash_cap * 8); }
      g->ghash[gid] = h; g->slot_gid[s] = gid; g->slot_hash[s] = h;
    }
    out_ids[r] = gid;
  }
  free(hashes);
}

/* ------------------------------------------------------------------ GroupsAccumulator */
struct dfo_acc {
  int kind; int32_t in_type, in_precision, in_scale;
  int32_t state_type, state_precision, state_scale;   /* type of the value/sum slot */
  int32_t out_type, out_precision, out_scale;
  int64_t n;            /* groups allocated */
  i128 *vi; double *vf; uint64_t *vu;    /* value / sum per group (by state_type class) */
  uint64_t *counts;     /* AVG counts (u64) ; COUNT uses ci */
  int64_t *ci;
  uint8_t *seen; int64_t cap;
  dfo_builder *out0, *out1;
};

static int is_signed_int(int t) { return t == DFO_INT8 || t == DFO_INT16 || t == DFO_INT32 || t == DFO_INT64 || t == DFO_DATE32; }
static int is_unsigned_int(int t) { return t == DFO_UINT8 || t == DFO_UINT16 || t == DFO_UINT32 || t == DFO_UINT64; }
static int is_float(int t) { return t == DFO_FLOAT32 || t == DFO_FLOAT64; }
static int imin(int a, int b) { return a < b ? a : b; }

dfo_acc *dfo_acc_new(int kind, int32_t in_type, int32_t in_precision, int32_t in_scale) {
  dfo_acc *a = (dfo_acc *)calloc(1, sizeof *a);
  a->kind = kind; a->in_type = in_type; a->in_precision = in_precision; a->in_scale = in_scale;
  switch (kind) {
    case DFO_AGG_COUNT: a->state_type = a->out_type = DFO_INT64; break;
    case DFO_AGG_SUM:
      /* sum_return_type (type_coercion/aggregates.rs:397-416) */
      if (is_signed_int(in_type) && in_type != DFO_DATE32) a->state_type = DFO_INT64;
      else if (is_unsigned_int(in_type)) a->state_type = DFO_UINT64;
      else if (is_float(in_type)) a->state_type = DFO_FLOAT64;
      else if (in_type == DFO_DECIMAL128) { a->state_type = DFO_DECIMAL128; a->state_precision = imin(38, in_precision + 10); a->state_scale = in_scale; }
      else { free(a); dfo_set_error("SUM: unsupported input type %d", in_type); return NULL; }
      a->out_type = a->state_type; a->out_precision = a->state_precision; a->out_scale = a->state_scale;
      break;
    case DFO_AGG_AVG:
      /* avg_return_type / avg_sum_type (aggregates.rs:455-505) */
      if (in_type == DFO_DECIMAL128) {
        a->state_type = DFO_DECIMAL128; a->state_precision = imin(38, in_precision + 10); a->state_scale = in_scale;
        a->out_type = DFO_DECIMAL128; a->out_precision = imin(38, in_precision + 4); a->out_scale = imin(38, in_scale + 4);
      } else if (in_type == DFO_FLOAT64) { a->state_type = a->out_type = DFO_FLOAT64; }
      else { free(a); dfo_set_error("AVG: input must be coerced to Float64/Decimal128 (got %d)", in_type); return NULL; }
      break;
    case DFO_AGG_MIN: case DFO_AGG_MAX:
      if (!dfo_type_width(in_type)) { free(a); dfo_set_error("MIN/MAX: unsupported input type %d", in_type); return NULL; }
      a->state_type = a->out_type = in_type; a->state_precision = a->out_precision = in_precision; a->state_scale = a->out_scale = in_scale;
      break;
    default: free(a); dfo_set_error("unknown aggregate kind"); return NULL;
  }
  return a;
}
void dfo_acc_free(dfo_acc *a) {
  if (!a) return;
  free(a->vi); free(a->vf); free(a->vu); free(a->counts); free(a->ci); free(a->seen);
  dfo_builder_free(a->out0); dfo_builder_free(a->out1); free(a);
}

static i128 int_min_of(int t) {
  switch (t) { case DFO_INT8: return INT8_MIN; case DFO_INT16: return INT16_MIN; case DFO_INT32: case DFO_DATE32: return INT32_MIN;
    case DFO_INT64: return INT64_MIN; case DFO_DECIMAL128: return (i128)((u128)1 << 127); default: return 0; }
}
static i128 int_max_of(int t) {
  switch (t) { case DFO_INT8: return INT8_MAX; case DFO_INT16: return INT16_MAX; case DFO_INT32: case DFO_DATE32: return INT32_MAX;
    case DFO_INT64: return INT64_MAX; case DFO_UINT8: return UINT8_MAX; case DFO_UINT16: return UINT16_MAX; case DFO_UINT32: return UINT32_MAX;
    case DFO_UINT64: return (i128)UINT64_MAX; case DFO_DECIMAL128: return (i128)(((u128)1 << 127) - 1); default: return 0; }
}

static void acc_resize(dfo_acc *a, int64_t total) {
  if (total <= a->n) return;
  if (total > a->cap) {
    int64_t nc = a->cap ? a->cap : 64; while (nc < total) nc *= 2;
    a->vi = (i128 *)dfo_xrealloc(a->vi, (size_t)nc * 16); a->vf = (double *)dfo_xrealloc(a->vf, (size_t)nc * 8);
    a->vu = (uint64_t *)dfo_xrealloc(a->vu, (size_t)nc * 8);
    a->counts = (uint64_t *)dfo_xrealloc(a->counts, (size_t)nc * 8); a->ci = (int64_t *)dfo_xrealloc(a->ci, (size_t)nc * 8);
    a->seen = (uint8_t *)dfo_xrealloc(a->seen, (size_t)nc);
    a->cap = nc;
  }
  for (int64_t i = a->n; i < total; i++) {
    a->counts[i] = 0; a->ci[i] = 0; a->seen[i] = 0; a->vu[i] = 0;
    if (a->kind == DFO_AGG_MIN) { a->vi[i] = int_max_of(a->state_type); a->vf[i] = a->state_type == DFO_FLOAT32 ? 3.40282346638528859812e+38 : 1.7976931348623157e308; a->vu[i] = (uint64_t)int_max_of(a->state_type); }
    else if (a->kind == DFO_AGG_MAX) { a->vi[i] = int_min_of(a->state_type); a->vf[i] = a->state_type == DFO_FLOAT32 ? -3.40282346638528859812e+38 : -1.7976931348623157e308; a->vu[i] = 0; }
    else { a->vi[i] = 0; a->vf[i] = 0.0; }
  }
  a->n = total;
}

static i128 read_int(const dfo_array *a, int64_t i) {
  switch (a->type) {
    case DFO_INT8: return ((const int8_t *)a->values)[i]; case DFO_INT16: return ((const int16_t *)a->values)[i];
    case DFO_INT32: case DFO_DATE32: return ((const int32_t *)a->values)[i]; case DFO_INT64: return ((const int64_t *)a->values)[i];
    case DFO_UINT8: return ((const uint8_t *)a->values)[i]; case DFO_UINT16: return ((const uint16_t *)a->values)[i];
    case DFO_UINT32: return ((const uint32_t *)a->values)[i]; case DFO_UINT64: return (i128)((const uint64_t *)a->values)[i];
    case DFO_DECIMAL128: { i128 v; memcpy(&v, (const uint8_t *)a->values + 16 * i, 16); return v; }
    default: return 0;
  }
}
static double read_f(const dfo_array *a, int64_t i) {
  return a->type == DFO_FLOAT32 ? (double)((const float *)a->values)[i] : ((const double *)a->values)[i];
}
static int filter_pass(const dfo_array *f, int64_t i) {
  return !f || (dfo_valid(f, i) && dfo_bit((const uint8_t *)f->values, i));
}

/* NullState::accumulate: skip NULL values and rows whose filter is not Some(true); mark group seen. */
static void accumulate_value(dfo_acc *a, const dfo_array *v, int64_t i, int64_t g, int kind) {
  a->seen[g] = 1;
  if (is_float(a->state_type)) {
    double x = read_f(v, i);
    if (kind == DFO_AGG_SUM) a->vf[g] += x;                      /* sequential add in row order (prim_op.rs:101-109) */
    else if (kind == DFO_AGG_MIN) {
      if (a->state_type == DFO_FLOAT32) { if ((float)a->vf[g] > (float)x) a->vf[g] = x; } else if (a->vf[g] > x) a->vf[g] = x;
    } else { if (a->vf[g] < x) a->vf[g] = x; }
  } else if (is_unsigned_int(a->state_type)) {
    uint64_t x = (uint64_t)read_int(v, i);
    if (kind == DFO_AGG_SUM) a->vu[g] += x; else if (kind == DFO_AGG_MIN) { if (a->vu[g] > x) a->vu[g] = x; } else if (a->vu[g] < x) a->vu[g] = x;
  } else {
    i128 x = read_int(v, i);
    if (kind == DFO_AGG_SUM) {
      if (a->state_type == DFO_INT64) a->vi[g] = (i128)(int64_t)((uint64_t)(int64_t)a->vi[g] + (uint64_t)(int64_t)x);   /* add_wrapping i64 */
      else a->vi[g] = (i128)((u128)a->vi[g] + (u128)x);                                                             /* add_wrapping i128 */
    } else if (kind == DFO_AGG_MIN) { if (a->vi[g] > x) a->vi[g] = x; } else if (a->vi[g] < x) a->vi[g] = x;
  }
}

int dfo_acc_update_batch(dfo_acc *a, const dfo_array *values, const int64_t *gids, const dfo_array *opt_filter,
                         int64_t n, int64_t total) {
  acc_resize(a, total);
  for (int64_t i = 0; i < n; i++) {
    int64_t g = gids[i];
    if (g < 0 || g >= total) { dfo_set_error("group id out of range"); return 1; }
    if (!filter_pass(opt_filter, i)) continue;
    if (a->kind == DFO_AGG_COUNT) {                     /* accumulate_indices (accumulate.rs:363-447) */
      int64_t r = i;
      if (values == NULL || dfo_resolve(values, &r) != NULL) a->ci[g] += 1;
      continue;
    }
    int64_t r = i; const dfo_array *v = dfo_resolve(values, &r);
    if (!v) continue;
    if (a->kind == DFO_AGG_AVG) { accumulate_value(a, v, r, g, DFO_AGG_SUM); a->counts[g] += 1; }
    else accumulate_value(a, v, r, g, a->kind);
  }
  return 0;
}

int dfo_acc_merge_batch(dfo_acc *a, const dfo_array *const *st, int nst, const int64_t *gids,
                        const dfo_array *opt_filter, int64_t n, int64_t total) {
  acc_resize(a, total);
  if (a->kind == DFO_AGG_COUNT) {                       /* count.rs:135-170: partial counts are never null */
    if (nst != 1) { dfo_set_error("COUNT merge expects 1 state"); return 1; }
    for (int64_t i = 0; i < n; i++) if (filter_pass(opt_filter, i)) a->ci[gids[i]] += ((const int64_t *)st[0]->values)[i];
    return 0;
  }
  if (a->kind == DFO_AGG_AVG) {                         /* average.rs:472-509 */
    if (nst != 2) { dfo_set_error("AVG merge expects 2 states"); return 1; }
    for (int64_t i = 0; i < n; i++) {
      if (!filter_pass(opt_filter, i)) continue;
      int64_t g = gids[i];
      if (dfo_valid(st[0], i)) { a->counts[g] += ((const uint64_t *)st[0]->values)[i]; a->seen[g] = 1; }
      if (dfo_valid(st[1], i)) accumulate_value(a, st[1], i, g, DFO_AGG_SUM);
    }
    return 0;
  }
  if (nst != 1) { dfo_set_error("merge expects 1 state"); return 1; }
  return dfo_acc_update_batch(a, st[0], gids, opt_filter, n, total);   /* prim_op.rs:119-127 */
}

static void emit_slot(dfo_acc *a, dfo_builder *b, int64_t g, int valid) {
  if (!valid) { /* keep the raw slot value under a null bit like PrimitiveArray::new(values, nulls) */
    dfo_builder_append_null(b); return;
  }
  switch (b->arr.type) {
    case DFO_FLOAT64: { double v = a->vf[g]; dfo_builder_append_value(b, &v); break; }
    case DFO_FLOAT32: { float v = (float)a->vf[g]; dfo_builder_append_value(b, &v); break; }
    case DFO_UINT8: case DFO_UINT16: case DFO_UINT32: case DFO_UINT64: { uint64_t v = a->vu[g]; dfo_builder_append_value(b, &v); break; }
    default: { i128 v = a->vi[g]; dfo_builder_append_value(b, &v); break; } /* little endian: low bytes first */
  }
}

int dfo_acc_evaluate(dfo_acc *a, const dfo_array **out) {
  dfo_builder_free(a->out0);
  a->out0 = dfo_builder_new(a->out_type, a->out_precision, a->out_scale);
  if (a->kind == DFO_AGG_COUNT) { for (int64_t g = 0; g < a->n; g++) dfo_builder_append_value(a->out0, &a->ci[g]); }
  else if (a->kind == DFO_AGG_AVG) {
    for (int64_t g = 0; g < a->n; g++) {
      if (!a->seen[g]) { dfo_builder_append_null(a->out0); continue; }
      if (a->out_type == DFO_FLOAT64) { double v = a->vf[g] / (double)a->counts[g]; dfo_builder_append_value(a->out0, &v); }   /* average.rs:166 */
      else {
        /* DecimalAverager::avg (utils.rs:108-124): sum * (10^target_scale / 10^sum_scale) checked, / count wrapping, precision check */
        i128 factor = dfo_pow10(a->out_scale) / dfo_pow10(a->state_scale), prod;
        if (__builtin_mul_overflow(a->vi[g], factor, &prod)) { dfo_set_error("Arithmetic Overflow in AvgAccumulator"); return 1; }
        i128 v = prod / (i128)a->counts[g];
        if (!dfo_decimal_fits(v, a->out_precision)) { dfo_set_error("Arithmetic Overflow in AvgAccumulator"); return 1; }
        dfo_builder_append_value(a->out0, &v);
      }
    }
  } else { for (int64_t g = 0; g < a->n; g++) emit_slot(a, a->out0, g, a->seen[g]); }
  *out = dfo_builder_array(a->out0);
  return 0;
}

int dfo_acc_state(dfo_acc *a, const dfo_array **o0, const dfo_array **o1, int *nst) {
  if (a->kind != DFO_AGG_AVG) { *nst = 1; *o1 = NULL; return dfo_acc_evaluate(a, o0); }
  dfo_builder_free(a->out0); dfo_builder_free(a->out1);
  a->out0 = dfo_builder_new(DFO_UINT64, 0, 0);
  a->out1 = dfo_builder_new(a->state_type, a->state_precision, a->state_scale);
  for (int64_t g = 0; g < a->n; g++) {
    if (!a->seen[g]) { dfo_builder_append_null(a->out0); dfo_builder_append_null(a->out1); continue; }
    dfo_builder_append_value(a->out0, &a->counts[g]); emit_slot(a, a->out1, g, 1);
  }
  *o0 = dfo_builder_array(a->out0); *o1 = dfo_builder_array(a->out1); *nst = 2;
  return 0;
}
