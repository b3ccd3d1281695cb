/* dfo_core.c -- CPU oracle: builders, cell access, hashing, take/filter.
 * TEST INFRASTRUCTURE ONLY (see dfo.h). */
#include "dfo_internal.h"
#include <stdarg.h>

static __thread char g_err[512];
void dfo_set_error(const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
const char *dfo_last_error(void) { return g_err; }

void *dfo_xrealloc(void *p, size_t n) {
  void *q = realloc(p, n ? n : 1);
  if (!q) { fprintf(stderr, "dfo: out of memory (%zu)\n", n); abort(); }
  return q;
}

int dfo_type_width(int32_t t) {
  switch (t) {
    case DFO_INT8: case DFO_UINT8: return 1;
    case DFO_INT16: case DFO_UINT16: return 2;
    case DFO_INT32: case DFO_UINT32: case DFO_FLOAT32: case DFO_DATE32: return 4;
    case DFO_INT64: case DFO_UINT64: case DFO_FLOAT64: return 8;
    case DFO_DECIMAL128: return 16;
    default: return 0;
  }
}

i128 dfo_pow10(int k) { i128 r = 1; while (k-- > 0) r *= 10; return r; }
int dfo_decimal_fits(i128 v, int precision) {
  i128 lim = dfo_pow10(precision);
  return v > -lim && v < lim;
}

/* ------------------------------------------------------------------ builders */
dfo_builder *dfo_builder_new(int32_t type, int32_t precision, int32_t scale) {
  dfo_builder *b = (dfo_builder *)calloc(1, sizeof *b);
  b->arr.type = type; b->arr.precision = precision; b->arr.scale = scale;
  if (type == DFO_UTF8) { b->offs = (int32_t *)dfo_xrealloc(NULL, 16 * sizeof(int32_t)); b->offs_cap = 16; b->offs[0] = 0; }
  dfo_builder_finish(b);
  return b;
}
void dfo_builder_free(dfo_builder *b) {
  if (!b) return;
  free(b->vals); free(b->valid); free(b->offs); free(b);
}
void dfo_builder_reserve(dfo_builder *b, int64_t rows, int64_t extra_bytes) {
  int64_t need = b->arr.length + rows;
  int w = dfo_type_width(b->arr.type);
  int64_t vb = b->arr.type == DFO_BOOL ? (need + 7) / 8 + 8 : (b->arr.type == DFO_UTF8 ? b->nbytes + extra_bytes : need * w);
  if (vb > b->vals_cap) {
    int64_t nc = b->vals_cap * 2 > vb ? b->vals_cap * 2 : vb + 64;
    b->vals = (uint8_t *)dfo_xrealloc(b->vals, (size_t)nc);
    memset(b->vals + b->vals_cap, 0, (size_t)(nc - b->vals_cap));
    b->vals_cap = nc;
  }
  int64_t mb = (need + 7) / 8 + 8;
  if (mb > b->valid_cap) {
    int64_t nc = b->valid_cap * 2 > mb ? b->valid_cap * 2 : mb + 64;
    b->valid = (uint8_t *)dfo_xrealloc(b->valid, (size_t)nc);
    memset(b->valid + b->valid_cap, 0, (size_t)(nc - b->valid_cap));
    b->valid_cap = nc;
  }
  if (b->arr.type == DFO_UTF8 && need + 1 > b->offs_cap) {
    int64_t nc = b->offs_cap * 2 > need + 1 ? b->offs_cap * 2 : need + 64;
    b->offs = (int32_t *)dfo_xrealloc(b->offs, (size_t)nc * sizeof(int32_t));
    b->offs_cap = nc;
  }
}
void dfo_builder_finish(dfo_builder *b) {
  b->arr.values = b->vals; b->arr.offsets = b->offs; b->arr.values_bytes = b->nbytes;
  b->arr.validity = b->arr.null_count > 0 ? b->valid : NULL;
  b->arr.dictionary = NULL; b->arr.key_type = 0;
}
const dfo_array *dfo_builder_array(dfo_builder *b) { dfo_builder_finish(b); return &b->arr; }

void dfo_builder_append_null(dfo_builder *b) {
  dfo_builder_reserve(b, 1, 0);
  int64_t i = b->arr.length;
  dfo_bit_set(b->valid, i, 0);
  int w = dfo_type_width(b->arr.type);
  if (w) memset(b->vals + i * w, 0, (size_t)w);
  else if (b->arr.type == DFO_BOOL) dfo_bit_set(b->vals, i, 0);
  else if (b->arr.type == DFO_UTF8) b->offs[i + 1] = b->offs[i];
  b->arr.length++; b->arr.null_count++;
  dfo_builder_finish(b);
}
void dfo_builder_append_value(dfo_builder *b, const void *v) {
  dfo_builder_reserve(b, 1, 0);
  int64_t i = b->arr.length; int w = dfo_type_width(b->arr.type);
  memcpy(b->vals + i * w, v, (size_t)w);
  dfo_bit_set(b->valid, i, 1);
  b->arr.length++;
  dfo_builder_finish(b);
}
void dfo_builder_append_bool(dfo_builder *b, int v) {
  dfo_builder_reserve(b, 1, 0);
  int64_t i = b->arr.length;
  dfo_bit_set(b->vals, i, v); dfo_bit_set(b->valid, i, 1);
  b->arr.length++;
  dfo_builder_finish(b);
}
void dfo_builder_append_utf8(dfo_builder *b, const uint8_t *p, int64_t len) {
  dfo_builder_reserve(b, 1, len);
  int64_t i = b->arr.length;
  if (len) memcpy(b->vals + b->nbytes, p, (size_t)len);
  b->nbytes += len; b->offs[i + 1] = (int32_t)b->nbytes;
  dfo_bit_set(b->valid, i, 1);
  b->arr.length++;
  dfo_builder_finish(b);
}
void dfo_builder_append_cell(dfo_builder *b, const dfo_array *src, int64_t i) {
  const dfo_array *a = dfo_resolve(src, &i);
  if (!a) { dfo_builder_append_null(b); return; }
  if (a->type == DFO_BOOL) dfo_builder_append_bool(b, dfo_bit((const uint8_t *)a->values, i));
  else if (a->type == DFO_UTF8)
    dfo_builder_append_utf8(b, (const uint8_t *)a->values + a->offsets[i], a->offsets[i + 1] - a->offsets[i]);
  else dfo_builder_append_value(b, (const uint8_t *)a->values + i * dfo_type_width(a->type));
}

/* ------------------------------------------------------------------ cells */
uint64_t dfo_hash_cell(const dfo_array *a, int64_t i, uint64_t seed) {
  switch (a->type) {
    case DFO_BOOL: return dfo_mix64((uint64_t)dfo_bit((const uint8_t *)a->values, i) ^ seed);
    case DFO_INT8: return dfo_mix64((uint64_t)(int64_t)((const int8_t *)a->values)[i] ^ seed);
    case DFO_INT16: return dfo_mix64((uint64_t)(int64_t)((const int16_t *)a->values)[i] ^ seed);
    case DFO_INT32: case DFO_DATE32: return dfo_mix64((uint64_t)(int64_t)((const int32_t *)a->values)[i] ^ seed);
    case DFO_INT64: return dfo_mix64((uint64_t)((const int64_t *)a->values)[i] ^ seed);
    case DFO_UINT8: return dfo_mix64((uint64_t)((const uint8_t *)a->values)[i] ^ seed);
    case DFO_UINT16: return dfo_mix64((uint64_t)((const uint16_t *)a->values)[i] ^ seed);
    case DFO_UINT32: case DFO_FLOAT32: return dfo_mix64((uint64_t)((const uint32_t *)a->values)[i] ^ seed);
    case DFO_UINT64: case DFO_FLOAT64: return dfo_mix64(((const uint64_t *)a->values)[i] ^ seed);
    case DFO_DECIMAL128: {
      const uint64_t *p = (const uint64_t *)a->values + 2 * i; /* little endian lo, hi */
      return dfo_mix64(dfo_mix64(p[1] ^ seed) ^ p[0]);
    }
    case DFO_UTF8: {
      const uint8_t *p = (const uint8_t *)a->values + a->offsets[i];
      int64_t len = a->offsets[i + 1] - a->offsets[i];
      uint64_t h = dfo_mix64((uint64_t)len ^ seed);
      for (int64_t o = 0; o < len; o += 8) {
        uint64_t w = 0; int64_t m = len - o < 8 ? len - o : 8;
        memcpy(&w, p + o, (size_t)m);
        h = dfo_mix64(h ^ w);
      }
      return h;
    }
    default: return 0;
  }
}

static inline int cmp_i128(i128 x, i128 y) { return x < y ? -1 : (x > y ? 1 : 0); }
/* IEEE 754 totalOrder as arrow-ord cmp/sort use (f64::total_cmp) */
static inline int64_t total_f64(uint64_t b) { int64_t s = (int64_t)b; return s ^ (int64_t)((uint64_t)(s >> 63) >> 1); }
static inline int32_t total_f32(uint32_t b) { int32_t s = (int32_t)b; return s ^ (int32_t)((uint32_t)(s >> 31) >> 1); }

int dfo_cell_cmp(const dfo_array *a, int64_t i, const dfo_array *b, int64_t j) {
#define CMPT(T) { T x = ((const T *)a->values)[i], y = ((const T *)b->values)[j]; return x < y ? -1 : (x > y ? 1 : 0); }
  switch (a->type) {
    case DFO_BOOL: { int x = dfo_bit((const uint8_t *)a->values, i), y = dfo_bit((const uint8_t *)b->values, j); return x - y; }
    case DFO_INT8: CMPT(int8_t) case DFO_INT16: CMPT(int16_t)
    case DFO_INT32: case DFO_DATE32: CMPT(int32_t) case DFO_INT64: CMPT(int64_t)
    case DFO_UINT8: CMPT(uint8_t) case DFO_UINT16: CMPT(uint16_t)
    case DFO_UINT32: CMPT(uint32_t) case DFO_UINT64: CMPT(uint64_t)
    case DFO_FLOAT32: { int32_t x = total_f32(((const uint32_t *)a->values)[i]), y = total_f32(((const uint32_t *)b->values)[j]); return x < y ? -1 : (x > y ? 1 : 0); }
    case DFO_FLOAT64: { int64_t x = total_f64(((const uint64_t *)a->values)[i]), y = total_f64(((const uint64_t *)b->values)[j]); return x < y ? -1 : (x > y ? 1 : 0); }
    case DFO_DECIMAL128: { i128 x, y; memcpy(&x, (const uint8_t *)a->values + 16 * i, 16); memcpy(&y, (const uint8_t *)b->values + 16 * j, 16); return cmp_i128(x, y); }
    case DFO_UTF8: {
      int64_t la = a->offsets[i + 1] - a->offsets[i], lb = b->offsets[j + 1] - b->offsets[j];
      int64_t m = la < lb ? la : lb;
      int c = m ? memcmp((const uint8_t *)a->values + a->offsets[i], (const uint8_t *)b->values + b->offsets[j], (size_t)m) : 0;
      if (c) return c < 0 ? -1 : 1;
      return la < lb ? -1 : (la > lb ? 1 : 0);
    }
    default: return 0;
  }
#undef CMPT
}
int dfo_cell_equal(const dfo_array *a, int64_t i, const dfo_array *b, int64_t j) {
  return dfo_cell_cmp(a, i, b, j) == 0;
}

/* ------------------------------------------------------------------ a1 create_hashes */
/* hash_utils.rs:357-417: column 0 assigns (rehash=false), later columns combine_hashes(new, old);
 * NULL cells leave the running hash untouched (:117-131, :203-206); dictionaries hash the VALUE
 * (:182-213); force_hash_collisions => all zero (:306-318). */
void dfo_create_hashes(const dfo_array *const *cols, int k, int64_t n, uint64_t seed,
                       int force_collisions, uint64_t *out) {
  for (int64_t i = 0; i < n; i++) out[i] = 0;
  if (force_collisions) return;
  for (int c = 0; c < k; c++) {
    for (int64_t i = 0; i < n; i++) {
      int64_t r = i;
      const dfo_array *a = dfo_resolve(cols[c], &r);
      if (!a) continue;
      uint64_t h = dfo_hash_cell(a, r, seed);
      out[i] = c == 0 ? h : dfo_combine_hashes(h, out[i]);
    }
  }
}

/* ------------------------------------------------------------------ arrow-select take / filter */
/* arrow_select::take: null index => null output (used at joins/utils.rs:1216,1224). */
int dfo_take(const dfo_array *a, const int64_t *indices, int64_t n, dfo_builder **out) {
  const dfo_array *la = a->type == DFO_DICTIONARY ? a->dictionary : a;
  dfo_builder *b = dfo_builder_new(la->type, la->precision, la->scale);
  for (int64_t i = 0; i < n; i++) {
    if (indices[i] < 0) dfo_builder_append_null(b);
    else if (indices[i] >= a->length) { dfo_set_error("take: index %lld out of bounds", (long long)indices[i]); dfo_builder_free(b); return 1; }
    else dfo_builder_append_cell(b, a, indices[i]);
  }
  *out = b; return 0;
}
/* arrow_select::filter: keep rows whose mask is valid AND true (filter.rs:315-327). */
int dfo_filter(const dfo_array *a, const dfo_array *mask, dfo_builder **out) {
  if (mask->type != DFO_BOOL || mask->length != a->length) { dfo_set_error("filter: bad mask"); return 1; }
  const dfo_array *la = a->type == DFO_DICTIONARY ? a->dictionary : a;
  dfo_builder *b = dfo_builder_new(la->type, la->precision, la->scale);
  for (int64_t i = 0; i < a->length; i++)
    if (dfo_valid(mask, i) && dfo_bit((const uint8_t *)mask->values, i)) dfo_builder_append_cell(b, a, i);
  *out = b; return 0;
}
