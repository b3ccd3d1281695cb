/* dfo_internal.h -- shared helpers of the CPU oracle (test infrastructure only). */
#ifndef DFO_INTERNAL_H
#define DFO_INTERNAL_H
#include "dfo.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef __int128 i128;
typedef unsigned __int128 u128;

void dfo_set_error(const char *fmt, ...);

static inline int dfo_bit(const uint8_t *bits, int64_t i) { return (bits[i >> 3] >> (i & 7)) & 1; }
static inline void dfo_bit_set(uint8_t *bits, int64_t i, int v) {
  if (v) bits[i >> 3] |= (uint8_t)(1u << (i & 7)); else bits[i >> 3] &= (uint8_t)~(1u << (i & 7));
}
static inline int dfo_valid(const dfo_array *a, int64_t i) {
  return a->validity == NULL || dfo_bit(a->validity, i);
}
static inline int64_t dfo_key_at(const dfo_array *a, int64_t i) {
  switch (a->key_type) {
    case DFO_INT8: return ((const int8_t *)a->values)[i];
    case DFO_INT16: return ((const int16_t *)a->values)[i];
    case DFO_INT32: return ((const int32_t *)a->values)[i];
    case DFO_INT64: return ((const int64_t *)a->values)[i];
    case DFO_UINT8: return ((const uint8_t *)a->values)[i];
    case DFO_UINT16: return ((const uint16_t *)a->values)[i];
    case DFO_UINT32: return ((const uint32_t *)a->values)[i];
    default: return (int64_t)((const uint64_t *)a->values)[i];
  }
}
/* Resolve dictionary indirection: returns the value array and rewrites *i.  Returns NULL
 * (cell is null) if the key or the referenced value is null. */
static inline const dfo_array *dfo_resolve(const dfo_array *a, int64_t *i) {
  if (!dfo_valid(a, *i)) return NULL;
  if (a->type == DFO_DICTIONARY) {
    int64_t k = dfo_key_at(a, *i);
    const dfo_array *d = a->dictionary;
    if (!dfo_valid(d, k)) return NULL;
    *i = k;
    return d;
  }
  return a;
}
static inline int32_t dfo_logical_type(const dfo_array *a) {
  return a->type == DFO_DICTIONARY ? a->dictionary->type : a->type;
}

/* splitmix64 finaliser -- the documented product hash */
static inline uint64_t dfo_mix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ULL;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
  return x ^ (x >> 31);
}
/* hash_utils.rs:38-41 */
static inline uint64_t dfo_combine_hashes(uint64_t l, uint64_t r) {
  uint64_t h = (uint64_t)(17 * 37) + l;
  return h * 37 + r;
}
uint64_t dfo_hash_cell(const dfo_array *a, int64_t i, uint64_t seed); /* a non-null, resolved */
int dfo_cell_equal(const dfo_array *a, int64_t i, const dfo_array *b, int64_t j); /* both non-null, resolved */
int dfo_cell_cmp(const dfo_array *a, int64_t i, const dfo_array *b, int64_t j);   /* total order */

/* 10^k as i128, k in [0,38] */
i128 dfo_pow10(int k);
int dfo_decimal_fits(i128 v, int precision);

void *dfo_xrealloc(void *p, size_t n);
void dfo_builder_reserve(dfo_builder *b, int64_t rows, int64_t extra_bytes);
void dfo_builder_append_value(dfo_builder *b, const void *v);           /* fixed width */
void dfo_builder_append_bool(dfo_builder *b, int v);
void dfo_builder_append_utf8(dfo_builder *b, const uint8_t *p, int64_t len);
void dfo_builder_finish(dfo_builder *b);

#endif
