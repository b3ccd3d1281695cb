/* dfo_expr.c -- CPU oracle restatement of the PhysicalExpr evaluation kernels (TEST INFRASTRUCTURE ONLY).
 *
 * Follows datafusion/physical-expr/src/expressions/binary.rs:259-315 (BinaryExpr::evaluate),
 * datum.rs:28-58 (apply / apply_cmp with scalar Datum broadcast), binary.rs:563-586 (and_kleene /
 * or_kleene), not.rs:71, is_null.rs:74, negative.rs:79, cast.rs:121 (DEFAULT_DATAFUSION_CAST_OPTIONS:
 * safe = false => overflow is an error), in_list.rs:349.
 *
 * The arithmetic itself lives in arrow-arith / arrow-ord / arrow-cast 50.0.0 (NOT under /root/reference,
 * pinned in datafusion-cli/Cargo.lock).  Published semantics restated here:
 *   - integer add/sub/mul are the *_wrapping kernels; div/rem are checked (DivideByZero error)
 *   - Decimal128: add/sub -> (min(38, max(p1-s1,p2-s2)+max(s1,s2)+1), max(s1,s2)); mul -> (min(38,p1+p2+1), s1+s2);
 *     div -> scale min(38, s1+4), precision min(38, p1 + (scale - s1 + s2)); rem -> (min(p1-s1,p2-s2)+max(s1,s2), max(s1,s2));
 *     all decimal arithmetic is overflow-CHECKED on the i128 payload
 *     (dtype pins: sqllogictest/test_files/decimal.slt:209-211, :262-264, :314-316, :340-342; tpch/q1.slt.part)
 *   - comparisons use IEEE-754 totalOrder for floats; NULL op x = NULL except distinct / not_distinct
 */
#include "dfo_internal.h"
#include <math.h>

static int is_sint(int t) { return t == DFO_INT8 || t == DFO_INT16 || t == DFO_INT32 || t == DFO_INT64; }
static int is_uint(int t) { return t == DFO_UINT8 || t == DFO_UINT16 || t == DFO_UINT32 || t == DFO_UINT64; }
static int is_flt(int t) { return t == DFO_FLOAT32 || t == DFO_FLOAT64; }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

static i128 get_int(const dfo_array *a, int64_t i) {
  switch (a->type) {
    case DFO_BOOL: return dfo_bit((const uint8_t *)a->values, i);
    case DFO_INT8: return ((const int8_t *)a->values)[i]; case DFO_INT16: return ((const int16_t *)a->values)[i];
    case DFO_INT32: case DFO_DATE32: return ((const int32_t *)a->values)[i]; case DFO_INT64: return ((const int64_t *)a->values)[i];
    case DFO_UINT8: return ((const uint8_t *)a->values)[i]; case DFO_UINT16: return ((const uint16_t *)a->values)[i];
    case DFO_UINT32: return ((const uint32_t *)a->values)[i]; case DFO_UINT64: return (i128)((const uint64_t *)a->values)[i];
    case DFO_DECIMAL128: { i128 v; memcpy(&v, (const uint8_t *)a->values + 16 * i, 16); return v; }
    default: return 0;
  }
}
static double get_f(const dfo_array *a, int64_t i) {
  return a->type == DFO_FLOAT32 ? (double)((const float *)a->values)[i] : ((const double *)a->values)[i];
}
/* wrap an i128 into the two's complement range of integer type t */
static i128 wrap_to(int t, i128 v) {
  switch (t) {
    case DFO_INT8: return (int8_t)(uint8_t)v; case DFO_INT16: return (int16_t)(uint16_t)v;
    case DFO_INT32: case DFO_DATE32: return (int32_t)(uint32_t)v; case DFO_INT64: return (int64_t)(uint64_t)v;
    case DFO_UINT8: return (uint8_t)v; case DFO_UINT16: return (uint16_t)v; case DFO_UINT32: return (uint32_t)v;
    case DFO_UINT64: return (i128)(uint64_t)v; default: return v;
  }
}
static int fits_int(int t, i128 v) { return wrap_to(t, v) == v; }
static void append_int(dfo_builder *b, i128 v) { dfo_builder_append_value(b, &v); } /* LE low bytes */

int dfo_binary(int op, const dfo_array *l, int ls, const dfo_array *r, int rs, dfo_builder **out) {
  int64_t n = ls ? (rs ? 1 : r->length) : l->length;
  if (!ls && !rs && l->length != r->length) { dfo_set_error("binary: length mismatch"); return 1; }
  int lt = dfo_logical_type(l), rt = dfo_logical_type(r);
  const dfo_array *ll = l->type == DFO_DICTIONARY ? l->dictionary : l, *rl = r->type == DFO_DICTIONARY ? r->dictionary : r;
  if (lt != rt) { dfo_set_error("binary: operand types differ (%d vs %d); the planner coerces first", lt, rt); return 1; }
  int t = lt; dfo_builder *b;

  if (op == DFO_OP_AND || op == DFO_OP_OR) {
    if (t != DFO_BOOL) { dfo_set_error("AND/OR need Boolean"); return 1; }
    b = dfo_builder_new(DFO_BOOL, 0, 0);
    for (int64_t i = 0; i < n; i++) {
      int64_t a = ls ? 0 : i, c = rs ? 0 : i;
      const dfo_array *x = dfo_resolve(l, &a), *y = dfo_resolve(r, &c);
      int xv = x ? dfo_bit((const uint8_t *)x->values, a) : -1, yv = y ? dfo_bit((const uint8_t *)y->values, c) : -1;
      if (op == DFO_OP_AND) { if (xv == 0 || yv == 0) dfo_builder_append_bool(b, 0); else if (xv < 0 || yv < 0) dfo_builder_append_null(b); else dfo_builder_append_bool(b, 1); }
      else { if (xv == 1 || yv == 1) dfo_builder_append_bool(b, 1); else if (xv < 0 || yv < 0) dfo_builder_append_null(b); else dfo_builder_append_bool(b, 0); }
    }
    *out = b; return 0;
  }

  if (op >= DFO_OP_EQ && op <= DFO_OP_NOT_DISTINCT) {
    if (t == DFO_DECIMAL128 && ll->scale != rl->scale) { dfo_set_error("compare: decimal scales differ; the planner coerces first"); return 1; }
    b = dfo_builder_new(DFO_BOOL, 0, 0);
    for (int64_t i = 0; i < n; i++) {
      int64_t a = ls ? 0 : i, c = rs ? 0 : i;
      const dfo_array *x = dfo_resolve(l, &a), *y = dfo_resolve(r, &c);
      if (op == DFO_OP_DISTINCT || op == DFO_OP_NOT_DISTINCT) {
        int same = (!x && !y) || (x && y && dfo_cell_cmp(x, a, y, c) == 0);
        dfo_builder_append_bool(b, op == DFO_OP_NOT_DISTINCT ? same : !same);
        continue;
      }
      if (!x || !y) { dfo_builder_append_null(b); continue; }
      int c3 = dfo_cell_cmp(x, a, y, c), v = 0;
      switch (op) { case DFO_OP_EQ: v = c3 == 0; break; case DFO_OP_NEQ: v = c3 != 0; break; case DFO_OP_LT: v = c3 < 0; break;
        case DFO_OP_LTEQ: v = c3 <= 0; break; case DFO_OP_GT: v = c3 > 0; break; default: v = c3 >= 0; }
      dfo_builder_append_bool(b, v);
    }
    *out = b; return 0;
  }

  if (op > DFO_OP_REM) { dfo_set_error("binary: unknown op %d", op); return 1; }
  if (t == DFO_DECIMAL128) {
    int p1 = ll->precision, s1 = ll->scale, p2 = rl->precision, s2 = rl->scale, rp, rsc; i128 lm = 1, rm = 1;
    switch (op) {
      case DFO_OP_ADD: case DFO_OP_SUB:
        rsc = imax(s1, s2); rp = imin(38, imax(p1 - s1, p2 - s2) + rsc + 1); lm = dfo_pow10(rsc - s1); rm = dfo_pow10(rsc - s2); break;
      case DFO_OP_MUL:
        rsc = s1 + s2; rp = imin(38, p1 + p2 + 1);
        if (rsc > 38) { dfo_set_error("Output scale of decimal multiply would exceed max scale of 38"); return 1; }
        break;
      case DFO_OP_DIV: {
        rsc = imin(38, s1 + 4); int mp = rsc - s1 + s2; rp = imin(38, mp + p1);
        if (mp > 0) lm = dfo_pow10(mp); else if (mp < 0) rm = dfo_pow10(-mp);
        break; }
      default: rsc = imax(s1, s2); rp = imin(p1 - s1, p2 - s2) + rsc; lm = dfo_pow10(rsc - s1); rm = dfo_pow10(rsc - s2); break;
    }
    b = dfo_builder_new(DFO_DECIMAL128, rp, rsc);
    for (int64_t i = 0; i < n; i++) {
      int64_t a = ls ? 0 : i, c = rs ? 0 : i;
      const dfo_array *x = dfo_resolve(l, &a), *y = dfo_resolve(r, &c);
      if (!x || !y) { dfo_builder_append_null(b); continue; }
      i128 xv = get_int(x, a), yv = get_int(y, c), res = 0; int ov = 0;
      switch (op) {
        case DFO_OP_ADD: ov = __builtin_mul_overflow(xv, lm, &xv) | __builtin_mul_overflow(yv, rm, &yv); if (!ov) ov = __builtin_add_overflow(xv, yv, &res); break;
        case DFO_OP_SUB: ov = __builtin_mul_overflow(xv, lm, &xv) | __builtin_mul_overflow(yv, rm, &yv); if (!ov) ov = __builtin_sub_overflow(xv, yv, &res); break;
        case DFO_OP_MUL: ov = __builtin_mul_overflow(xv, yv, &res); break;
        default:
          ov = __builtin_mul_overflow(xv, lm, &xv) | __builtin_mul_overflow(yv, rm, &yv);
          if (!ov) { if (yv == 0) { dfo_builder_free(b); dfo_set_error("Divide by zero error"); return 1; }
            res = op == DFO_OP_DIV ? xv / yv : xv % yv; }
      }
      if (ov) { dfo_builder_free(b); dfo_set_error("Arithmetic overflow: decimal op %d", op); return 1; }
      append_int(b, res);
    }
    *out = b; return 0;
  }
  if (is_flt(t)) {
    b = dfo_builder_new(t, 0, 0);
    for (int64_t i = 0; i < n; i++) {
      int64_t a = ls ? 0 : i, c = rs ? 0 : i;
      const dfo_array *x = dfo_resolve(l, &a), *y = dfo_resolve(r, &c);
      if (!x || !y) { dfo_builder_append_null(b); continue; }
      if (t == DFO_FLOAT64) {
        double xv = get_f(x, a), yv = get_f(y, c), v;
        switch (op) { case DFO_OP_ADD: v = xv + yv; break; case DFO_OP_SUB: v = xv - yv; break; case DFO_OP_MUL: v = xv * yv; break; case DFO_OP_DIV: v = xv / yv; break; default: v = fmod(xv, yv); }
        dfo_builder_append_value(b, &v);
      } else {
        float xv = (float)get_f(x, a), yv = (float)get_f(y, c), v;
        switch (op) { case DFO_OP_ADD: v = xv + yv; break; case DFO_OP_SUB: v = xv - yv; break; case DFO_OP_MUL: v = xv * yv; break; case DFO_OP_DIV: v = xv / yv; break; default: v = fmodf(xv, yv); }
        dfo_builder_append_value(b, &v);
      }
    }
    *out = b; return 0;
  }
  if (is_sint(t) || is_uint(t)) {
    b = dfo_builder_new(t, 0, 0);
    for (int64_t i = 0; i < n; i++) {
      int64_t a = ls ? 0 : i, c = rs ? 0 : i;
      const dfo_array *x = dfo_resolve(l, &a), *y = dfo_resolve(r, &c);
      if (!x || !y) { dfo_builder_append_null(b); continue; }
      i128 xv = get_int(x, a), yv = get_int(y, c), v;
      switch (op) {
        case DFO_OP_ADD: v = wrap_to(t, xv + yv); break; case DFO_OP_SUB: v = wrap_to(t, xv - yv); break;
        case DFO_OP_MUL: v = wrap_to(t, (i128)((u128)xv * (u128)yv)); break;
        default:
          if (yv == 0) { dfo_builder_free(b); dfo_set_error("Divide by zero error"); return 1; }
          if (op == DFO_OP_DIV) { v = xv / yv; if (!fits_int(t, v)) { dfo_builder_free(b); dfo_set_error("Arithmetic overflow: %lld / %lld", (long long)xv, (long long)yv); return 1; } }
          else v = xv % yv;
      }
      append_int(b, v);
    }
    *out = b; return 0;
  }
  dfo_set_error("binary: unsupported type %d for arithmetic", t); return 1;
}

int dfo_not(const dfo_array *a, dfo_builder **out) {
  if (dfo_logical_type(a) != DFO_BOOL) { dfo_set_error("NOT needs Boolean"); return 1; }
  dfo_builder *b = dfo_builder_new(DFO_BOOL, 0, 0);
  for (int64_t i = 0; i < a->length; i++) { int64_t r = i; const dfo_array *x = dfo_resolve(a, &r);
    if (!x) dfo_builder_append_null(b); else dfo_builder_append_bool(b, !dfo_bit((const uint8_t *)x->values, r)); }
  *out = b; return 0;
}
int dfo_is_null(const dfo_array *a, int negate, dfo_builder **out) {
  dfo_builder *b = dfo_builder_new(DFO_BOOL, 0, 0);
  for (int64_t i = 0; i < a->length; i++) { int64_t r = i; int isn = dfo_resolve(a, &r) == NULL; dfo_builder_append_bool(b, negate ? !isn : isn); }
  *out = b; return 0;
}
int dfo_negative(const dfo_array *a, dfo_builder **out) {
  const dfo_array *la = a->type == DFO_DICTIONARY ? a->dictionary : a; int t = la->type;
  dfo_builder *b = dfo_builder_new(t, la->precision, la->scale);
  for (int64_t i = 0; i < a->length; i++) {
    int64_t r = i; const dfo_array *x = dfo_resolve(a, &r);
    if (!x) { dfo_builder_append_null(b); continue; }
    if (t == DFO_FLOAT64) { double v = -get_f(x, r); dfo_builder_append_value(b, &v); }
    else if (t == DFO_FLOAT32) { float v = -(float)get_f(x, r); dfo_builder_append_value(b, &v); }
    else if (is_sint(t) || t == DFO_DECIMAL128) append_int(b, wrap_to(t, (i128)(0 - (u128)get_int(x, r))));
    else { dfo_builder_free(b); dfo_set_error("negative: unsupported type %d", t); return 1; }
  }
  *out = b; return 0;
}

/* arrow-cast 50 semantics with CastOptions{safe:false} (physical-expr/src/expressions/cast.rs:40-47) */
int dfo_cast(const dfo_array *a, int32_t to, int32_t p, int32_t s, dfo_builder **out) {
  const dfo_array *la = a->type == DFO_DICTIONARY ? a->dictionary : a; int from = la->type;
  dfo_builder *b = dfo_builder_new(to, p, s);
#define FAIL(...) { dfo_builder_free(b); dfo_set_error(__VA_ARGS__); return 1; }
  for (int64_t i = 0; i < a->length; i++) {
    int64_t r = i; const dfo_array *x = dfo_resolve(a, &r);
    if (!x) { dfo_builder_append_null(b); continue; }
    if (from == to && from != DFO_DECIMAL128) { dfo_builder_append_cell(b, x, r); continue; }
    if (is_sint(from) || is_uint(from) || from == DFO_DATE32 || from == DFO_BOOL) {
      i128 v = get_int(x, r);
      if (is_sint(to) || is_uint(to) || to == DFO_DATE32) { if (!fits_int(to, v)) FAIL("Cast error: Can't cast value %lld to type %d", (long long)v, to); append_int(b, v); }
      else if (to == DFO_FLOAT64) { double f = (double)v; dfo_builder_append_value(b, &f); }
      else if (to == DFO_FLOAT32) { float f = (float)v; dfo_builder_append_value(b, &f); }
      else if (to == DFO_DECIMAL128) { i128 m; if (__builtin_mul_overflow(v, dfo_pow10(s), &m) || !dfo_decimal_fits(m, p)) FAIL("Invalid argument error: %lld is too large to store in a Decimal128 of precision %d", (long long)v, p); append_int(b, m); }
      else if (to == DFO_BOOL) dfo_builder_append_bool(b, v != 0);
      else FAIL("cast %d -> %d unsupported", from, to);
    } else if (is_flt(from)) {
      double f = get_f(x, r);
      if (to == DFO_FLOAT64) dfo_builder_append_value(b, &f);
      else if (to == DFO_FLOAT32) { float g = (float)f; dfo_builder_append_value(b, &g); }
      else if (is_sint(to) || is_uint(to)) {
        /* num::cast::NumCast semantics: truncate toward zero, error if out of range / NaN */
        if (isnan(f) || f <= -1.7014118346046923e38 || f >= 1.7014118346046923e38) FAIL("Cast error: Can't cast value %g to type %d", f, to);
        i128 v = (i128)f; if (!fits_int(to, v)) FAIL("Cast error: Can't cast value %g to type %d", f, to);
        append_int(b, v);
      } else if (to == DFO_DECIMAL128) {
        double m = round(f * (double)dfo_pow10(s)); i128 v = (i128)m;
        if (isnan(m) || fabs(m) >= 1.7e38 || !dfo_decimal_fits(v, p)) FAIL("Cast error: Cannot cast to Decimal128(%d, %d)", p, s);
        append_int(b, v);
      } else FAIL("cast %d -> %d unsupported", from, to);
    } else if (from == DFO_DECIMAL128) {
      i128 v = get_int(x, r); int fs = la->scale;
      if (to == DFO_DECIMAL128) {
        i128 o;
        if (s >= fs) { if (__builtin_mul_overflow(v, dfo_pow10(s - fs), &o)) FAIL("Cast error: decimal overflow"); }
        else { i128 div = dfo_pow10(fs - s), half = div / 2, d = v / div, rem = v % div; /* round half away from zero */
          if (v >= 0 && rem >= half) d += 1; else if (v < 0 && rem <= -half) d -= 1; o = d; }
        if (!dfo_decimal_fits(o, p)) FAIL("Invalid argument error: value is too large to store in a Decimal128 of precision %d", p);
        append_int(b, o);
      } else if (to == DFO_FLOAT64) { double f = (double)v / (double)dfo_pow10(fs); dfo_builder_append_value(b, &f); }
      else if (to == DFO_FLOAT32) { float f = (float)((double)v / (double)dfo_pow10(fs)); dfo_builder_append_value(b, &f); }
      else if (is_sint(to) || is_uint(to)) { i128 d = v / dfo_pow10(fs); if (!fits_int(to, d)) FAIL("Cast error: value out of range"); append_int(b, d); }
      else FAIL("cast %d -> %d unsupported", from, to);
    } else FAIL("cast %d -> %d unsupported", from, to);
  }
#undef FAIL
  *out = b; return 0;
}

/* InListExpr::evaluate (in_list.rs:349): x IN (list) = true if any non-null element equals x; else NULL if x is NULL or the
 * list holds a NULL; else false.  NOT IN negates non-null results. */
int dfo_in_list(const dfo_array *a, const dfo_array *list, int negated, dfo_builder **out) {
  if (dfo_logical_type(a) != dfo_logical_type(list)) { dfo_set_error("in_list: type mismatch"); return 1; }
  dfo_builder *b = dfo_builder_new(DFO_BOOL, 0, 0);
  int list_has_null = 0;
  for (int64_t j = 0; j < list->length; j++) { int64_t r = j; if (!dfo_resolve(list, &r)) list_has_null = 1; }
  for (int64_t i = 0; i < a->length; i++) {
    int64_t r = i; const dfo_array *x = dfo_resolve(a, &r);
    if (!x) { dfo_builder_append_null(b); continue; }
    int found = 0;
    for (int64_t j = 0; j < list->length && !found; j++) { int64_t q = j; const dfo_array *y = dfo_resolve(list, &q); if (y && dfo_cell_equal(x, r, y, q)) found = 1; }
    if (found) dfo_builder_append_bool(b, !negated);
    else if (list_has_null) dfo_builder_append_null(b);
    else dfo_builder_append_bool(b, negated);
  }
  *out = b; return 0;
}
