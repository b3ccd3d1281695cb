// expr.hip -- a12: the vectorised kernels behind PhysicalExpr::evaluate
// (physical-expr/src/expressions/binary.rs:259-315 -> datum.rs:28-58; and_kleene/or_kleene binary.rs:563-586;
//  cast.rs:121 with CastOptions{safe:false}; not.rs:71; is_null.rs:74; negative.rs:79; in_list.rs:349).
// One lane per row, coalesced loads; a scalar Datum operand is a length-1 column read at index 0 (one
// broadcast transaction per wave).  Boolean results are bit-packed with one 64-lane ballot per wave.
// Semantics of arrow-arith / arrow-ord / arrow-cast 50.0.0 (third-party, pinned in datafusion-cli/Cargo.lock): integer
// add/sub/mul wrap, div/rem are checked; Decimal128 result (precision, scale) rules and checked i128 payloads as listed in
// DESIGN.md section 4; floats compare by IEEE totalOrder; NULL op x = NULL except IS [NOT] DISTINCT FROM.
#include "int128.h"

namespace dfgpu {

struct Operand { ColView v; int32_t scalar; };
__device__ inline bool op_resolve(const Operand& o, int64_t i, int64_t* r) { return cell_resolve(o.v, o.scalar ? 0 : i, r); }

__device__ inline i128 cell_int(const ColView& c, int64_t r) {
  switch (c.type) {
    case DFGPU_BOOL: return bit_get((const uint64_t*)c.values, r);
    case DFGPU_INT8: return ((const int8_t*)c.values)[r];
    case DFGPU_INT16: return ((const int16_t*)c.values)[r];
    case DFGPU_INT32: case DFGPU_DATE32: return ((const int32_t*)c.values)[r];
    case DFGPU_INT64: return ((const int64_t*)c.values)[r];
    case DFGPU_UINT8: return ((const uint8_t*)c.values)[r];
    case DFGPU_UINT16: return ((const uint16_t*)c.values)[r];
    case DFGPU_UINT32: return ((const uint32_t*)c.values)[r];
    case DFGPU_UINT64: return (i128)((const uint64_t*)c.values)[r];
    case DFGPU_DECIMAL128: return load_i128(c.values, r);
    default: return 0;
  }
}
__device__ inline double cell_f64(const ColView& c, int64_t r) { return c.type == DFGPU_FLOAT32 ? (double)((const float*)c.values)[r] : ((const double*)c.values)[r]; }
__device__ inline void store_int(void* out, int32_t type, int64_t i, i128 v) {
  switch (type) {
    case DFGPU_INT8: case DFGPU_UINT8: ((uint8_t*)out)[i] = (uint8_t)v; break;
    case DFGPU_INT16: case DFGPU_UINT16: ((uint16_t*)out)[i] = (uint16_t)v; break;
    case DFGPU_INT32: case DFGPU_UINT32: case DFGPU_DATE32: ((uint32_t*)out)[i] = (uint32_t)v; break;
    case DFGPU_INT64: case DFGPU_UINT64: ((uint64_t*)out)[i] = (uint64_t)v; break;
    default: store_i128(out, i, v);
  }
}
__device__ inline i128 wrap_to(int32_t t, i128 v) {
  switch (t) {
    case DFGPU_INT8: return (int8_t)(uint8_t)v; case DFGPU_INT16: return (int16_t)(uint16_t)v;
    case DFGPU_INT32: case DFGPU_DATE32: return (int32_t)(uint32_t)v; case DFGPU_INT64: return (int64_t)(uint64_t)v;
    case DFGPU_UINT8: return (uint8_t)v; case DFGPU_UINT16: return (uint16_t)v; case DFGPU_UINT32: return (uint32_t)v;
    case DFGPU_UINT64: return (i128)(uint64_t)v; default: return v;
  }
}
// three-way compare of resolved non-null cells; floats by IEEE totalOrder (arrow-ord cmp)
__device__ inline int cell_cmp(const ColView& a, int64_t i, const ColView& b, int64_t j) {
  switch (a.type) {
    case DFGPU_FLOAT32: { int32_t x = ((const int32_t*)a.values)[i], y = ((const int32_t*)b.values)[j]; x ^= (int32_t)((uint32_t)(x >> 31) >> 1); y ^= (int32_t)((uint32_t)(y >> 31) >> 1); return x < y ? -1 : (x > y ? 1 : 0); }
    case DFGPU_FLOAT64: { int64_t x = ((const int64_t*)a.values)[i], y = ((const int64_t*)b.values)[j]; x ^= (int64_t)((uint64_t)(x >> 63) >> 1); y ^= (int64_t)((uint64_t)(y >> 63) >> 1); return x < y ? -1 : (x > y ? 1 : 0); }
    case DFGPU_UTF8: {
      int32_t oa = a.offsets[i], ob = b.offsets[j], la = a.offsets[i + 1] - oa, lb = b.offsets[j + 1] - ob, m = la < lb ? la : lb;
      const uint8_t* p = (const uint8_t*)a.values + oa; const uint8_t* q = (const uint8_t*)b.values + ob;
      for (int32_t k = 0; k < m; k++) if (p[k] != q[k]) return p[k] < q[k] ? -1 : 1;
      return la < lb ? -1 : (la > lb ? 1 : 0);
    }
    default: { i128 x = cell_int(a, i), y = cell_int(b, j); return x < y ? -1 : (x > y ? 1 : 0); }
  }
}

// ---------------------------------------------------------------- comparisons -> bit-packed Boolean
__global__ void __launch_bounds__(BLOCK) k_compare(int op, Operand l, Operand r, int64_t n, uint64_t* out_bits, uint64_t* out_valid) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  bool v = false, ok = false;
  if (i < n) {
    int64_t a, b; bool va = op_resolve(l, i, &a), vb = op_resolve(r, i, &b);
    if (op == DFGPU_OP_DISTINCT || op == DFGPU_OP_NOT_DISTINCT) {
      bool same = (!va && !vb) || (va && vb && cell_cmp(l.v, a, r.v, b) == 0);
      v = op == DFGPU_OP_NOT_DISTINCT ? same : !same; ok = true;
    } else if (va && vb) {
      int c = cell_cmp(l.v, a, r.v, b); ok = true;
      switch (op) { case DFGPU_OP_EQ: v = c == 0; break; case DFGPU_OP_NEQ: v = c != 0; break; case DFGPU_OP_LT: v = c < 0; break;
        case DFGPU_OP_LTEQ: v = c <= 0; break; case DFGPU_OP_GT: v = c > 0; break; default: v = c >= 0; }
    }
  }
  uint64_t mv = ballot64(v), mo = ballot64(ok);
  if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) { out_bits[i >> 6] = mv; if (out_valid) out_valid[i >> 6] = mo; }
}

// Fast path of the FilterExec predicates that dominate TPC-H (`l_shipdate > date`, `o_orderdate < date`, `key = c`):
// non-null fixed-width integer column vs scalar.  64 bytes per lane in flight, one 64-bit ballot per 64 rows, one chunk
// per wave (1:1 grid).  Operator and "whole chunk in range" are compile-time: a bounds-checked load compiles to a branch
// plus s_waitcnt vmcnt(0) per load, i.e. no memory parallelism (profiles/experiments/compare_stream_microbench.hip:
// 0.56 -> 0.40 ms per 600M Date32 rows = 6.2 TB/s).
template <typename T> constexpr int cmp_rows() { return 64 / (int)sizeof(T); }
template <typename T, int OP, bool FULL>
__device__ inline void compare_scalar_chunk(const T* v, T s, int64_t n, uint64_t* out_bits, int64_t base, int lane) {
  constexpr int ROWS = cmp_rows<T>();
  T x[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; r++) { int64_t j = base + r * WAVE + lane; x[r] = (FULL || j < n) ? v[j] : s; }
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    int64_t j = base + r * WAVE + lane; bool b;
    switch (OP) { case DFGPU_OP_EQ: b = x[r] == s; break; case DFGPU_OP_NEQ: b = x[r] != s; break; case DFGPU_OP_LT: b = x[r] < s; break;
      case DFGPU_OP_LTEQ: b = x[r] <= s; break; case DFGPU_OP_GT: b = x[r] > s; break; default: b = x[r] >= s; }
    uint64_t m = ballot64(b && (FULL || j < n));
    if (lane == 0 && (FULL || base + r * WAVE < n)) out_bits[(base >> 6) + r] = m;
  }
}
template <typename T, int OP>
__global__ void __launch_bounds__(BLOCK) k_compare_scalar_fast(const T* v, T s, int64_t n, uint64_t* out_bits) {
  int lane = lane_id();
  int64_t base = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * (WAVE * cmp_rows<T>());
  if (base + WAVE * cmp_rows<T>() <= n) compare_scalar_chunk<T, OP, true>(v, s, n, out_bits, base, lane);
  else if (base < n) compare_scalar_chunk<T, OP, false>(v, s, n, out_bits, base, lane);
}
template <typename T>
static void launch_compare_scalar_fast(dfgpu_ctx* ctx, int op, const T* v, T s, int64_t n, uint64_t* out_bits) {
  dim3 g(grid_for(n, BLOCK * cmp_rows<T>())), block(BLOCK);
  switch (op) {
#define CMP_CASE(OP) case OP: hipLaunchKernelGGL((k_compare_scalar_fast<T, OP>), g, block, 0, ctx->stream, v, s, n, out_bits); break;
    CMP_CASE(DFGPU_OP_EQ) CMP_CASE(DFGPU_OP_NEQ) CMP_CASE(DFGPU_OP_LT) CMP_CASE(DFGPU_OP_LTEQ) CMP_CASE(DFGPU_OP_GT) CMP_CASE(DFGPU_OP_GTEQ)
#undef CMP_CASE
    default: fail(DFGPU_INTERNAL, "compare fast path: operator %d", op);
  }
}
// dictionary column vs scalar: the predicate is evaluated once per dictionary entry (dict_bits / dict_valid), rows map their code
__global__ void __launch_bounds__(BLOCK) k_dict_predicate(const void* keys, int key_type, const uint64_t* key_valid, int64_t n, const uint64_t* dict_bits, const uint64_t* dict_valid,
                                                          int64_t dict_len, uint64_t* out_bits, uint64_t* out_valid) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  bool v = false, ok = false;
  if (i < n && valid_at(key_valid, i)) {
    int64_t c = key_at(keys, key_type, i);
    if (c >= 0 && c < dict_len) { ok = valid_at(dict_valid, c); v = ok && bit_get(dict_bits, c); }
  }
  uint64_t mv = ballot64(v), mo = ballot64(ok);
  if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) { out_bits[i >> 6] = mv; if (out_valid) out_valid[i >> 6] = mo; }
}
// the same for Int32 codes without NULLs on either side.  One row per lane with the dictionary bits read from global memory ran at 0.44 ms per 100 M rows over a dictionary of
// 1 M entries, and eight rows per lane did not change it (0.47 ms, profiles/r04_n_timeline_cbu.txt): the time is the 100 M random 8-byte reads of a 125 KB bitmap that no L1
// holds, not the 400 MB of codes.  Here every workgroup first copies the bitmap into LDS (up to 144 KB = 1.18 M entries) and the rows read their bit from there.
template <typename KT>
__global__ void __launch_bounds__(1024) k_dict_predicate_i32(const KT* __restrict__ keys, int64_t n, const uint64_t* __restrict__ dict_bits, int64_t dict_len, uint64_t* __restrict__ out_bits) {
  extern __shared__ uint32_t dp_bits[];
  const int64_t dwords = (dict_len + 63) >> 6;
  for (int64_t x = threadIdx.x; x < dwords; x += 1024) { const uint64_t v = dict_bits[x]; dp_bits[2 * x] = (uint32_t)v; dp_bits[2 * x + 1] = (uint32_t)(v >> 32); }
  __syncthreads();
  const int lane = lane_id();
  const int64_t nchunks = (n + 511) >> 9, wave = ((int64_t)blockIdx.x * 1024 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 1024) >> 6, nw = (n + 63) >> 6;
  for (int64_t ch = wave; ch < nchunks; ch += nwaves) {
    const int64_t base = ch << 9;
    int32_t c[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { const int64_t i = base + q * 64 + lane; c[q] = i < n ? (int32_t)keys[i] : -1; }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const bool v = c[q] >= 0 && c[q] < dict_len && ((dp_bits[c[q] >> 5] >> (c[q] & 31)) & 1u);
      const uint64_t m = ballot64(v); if (lane == 0 && (base >> 6) + q < nw) out_bits[(base >> 6) + q] = m;
    }
  }
}
static int swap_cmp(int op) { switch (op) { case DFGPU_OP_LT: return DFGPU_OP_GT; case DFGPU_OP_LTEQ: return DFGPU_OP_GTEQ; case DFGPU_OP_GT: return DFGPU_OP_LT; case DFGPU_OP_GTEQ: return DFGPU_OP_LTEQ; default: return op; } }

// ---------------------------------------------------------------- Kleene AND / OR on bitmap words
__global__ void k_kleene(int is_and, const uint64_t* lv, const uint64_t* lok, int lscalar, const uint64_t* rv, const uint64_t* rok, int rscalar,
                         int64_t nw, uint64_t* out, uint64_t* out_ok) {
  int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nw) return;
  uint64_t a = lscalar ? ((lv[0] & 1) ? ~0ull : 0ull) : lv[w], ao = lok ? (lscalar ? ((lok[0] & 1) ? ~0ull : 0ull) : lok[w]) : ~0ull;
  uint64_t b = rscalar ? ((rv[0] & 1) ? ~0ull : 0ull) : rv[w], bo = rok ? (rscalar ? ((rok[0] & 1) ? ~0ull : 0ull) : rok[w]) : ~0ull;
  a &= ao; b &= bo;                      // value bits only where valid
  if (is_and) { out[w] = a & b; if (out_ok) out_ok[w] = (ao & bo) | (ao & ~a) | (bo & ~b); }
  else { out[w] = a | b; if (out_ok) out_ok[w] = (ao & bo) | a | b; }
}
__global__ void k_not_words(const uint64_t* in, uint64_t* out, int64_t nw) {
  int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w < nw) out[w] = ~in[w];
}
__global__ void __launch_bounds__(BLOCK) k_is_null(ColView c, int64_t n, int negate, uint64_t* out) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  bool v = false;
  if (i < n) { int64_t r; bool isn = !cell_resolve(c, i, &r); v = negate ? !isn : isn; }
  uint64_t m = ballot64(v);
  if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) out[i >> 6] = m;
}
// validity of a row-wise result: both operands non-null
__global__ void __launch_bounds__(BLOCK) k_both_valid(Operand l, Operand r, int64_t n, uint64_t* out) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  bool ok = false;
  if (i < n) { int64_t a, b; ok = op_resolve(l, i, &a) && op_resolve(r, i, &b); }
  uint64_t m = ballot64(ok);
  if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) out[i >> 6] = m;
}

// ---------------------------------------------------------------- arithmetic
struct DecRule { i128 lmul, rmul; };
__device__ inline void raise(uint32_t* flags, uint32_t f) { if (flags) atomicOr(flags, f); }
// emask: row selection of the caller (dfgpu_ctx_set_row_selection): a row it drops cannot raise
__global__ void __launch_bounds__(BLOCK) k_arith(int op, Operand l, Operand r, int64_t n, int32_t out_type, DecRule dr, void* out, uint32_t* flags_, const uint64_t* emask) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t* flags = (emask == nullptr || bit_get(emask, i)) ? flags_ : nullptr;
  int64_t a, b;
  if (!op_resolve(l, i, &a) || !op_resolve(r, i, &b)) {     // NULL in -> NULL out (validity written by k_both_valid); keep bytes defined
    if (out_type == DFGPU_FLOAT64) ((double*)out)[i] = 0; else if (out_type == DFGPU_FLOAT32) ((float*)out)[i] = 0; else store_int(out, out_type, i, 0);
    return;
  }
  if (out_type == DFGPU_FLOAT64) {
    double x = cell_f64(l.v, a), y = cell_f64(r.v, b), v;
    switch (op) { case DFGPU_OP_ADD: v = x + y; break; case DFGPU_OP_SUB: v = x - y; break; case DFGPU_OP_MUL: v = x * y; break; case DFGPU_OP_DIV: v = x / y; break; default: v = fmod(x, y); }
    ((double*)out)[i] = v; return;
  }
  if (out_type == DFGPU_FLOAT32) {
    float x = (float)cell_f64(l.v, a), y = (float)cell_f64(r.v, b), v;
    switch (op) { case DFGPU_OP_ADD: v = x + y; break; case DFGPU_OP_SUB: v = x - y; break; case DFGPU_OP_MUL: v = x * y; break; case DFGPU_OP_DIV: v = x / y; break; default: v = fmodf(x, y); }
    ((float*)out)[i] = v; return;
  }
  i128 x = cell_int(l.v, a), y = cell_int(r.v, b), v = 0;
  if (out_type == DFGPU_DECIMAL128) {                       // checked i128 arithmetic (arrow-arith decimal_op)
    bool ok = true;
    if (op != DFGPU_OP_MUL) ok = mul128_checked(x, dr.lmul, &x) && mul128_checked(y, dr.rmul, &y);
    if (ok) switch (op) {
      case DFGPU_OP_ADD: ok = add128_checked(x, y, &v); break;
      case DFGPU_OP_SUB: ok = sub128_checked(x, y, &v); break;
      case DFGPU_OP_MUL: ok = mul128_checked(x, y, &v); break;
      default:
        if (y == 0) { raise(flags, DFGPU_FLAG_DIV_ZERO); v = 0; }
        else { i128 rem; i128 q = sdiv128(x, y, &rem); v = op == DFGPU_OP_DIV ? q : rem; }
    }
    if (!ok) { raise(flags, DFGPU_FLAG_OVERFLOW); v = 0; }
    store_i128(out, i, v); return;
  }
  switch (op) {                                             // integer: *_wrapping kernels, checked div/rem
    case DFGPU_OP_ADD: v = wrap_to(out_type, x + y); break;
    case DFGPU_OP_SUB: v = wrap_to(out_type, x - y); break;
    case DFGPU_OP_MUL: v = wrap_to(out_type, (i128)((u128)x * (u128)y)); break;
    default:
      if (y == 0) { raise(flags, DFGPU_FLAG_DIV_ZERO); v = 0; }
      else { i128 rem; i128 q = sdiv128(x, y, &rem);
        if (op == DFGPU_OP_DIV) { v = q; if (wrap_to(out_type, q) != q) { raise(flags, DFGPU_FLAG_OVERFLOW); v = 0; } } else v = rem; }
  }
  store_int(out, out_type, i, v);
}

__global__ void __launch_bounds__(BLOCK) k_negative(ColView c, int64_t n, void* out) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  int64_t r; bool ok = cell_resolve(c, i, &r);
  if (c.type == DFGPU_FLOAT64) ((double*)out)[i] = ok ? -cell_f64(c, r) : 0.0;
  else if (c.type == DFGPU_FLOAT32) ((float*)out)[i] = ok ? -(float)cell_f64(c, r) : 0.0f;
  else store_int(out, c.type, i, ok ? (i128)((u128)0 - (u128)cell_int(c, r)) : 0);
}

// ---------------------------------------------------------------- cast (safe = false)
// ---- x op1 (s op2 y) in one pass (s scalar): TPC-H's `l_extendedprice * (1 - l_discount)`, `disc_price * (1 + l_tax)`.  The inner
// result never reaches HBM (two kernels move 80 B per Decimal128 row, this one 48).  Same checked arithmetic, same result type.
__device__ inline bool dec_core(int op, i128 x, i128 y, DecRule dr, i128* v, uint32_t* flags) {
  bool ok = true; *v = 0;
  if (op != DFGPU_OP_MUL) ok = mul128_checked(x, dr.lmul, &x) && mul128_checked(y, dr.rmul, &y);
  if (ok) switch (op) {
    case DFGPU_OP_ADD: ok = add128_checked(x, y, v); break;
    case DFGPU_OP_SUB: ok = sub128_checked(x, y, v); break;
    case DFGPU_OP_MUL: ok = mul128_checked(x, y, v); break;
    default:
      if (y == 0) { raise(flags, DFGPU_FLAG_DIV_ZERO); *v = 0; }
      else { i128 rem; i128 q = sdiv128(x, y, &rem); *v = op == DFGPU_OP_DIV ? q : rem; }
  }
  if (!ok) { raise(flags, DFGPU_FLAG_OVERFLOW); *v = 0; }
  return ok;
}
__device__ inline double f64_core(int op, double x, double y) {
  switch (op) { case DFGPU_OP_ADD: return x + y; case DFGPU_OP_SUB: return x - y; case DFGPU_OP_MUL: return x * y; case DFGPU_OP_DIV: return x / y; default: return fmod(x, y); }
}
struct Fused2 { int op_in, op_out; int s_left, inner_left; DecRule dr_in, dr_out; i128 s_i; double s_f; };
template <bool DEC>
__global__ void __launch_bounds__(BLOCK) k_arith_fused2(Fused2 f, const void* xv, const void* yv, int64_t n, void* out, uint32_t* flags_, const uint64_t* emask) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t* flags = (emask == nullptr || bit_get(emask, i)) ? flags_ : nullptr;
  if constexpr (DEC) {
    i128 x = load_i128(xv, i), y = load_i128(yv, i), t, v;
    dec_core(f.op_in, f.s_left ? f.s_i : y, f.s_left ? y : f.s_i, f.dr_in, &t, flags);
    dec_core(f.op_out, f.inner_left ? t : x, f.inner_left ? x : t, f.dr_out, &v, flags);
    store_i128(out, i, v);
  } else {
    double x = ((const double*)xv)[i], y = ((const double*)yv)[i];
    double t = f64_core(f.op_in, f.s_left ? f.s_f : y, f.s_left ? y : f.s_f);
    ((double*)out)[i] = f64_core(f.op_out, f.inner_left ? t : x, f.inner_left ? x : t);
  }
}

__global__ void __launch_bounds__(BLOCK) k_cast(ColView c, int64_t n, int32_t to, int32_t p, int32_t s, void* out, uint32_t* flags, const uint64_t* emask) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  bool bit = false;
  if (i < n) {
    int64_t r; bool ok = cell_resolve(c, i, &r);
    int from = c.type; bool err = false;
    bool from_int = from == DFGPU_BOOL || from == DFGPU_DATE32 || (from >= DFGPU_INT8 && from <= DFGPU_UINT64);
    bool to_int = to == DFGPU_DATE32 || (to >= DFGPU_INT8 && to <= DFGPU_UINT64);
    i128 iv = 0; double fv = 0;
    if (ok) {
      if (from_int) {
        i128 v = cell_int(c, r);
        if (to_int) { iv = v; err = wrap_to(to, v) != v; }
        else if (to == DFGPU_FLOAT64 || to == DFGPU_FLOAT32) fv = (double)(int64_t)v + (from == DFGPU_UINT64 && v > (i128)INT64_MAX ? 18446744073709551616.0 : 0.0);
        else if (to == DFGPU_DECIMAL128) { err = !mul128_checked(v, pow10_i128(s), &iv) || !decimal_fits(iv, p); }
        else if (to == DFGPU_BOOL) bit = v != 0;
      } else if (from == DFGPU_FLOAT32 || from == DFGPU_FLOAT64) {
        double f = cell_f64(c, r);
        if (to == DFGPU_FLOAT64 || to == DFGPU_FLOAT32) fv = f;
        else if (to_int) {
          if (f != f || f <= -9.3e18 || f >= 1.85e19) err = true;
          else { iv = f < 0 ? (i128)(int64_t)f : (i128)(uint64_t)f; err = wrap_to(to, iv) != iv; }
        } else if (to == DFGPU_DECIMAL128) {
          double m = round(f * pow10_f64(s));
          if (m != m || fabs(m) >= 1.7e38) err = true;
          else { bool neg = m < 0; double am = fabs(m); uint64_t hi = (uint64_t)(am / 18446744073709551616.0); double lo = am - (double)hi * 18446744073709551616.0;
                 u128 u = ((u128)hi << 64) + (u128)(uint64_t)lo; iv = neg ? (i128)((u128)0 - u) : (i128)u; err = !decimal_fits(iv, p); }
        }
      } else if (from == DFGPU_DECIMAL128) {
        i128 v = cell_int(c, r); int fs = c.scale;
        if (to == DFGPU_DECIMAL128) {
          if (s >= fs) err = !mul128_checked(v, pow10_i128(s - fs), &iv);
          else { i128 div = pow10_i128(fs - s), rem; i128 d = sdiv128(v, div, &rem); i128 half = sdiv128(div, 2, nullptr);   // round half away from zero
                 if (v >= 0 && rem >= half) d += 1; else if (v < 0 && rem <= -half) d -= 1; iv = d; }
          if (!err) err = !decimal_fits(iv, p);
        } else if (to == DFGPU_FLOAT64 || to == DFGPU_FLOAT32) {
          bool neg = v < 0; u128 u = neg ? (u128)0 - (u128)v : (u128)v;
          double d = (double)(uint64_t)(u >> 64) * 18446744073709551616.0 + (double)(uint64_t)u; fv = (neg ? -d : d) / pow10_f64(fs);
        } else if (to_int) { iv = sdiv128(v, pow10_i128(fs), nullptr); err = wrap_to(to, iv) != iv; }
      }
      if (err) { if (emask == nullptr || bit_get(emask, i)) atomicOr(flags, DFGPU_FLAG_CAST); iv = 0; fv = 0; }
    }
    if (to == DFGPU_FLOAT64) ((double*)out)[i] = fv;
    else if (to == DFGPU_FLOAT32) ((float*)out)[i] = (float)fv;
    else if (to != DFGPU_BOOL) store_int(out, to, i, iv);
  }
  if (to == DFGPU_BOOL) { uint64_t m = ballot64(bit); if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) ((uint64_t*)out)[i >> 6] = m; }
}

__global__ void __launch_bounds__(BLOCK) k_in_list(ColView c, ColView list, int64_t list_len, int64_t n, int negated, uint64_t* out_bits, uint64_t* out_valid) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  bool v = false, ok = false;
  if (i < n) {
    int64_t r;
    if (cell_resolve(c, i, &r)) {
      bool found = false, has_null = false;
      for (int64_t j = 0; j < list_len && !found; j++) { int64_t q; if (!cell_resolve(list, j, &q)) has_null = true; else if (cell_equal(c, r, list, q)) found = true; }
      if (!found) for (int64_t j = 0; j < list_len; j++) { int64_t q; if (!cell_resolve(list, j, &q)) has_null = true; }
      if (found) { v = !negated; ok = true; } else if (!has_null) { v = negated != 0; ok = true; }
    }
  }
  uint64_t mv = ballot64(v), mo = ballot64(ok);
  if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) { out_bits[i >> 6] = mv; out_valid[i >> 6] = mo; }
}

static bool may_have_nulls(const dfgpu_array* a) { return a->validity != nullptr || (a->dictionary && a->dictionary->validity != nullptr); }
static Operand make_operand(const dfgpu_array* a, int scalar) {
  if (scalar && a->length != 1) fail(DFGPU_INVALID_ARGUMENT, "scalar operand must have length 1");
  return Operand{ make_view(a), scalar ? 1 : 0 };
}

}  // namespace dfgpu

using namespace dfgpu;
extern "C" {

dfgpu_status dfgpu_binary(dfgpu_ctx* ctx, int32_t op, const dfgpu_array* l, int32_t ls, const dfgpu_array* r, int32_t rs, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!l || !r || !out) fail(DFGPU_INVALID_ARGUMENT, "binary: null argument");
    int64_t n = ls ? (rs ? 1 : r->length) : l->length;
    if (!ls && !rs && l->length != r->length) fail(DFGPU_INVALID_ARGUMENT, "binary: operand lengths differ (%lld vs %lld)", (long long)l->length, (long long)r->length);
    int32_t lt = logical_type(l), rt = logical_type(r);
    if (lt != rt) fail(DFGPU_INVALID_ARGUMENT, "binary: operand types differ (%d vs %d); the planner coerces first", lt, rt);
    Operand lo = make_operand(l, ls), ro = make_operand(r, rs);
    bool nulls = may_have_nulls(l) || may_have_nulls(r);
    dim3 grid(grid_for(n, BLOCK)), block(BLOCK);
    int64_t nw = (n + 63) / 64;
    if (op == DFGPU_OP_AND || op == DFGPU_OP_OR) {
      if (lt != DFGPU_BOOL) fail(DFGPU_INVALID_ARGUMENT, "AND/OR need Boolean operands");
      if (l->type == DFGPU_DICTIONARY || r->type == DFGPU_DICTIONARY) fail(DFGPU_NOT_IMPLEMENTED, "AND/OR on dictionary arrays");
      ArrayHolder h(new_fixed(ctx, DFGPU_BOOL, n, 0, 0, nulls));
      if (nw) hipLaunchKernelGGL(k_kleene, dim3(grid_for(nw, BLOCK)), block, 0, ctx->stream, op == DFGPU_OP_AND ? 1 : 0,
                                 (const uint64_t*)l->values->ptr, l->validity ? (const uint64_t*)l->validity->ptr : nullptr, ls ? 1 : 0,
                                 (const uint64_t*)r->values->ptr, r->validity ? (const uint64_t*)r->validity->ptr : nullptr, rs ? 1 : 0, nw,
                                 (uint64_t*)h.get()->values->ptr, nulls ? (uint64_t*)h.get()->validity->ptr : nullptr);
      KERNEL_CHECK(); if (nulls) h.get()->null_count = -1;
      *out = h.release(); return;
    }
    if (op >= DFGPU_OP_EQ && op <= DFGPU_OP_NOT_DISTINCT) {
      if (lt == DFGPU_DECIMAL128 && lo.v.scale != ro.v.scale) fail(DFGPU_INVALID_ARGUMENT, "compare: decimal scales differ; the planner coerces first");
      bool need_valid = nulls && op != DFGPU_OP_DISTINCT && op != DFGPU_OP_NOT_DISTINCT;
      ArrayHolder h(new_fixed(ctx, DFGPU_BOOL, n, 0, 0, need_valid));
      {   // dictionary column vs scalar (`c_mktsegment = 'BUILDING'`): compare the dictionary values, then map the codes
        const dfgpu_array* dcol = rs && !ls && l->type == DFGPU_DICTIONARY && r->type != DFGPU_DICTIONARY ? l : (ls && !rs && r->type == DFGPU_DICTIONARY && l->type != DFGPU_DICTIONARY ? r : nullptr);
        if (dcol && n && op <= DFGPU_OP_GTEQ && dcol->dictionary && dcol->dictionary->length < n) {
          dfgpu_array* dres = nullptr;
          dfgpu_status st = dcol == l ? dfgpu_binary(ctx, op, dcol->dictionary, 0, r, 1, &dres) : dfgpu_binary(ctx, op, l, 1, dcol->dictionary, 0, &dres);
          if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str());
          ArrayHolder dh(dres);
          bool nv = dcol->validity || dres->validity;
          ArrayHolder hd(new_fixed(ctx, DFGPU_BOOL, n, 0, 0, nv));
          KernelTimer kt_(ctx, "k_dict_predicate");
          const size_t dp_lds = (size_t)((dres->length + 63) / 64) * 8;
          const int32_t kt = dcol->key_type;
          if (!nv && (kt == DFGPU_INT32 || kt == DFGPU_INT16 || kt == DFGPU_INT8 || kt == DFGPU_UINT8 || kt == DFGPU_UINT16) && dp_lds <= 144 * 1024 && n >= (1 << 16)) {
            const int per_cu = dp_lds <= 32 * 1024 ? 2 : 1;       // a large bitmap leaves room for one workgroup of 16 waves per CU
#define DP_LAUNCH(KT) do { HIP_CHECK(hipFuncSetAttribute((const void*)k_dict_predicate_i32<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)); \
            hipLaunchKernelGGL((k_dict_predicate_i32<KT>), dim3(grid_for(n, 1024 * 8, ctx->num_cus * per_cu)), dim3(1024), dp_lds, ctx->stream, (const KT*)dcol->values->ptr, n, (const uint64_t*)dres->values->ptr, dres->length, (uint64_t*)hd.get()->values->ptr); } while (0)
            switch (kt) { case DFGPU_INT8: DP_LAUNCH(int8_t); break; case DFGPU_INT16: DP_LAUNCH(int16_t); break; case DFGPU_UINT8: DP_LAUNCH(uint8_t); break; case DFGPU_UINT16: DP_LAUNCH(uint16_t); break; default: DP_LAUNCH(int32_t); break; }
#undef DP_LAUNCH
            KERNEL_CHECK(); *out = hd.release(); return;
          }
          hipLaunchKernelGGL(k_dict_predicate, grid, block, 0, ctx->stream, dcol->values->ptr, dcol->key_type, dcol->validity ? (const uint64_t*)dcol->validity->ptr : nullptr, n,
                             (const uint64_t*)dres->values->ptr, dres->validity ? (const uint64_t*)dres->validity->ptr : nullptr, dres->length,
                             (uint64_t*)hd.get()->values->ptr, nv ? (uint64_t*)hd.get()->validity->ptr : nullptr);
          KERNEL_CHECK(); if (nv) hd.get()->null_count = -1;
          *out = hd.release(); return;
        }
      }
      {   // fast path: non-null fixed-width integer column vs non-null scalar
        const dfgpu_array* col = rs && !ls ? l : (ls && !rs ? r : nullptr); const dfgpu_array* sc = col == l ? r : l;
        int fop = col == l ? op : swap_cmp(op);
        if (col && n && !nulls && op <= DFGPU_OP_GTEQ && col->type != DFGPU_DICTIONARY && sc->type != DFGPU_DICTIONARY && sc->has_host_scalar && sc->host_scalar_valid &&
            (lt == DFGPU_INT32 || lt == DFGPU_DATE32 || lt == DFGPU_INT64)) {
          KernelTimer kt_(ctx, "k_compare_scalar_fast");
          if (lt == DFGPU_INT64) { int64_t sv; memcpy(&sv, sc->host_scalar, 8); launch_compare_scalar_fast<int64_t>(ctx, fop, (const int64_t*)col->values->ptr, sv, n, (uint64_t*)h.get()->values->ptr); }
          else { int32_t sv; memcpy(&sv, sc->host_scalar, 4); launch_compare_scalar_fast<int32_t>(ctx, fop, (const int32_t*)col->values->ptr, sv, n, (uint64_t*)h.get()->values->ptr); }
          KERNEL_CHECK();
          *out = h.release(); return;
        }
      }
      KernelTimer kt_(ctx, "k_compare");
      if (n) hipLaunchKernelGGL(k_compare, grid, block, 0, ctx->stream, op, lo, ro, n, (uint64_t*)h.get()->values->ptr, need_valid ? (uint64_t*)h.get()->validity->ptr : nullptr);
      KERNEL_CHECK(); if (need_valid) h.get()->null_count = -1;
      *out = h.release(); return;
    }
    if (op < DFGPU_OP_ADD || op > DFGPU_OP_REM) fail(DFGPU_INVALID_ARGUMENT, "binary: unknown operator %d", op);
    int32_t ot = lt, rp = 0, rsc = 0; DecRule dr{ 1, 1 };
    if (lt == DFGPU_DECIMAL128) {
      int p1 = lo.v.precision, s1 = lo.v.scale, p2 = ro.v.precision, s2 = ro.v.scale;
      auto mn = [](int a, int b) { return a < b ? a : b; }; auto mx = [](int a, int b) { return a > b ? a : b; };
      switch (op) {
        case DFGPU_OP_ADD: case DFGPU_OP_SUB: rsc = mx(s1, s2); rp = mn(38, mx(p1 - s1, p2 - s2) + rsc + 1); dr.lmul = pow10_i128(rsc - s1); dr.rmul = pow10_i128(rsc - s2); break;
        case DFGPU_OP_MUL: rsc = s1 + s2; rp = mn(38, p1 + p2 + 1); if (rsc > 38) fail(DFGPU_EXECUTION, "Arrow error: Output scale of decimal multiply would exceed max scale of 38"); break;
        case DFGPU_OP_DIV: { rsc = mn(38, s1 + 4); int mp = rsc - s1 + s2; rp = mn(38, mp + p1); if (mp > 0) dr.lmul = pow10_i128(mp); else if (mp < 0) dr.rmul = pow10_i128(-mp); break; }
        default: rsc = mx(s1, s2); rp = mn(p1 - s1, p2 - s2) + rsc; dr.lmul = pow10_i128(rsc - s1); dr.rmul = pow10_i128(rsc - s2); break;
      }
    } else if (!(is_signed_int(lt) || is_unsigned_int(lt) || is_float(lt))) fail(DFGPU_NOT_IMPLEMENTED, "arithmetic on type %d", lt);
    ArrayHolder h(new_fixed(ctx, ot, n, rp, rsc, nulls));
    if (n) {
      KernelTimer kt_(ctx, "k_arith");
      hipLaunchKernelGGL(k_arith, grid, block, 0, ctx->stream, op, lo, ro, n, ot, dr, h.get()->values->ptr, ctx->d_flags, row_selection_words(ctx, n));
      if (nulls) hipLaunchKernelGGL(k_both_valid, grid, block, 0, ctx->stream, lo, ro, n, (uint64_t*)h.get()->validity->ptr);
      KERNEL_CHECK();
    }
    if (nulls) h.get()->null_count = -1;
    if (lt == DFGPU_DECIMAL128 || op == DFGPU_OP_DIV || op == DFGPU_OP_REM) check_flags(ctx, "binary arithmetic");
    *out = h.release();
  });
}

// decimal result type + operand scaling of one arithmetic node (arrow-arith decimal_op; the switch dfgpu_binary uses)
struct ArithPlan { int32_t rp = 0, rsc = 0; DecRule dr{1, 1}; };
static ArithPlan plan_decimal(int op, int p1, int s1, int p2, int s2) {
  ArithPlan a; auto mn = [](int x, int y) { return x < y ? x : y; }; auto mx = [](int x, int y) { return x > y ? x : y; };
  switch (op) {
    case DFGPU_OP_ADD: case DFGPU_OP_SUB: a.rsc = mx(s1, s2); a.rp = mn(38, mx(p1 - s1, p2 - s2) + a.rsc + 1); a.dr.lmul = pow10_i128(a.rsc - s1); a.dr.rmul = pow10_i128(a.rsc - s2); break;
    case DFGPU_OP_MUL: a.rsc = s1 + s2; a.rp = mn(38, p1 + p2 + 1); if (a.rsc > 38) fail(DFGPU_EXECUTION, "Arrow error: Output scale of decimal multiply would exceed max scale of 38"); break;
    case DFGPU_OP_DIV: { a.rsc = mn(38, s1 + 4); int mp = a.rsc - s1 + s2; a.rp = mn(38, mp + p1); if (mp > 0) a.dr.lmul = pow10_i128(mp); else if (mp < 0) a.dr.rmul = pow10_i128(-mp); break; }
    default: a.rsc = mx(s1, s2); a.rp = mn(p1 - s1, p2 - s2) + a.rsc; a.dr.lmul = pow10_i128(a.rsc - s1); a.dr.rmul = pow10_i128(a.rsc - s2); break;
  }
  return a;
}
extern "C++" {
namespace dfgpu {
void decimal_arith_plan(int op, int p1, int s1, int p2, int s2, int* rp, int* rs, i128* lmul, i128* rmul) {
  ArithPlan a = plan_decimal(op, p1, s1, p2, s2); *rp = a.rp; *rs = a.rsc; *lmul = a.dr.lmul; *rmul = a.dr.rmul;
}
}  // namespace dfgpu
}
dfgpu_status dfgpu_binary_fused2(dfgpu_ctx* ctx, int32_t op_outer, const dfgpu_array* x, int32_t op_inner, const dfgpu_array* scalar, const dfgpu_array* y,
                                 int32_t scalar_on_left, int32_t inner_on_left, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!x || !scalar || !y || !out) fail(DFGPU_INVALID_ARGUMENT, "binary_fused2: null argument");
    auto arith = [](int op) { return op >= DFGPU_OP_ADD && op <= DFGPU_OP_REM; };
    int32_t t = x->type;
    // the plain shape only; everything else is two dfgpu_binary calls (the caller falls back on NOT_IMPLEMENTED)
    if (!arith(op_outer) || !arith(op_inner) || (t != DFGPU_DECIMAL128 && t != DFGPU_FLOAT64) || y->type != t || scalar->type != t || x->length != y->length || scalar->length != 1 ||
        x->validity || y->validity || !scalar->has_host_scalar || !scalar->host_scalar_valid || x->length == 0)
      fail(DFGPU_NOT_IMPLEMENTED, "binary_fused2: unsupported operand shape");
    int64_t n = x->length;
    Fused2 f{}; f.op_in = op_inner; f.op_out = op_outer; f.s_left = scalar_on_left ? 1 : 0; f.inner_left = inner_on_left ? 1 : 0; f.dr_in = DecRule{1, 1}; f.dr_out = DecRule{1, 1};
    int32_t rp = 0, rsc = 0;
    if (t == DFGPU_DECIMAL128) {
      memcpy(&f.s_i, scalar->host_scalar, 16);
      int sp = scalar->precision, ss = scalar->scale, yp = y->precision, ys = y->scale;
      ArithPlan in = scalar_on_left ? plan_decimal(op_inner, sp, ss, yp, ys) : plan_decimal(op_inner, yp, ys, sp, ss);
      ArithPlan o = inner_on_left ? plan_decimal(op_outer, in.rp, in.rsc, x->precision, x->scale) : plan_decimal(op_outer, x->precision, x->scale, in.rp, in.rsc);
      f.dr_in = in.dr; f.dr_out = o.dr; rp = o.rp; rsc = o.rsc;
    } else memcpy(&f.s_f, scalar->host_scalar, 8);
    ArrayHolder h(new_fixed(ctx, t, n, rp, rsc, false));
    { KernelTimer kt_(ctx, "k_arith");
      dim3 grid(grid_for(n, BLOCK)), block(BLOCK);
      if (t == DFGPU_DECIMAL128) hipLaunchKernelGGL((k_arith_fused2<true>), grid, block, 0, ctx->stream, f, x->values->ptr, y->values->ptr, n, h.get()->values->ptr, ctx->d_flags, row_selection_words(ctx, n));
      else hipLaunchKernelGGL((k_arith_fused2<false>), grid, block, 0, ctx->stream, f, x->values->ptr, y->values->ptr, n, h.get()->values->ptr, ctx->d_flags, row_selection_words(ctx, n));
      KERNEL_CHECK(); }
    h.get()->null_count = 0;
    if (t == DFGPU_DECIMAL128 || op_outer == DFGPU_OP_DIV || op_outer == DFGPU_OP_REM || op_inner == DFGPU_OP_DIV || op_inner == DFGPU_OP_REM) check_flags(ctx, "binary arithmetic");
    *out = h.release();
  });
}

dfgpu_status dfgpu_not(dfgpu_ctx* ctx, const dfgpu_array* a, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!a || a->type != DFGPU_BOOL) fail(DFGPU_INVALID_ARGUMENT, "NOT needs a Boolean array");
    ArrayHolder h(new_fixed(ctx, DFGPU_BOOL, a->length));
    int64_t nw = (a->length + 63) / 64;
    if (nw) hipLaunchKernelGGL(k_not_words, dim3(grid_for(nw, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)a->values->ptr, (uint64_t*)h.get()->values->ptr, nw);
    KERNEL_CHECK();
    h.get()->validity = a->validity; h.get()->null_count = a->null_count;    // validity buffer is shared (immutable)
    *out = h.release();
  });
}
dfgpu_status dfgpu_is_null(dfgpu_ctx* ctx, const dfgpu_array* a, int32_t negate, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!a) fail(DFGPU_INVALID_ARGUMENT, "is_null: null argument");
    ArrayHolder h(new_fixed(ctx, DFGPU_BOOL, a->length));
    if (a->length) hipLaunchKernelGGL(k_is_null, dim3(grid_for(a->length, BLOCK)), dim3(BLOCK), 0, ctx->stream, make_view(a), a->length, negate ? 1 : 0, (uint64_t*)h.get()->values->ptr);
    KERNEL_CHECK();
    *out = h.release();
  });
}
dfgpu_status dfgpu_negative(dfgpu_ctx* ctx, const dfgpu_array* a, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!a) fail(DFGPU_INVALID_ARGUMENT, "negative: null argument");
    int32_t t = logical_type(a);
    if (!(is_signed_int(t) || is_float(t) || t == DFGPU_DECIMAL128)) fail(DFGPU_NOT_IMPLEMENTED, "negative on type %d", t);
    ColView v = make_view(a);
    ArrayHolder h(new_fixed(ctx, t, a->length, v.precision, v.scale));
    if (a->length) hipLaunchKernelGGL(k_negative, dim3(grid_for(a->length, BLOCK)), dim3(BLOCK), 0, ctx->stream, v, a->length, h.get()->values->ptr);
    KERNEL_CHECK();
    if (a->type != DFGPU_DICTIONARY) { h.get()->validity = a->validity; h.get()->null_count = a->null_count; }
    else if (may_have_nulls(a)) { h.get()->validity = alloc_buffer(ctx, bitmap_bytes(a->length), true); hipLaunchKernelGGL(k_is_null, dim3(grid_for(a->length, BLOCK)), dim3(BLOCK), 0, ctx->stream, v, a->length, 1, (uint64_t*)h.get()->validity->ptr); h.get()->null_count = -1; }
    *out = h.release();
  });
}
dfgpu_status dfgpu_cast(dfgpu_ctx* ctx, const dfgpu_array* a, int32_t to, int32_t p, int32_t s, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!a || !out) fail(DFGPU_INVALID_ARGUMENT, "cast: null argument");
    int32_t from = logical_type(a);
    if (from == to && from != DFGPU_DECIMAL128 && a->type != DFGPU_DICTIONARY) { dfgpu_array_retain(const_cast<dfgpu_array*>(a)); *out = const_cast<dfgpu_array*>(a); return; }
    bool okf = from == DFGPU_BOOL || from == DFGPU_DATE32 || (from >= DFGPU_INT8 && from <= DFGPU_FLOAT64) || from == DFGPU_DECIMAL128;
    bool okt = to == DFGPU_BOOL || to == DFGPU_DATE32 || (to >= DFGPU_INT8 && to <= DFGPU_FLOAT64) || to == DFGPU_DECIMAL128;
    if (!okf || !okt || (to == DFGPU_BOOL && !(from == DFGPU_BOOL || (from >= DFGPU_INT8 && from <= DFGPU_UINT64)))) fail(DFGPU_NOT_IMPLEMENTED, "cast %d -> %d", from, to);
    if (to == DFGPU_DECIMAL128 && (p < 1 || p > 38 || s < 0 || s > p)) fail(DFGPU_INVALID_ARGUMENT, "cast: bad Decimal128(%d, %d)", p, s);
    ColView v = make_view(a);
    ArrayHolder h(new_fixed(ctx, to, a->length, to == DFGPU_DECIMAL128 ? p : 0, to == DFGPU_DECIMAL128 ? s : 0));
    { KernelTimer kt_(ctx, "k_cast");
      if (a->length) hipLaunchKernelGGL(k_cast, dim3(grid_for(a->length, BLOCK)), dim3(BLOCK), 0, ctx->stream, v, a->length, to, p, s, h.get()->values->ptr, ctx->d_flags, row_selection_words(ctx, a->length));
      KERNEL_CHECK(); }
    if (a->type != DFGPU_DICTIONARY) { h.get()->validity = a->validity; h.get()->null_count = a->null_count; }
    else if (may_have_nulls(a)) { h.get()->validity = alloc_buffer(ctx, bitmap_bytes(a->length), true); hipLaunchKernelGGL(k_is_null, dim3(grid_for(a->length, BLOCK)), dim3(BLOCK), 0, ctx->stream, v, a->length, 1, (uint64_t*)h.get()->validity->ptr); h.get()->null_count = -1; }
    check_flags(ctx, "cast");
    *out = h.release();
  });
}
dfgpu_status dfgpu_in_list(dfgpu_ctx* ctx, const dfgpu_array* a, const dfgpu_array* list, int32_t negated, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!a || !list || !out) fail(DFGPU_INVALID_ARGUMENT, "in_list: null argument");
    if (logical_type(a) != logical_type(list)) fail(DFGPU_INVALID_ARGUMENT, "in_list: value type %d vs list type %d", logical_type(a), logical_type(list));
    ArrayHolder h(new_fixed(ctx, DFGPU_BOOL, a->length, 0, 0, true));
    if (a->length) hipLaunchKernelGGL(k_in_list, dim3(grid_for(a->length, BLOCK)), dim3(BLOCK), 0, ctx->stream, make_view(a), make_view(list), list->length, a->length, negated ? 1 : 0,
                                      (uint64_t*)h.get()->values->ptr, (uint64_t*)h.get()->validity->ptr);
    KERNEL_CHECK(); h.get()->null_count = -1;
    *out = h.release();
  });
}

}  // extern "C"
