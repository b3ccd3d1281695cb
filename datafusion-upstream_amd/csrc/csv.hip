// csv.hip -- delimited text -> Arrow columns in HBM (the other format of SURVEY 8(f) rank 2).
//
// Replaces what CsvExec's stream does per file on the CPU (core/src/datasource/physical_plan/csv.rs: CsvOpener -> the `arrow-csv` reader, arrow-rs 50 -- a dependency that
// is not part of /root/reference; restated here: RFC 4180 records, a field may be quoted, a doubled quote inside quotes is one quote, an empty field of a
// non-string column is NULL, an empty string field is the empty string).  The schema comes from the caller (CsvExec has it from the table definition).  Every step
// runs on the device over the file image:
//   k_csv_quotes    quote characters per 4096-byte block (parity decides which record separators are real)
//   k_csv_rows      line feeds outside quotes -> bitmap (parity before each block from a scan of the block counts)
//   k_csv_fields    one lane per record: walks its bytes once, leaves (start, length, quoted) of every projected field
//   k_csv_parse_*   fields -> Int / Date32 / Decimal128 / Boolean / Float64 values + validity; Utf8 through lengths -> scan -> copy (doubled quotes collapse)
// Float64 takes the exact path (<= 15 significant digits, |exponent| <= 22: one correctly rounded multiplication or division); anything longer raises the cast flag
// instead of guessing.
#include "device_utils.h"

namespace dfgpu {

constexpr int CSV_BLK = 1024;             // bytes per wave: 16 per lane

// the 16 bytes of lane `l` of the 1 KB block at b0 (zero beyond the image); whole-vector load when the block lies inside the image and the base is 16-byte aligned
__device__ inline void csv_load16(const uint8_t* __restrict__ d, int64_t n, int64_t p, uint8_t* b) {
  if (p + 16 <= n && (((uintptr_t)(d + p)) & 15) == 0) { *(uint4*)b = *(const uint4*)(d + p); return; }
#pragma unroll
  for (int k = 0; k < 16; k++) b[k] = p + k < n ? d[p + k] : (uint8_t)0;
}
// escape character (CsvExec::escape, csv.rs:59; csv-core: inside quoted fields `escape` + any byte is that byte): a quote preceded by an odd run of escape characters is data.
// Quotes are sparse and such runs short: every quote of a lane looks back over the bytes in front of it.
__device__ inline uint32_t csv_real_quotes(const uint8_t* __restrict__ d, int64_t p, uint32_t q, int esc) {
  if (esc < 0 || !q) return q;
  uint32_t out = q;
  for (uint32_t m = q; m; m &= m - 1) { const int k = __ffs((int)m) - 1; int64_t i = p + k; uint32_t run = 0; while (i > 0 && d[i - 1] == (uint8_t)esc) { run++; i--; } if (run & 1u) out &= ~(1u << k); }
  return out;
}
__device__ inline uint32_t csv_mask16(const uint8_t* b, uint8_t c) { uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) m |= (uint32_t)(b[k] == c) << k;
  return m; }

// quote characters per 1 KB block (their parity decides which line feeds are real)
__global__ void __launch_bounds__(BLOCK) k_csv_quotes(const uint8_t* __restrict__ d, int64_t n, uint8_t quote, int esc, int64_t nblk, uint32_t* __restrict__ counts) {
  const int64_t blk = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); if (blk >= nblk) return;
  alignas(16) uint8_t b[16]; csv_load16(d, n, blk * CSV_BLK + lane_id() * 16, b);
  uint32_t c = (uint32_t)__popc(csv_real_quotes(d, blk * CSV_BLK + lane_id() * 16, csv_mask16(b, quote), esc));
#pragma unroll
  for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o, 64);
  if (lane_id() == 0) counts[blk] = c;
}
// bit i of `rows` = byte i ends a record: a line feed outside quotes that does not close a blank line (nothing, or a lone carriage return, since the previous line feed:
// arrow-csv skips blank lines), or the last byte of the image when that is not a line feed.  One wave per 1 KB, no LDS: prefix parity inside the lane's 16 bits by
// shift-xor, across lanes by a ballot of the lanes' parities.
__global__ void __launch_bounds__(BLOCK) k_csv_rows(const uint8_t* __restrict__ d, int64_t n, uint8_t quote, int esc, int64_t nblk, const uint64_t* __restrict__ before, uint16_t* __restrict__ rows, uint32_t* flags) {
  const int64_t blk = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); if (blk >= nblk) return;
  const int64_t p = blk * CSV_BLK + lane_id() * 16;
  alignas(16) uint8_t b[16]; csv_load16(d, n, p, b);
  const uint32_t q = csv_real_quotes(d, p, csv_mask16(b, quote), esc), nl = csv_mask16(b, '\n'), cr = csv_mask16(b, '\r');
  uint32_t incl = q; incl ^= incl << 1; incl ^= incl << 2; incl ^= incl << 4; incl ^= incl << 8; incl &= 0xFFFFu;     // bit k = parity of quotes in bytes 0..k of the lane
  const uint32_t lane_par = (uint32_t)__popc(q) & 1u;
  const uint32_t in0 = ((uint32_t)(before[blk] & 1) + (uint32_t)__popcll(ballot64(lane_par != 0) & lanemask_lt())) & 1u;       // inside quotes in front of the lane's first byte
  const uint32_t flip = in0 ? 0xFFFFu : 0u;
  const uint32_t inside = ((incl << 1) & 0xFFFFu) ^ flip, after = incl ^ flip;        // before / after each byte
  // the two bytes in front of the lane's first (the previous lane's last two; lane 0 reads them)
  uint32_t pnl = __shfl_up(nl >> 14, 1, 64), pcr = __shfl_up(cr >> 15, 1, 64);
  if (lane_id() == 0) { pnl = (p >= 1 ? (uint32_t)(d[p - 1] == '\n') << 1 : 2u) | (p >= 2 ? (uint32_t)(d[p - 2] == '\n') : 1u); if (p == 1) pnl |= 1u; pcr = p >= 1 ? (uint32_t)(d[p - 1] == '\r') : 0u; }
  const uint32_t prev_nl = ((nl << 1) | (pnl >> 1)) & 0xFFFFu, prev2_nl = ((nl << 2) | pnl) & 0xFFFFu, prev_cr = ((cr << 1) | pcr) & 0xFFFFu;    // "start of image" counts as a line feed
  const uint32_t blank = nl & (prev_nl | (prev_cr & prev2_nl));
  uint32_t end = nl & ~inside & ~blank;
  if (p <= n - 1 && n - 1 < p + 16) { const uint32_t k = (uint32_t)(n - 1 - p);
    if (!((nl >> k) & 1u) && !((after >> k) & 1u)) end |= 1u << k;
    if ((after >> k) & 1u) atomicOr(flags, DFGPU_FLAG_CAST); }                           // the image ends inside a quoted field
  rows[p >> 4] = (uint16_t)end;
}

struct CsvField { uint32_t start, len; };      // len bit 31: the field was quoted (content excludes the outer quotes; doubled quotes still doubled)
// one lane per record: fields wanted[k] (ascending file column indices) -> out[k * nrows + row]
__global__ void __launch_bounds__(BLOCK) k_csv_fields(const uint8_t* __restrict__ d, const uint32_t* __restrict__ ends, int64_t first_row, int64_t nrows, uint8_t delim, uint8_t quote, int esc,
                                                      const int32_t* __restrict__ wanted, int32_t nwanted, int32_t ncols_file, CsvField* __restrict__ out, uint32_t* flags) {
  const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (r >= nrows) return;
  const int64_t fr = first_row + r;
  uint32_t p = fr == 0 ? 0u : ends[fr - 1] + 1u, e = ends[fr];
  if (e >= p && d[e] == '\n') { if (e > p && d[e - 1] == '\r') e--; } else e++;        // e = one past the record's last content byte
  int32_t col = 0, w = 0; bool bad = false;
  while (w < nwanted) {
    uint32_t s = p, len; bool quoted = false;
    if (p < e && d[p] == quote) {
      quoted = true; s = ++p;
      for (;;) { if (p >= e) { bad = true; break; } if (esc >= 0 && d[p] == (uint8_t)esc && p + 1 < e) { p += 2; continue; } if (d[p] == quote) { if (p + 1 < e && d[p + 1] == quote) { p += 2; continue; } break; } p++; }
      len = p - s; if (!bad) p++;                                   // past the closing quote
      if (!bad && p < e && d[p] != delim) bad = true;
    } else { while (p < e && d[p] != delim) p++; len = p - s; }
    if (bad) break;
    if (col == wanted[w]) { out[(int64_t)w * nrows + r] = CsvField{s, len | (quoted ? 0x80000000u : 0u)}; w++; }
    col++;
    if (p < e) p++; else if (w < nwanted) { bad = col <= wanted[w]; break; }            // record ended before a wanted column
  }
  (void)ncols_file;
  if (bad) { atomicOr(flags, DFGPU_FLAG_CAST); for (; w < nwanted; w++) out[(int64_t)w * nrows + r] = CsvField{0u, 0u}; }      // the parse kernels still run: give them empty fields
}

__device__ inline bool csv_int(const uint8_t* s, uint32_t len, bool allow_neg, uint64_t max_mag_pos, i128* out) {
  if (!len) return false;
  uint32_t i = 0; bool neg = false;
  if (s[0] == '-' || s[0] == '+') { neg = s[0] == '-'; i = 1; if (len == 1 || (neg && !allow_neg)) return false; }
  u128 v = 0;
  for (; i < len; i++) { uint32_t c = s[i] - '0'; if (c > 9) return false; v = v * 10 + c; if (v > ((u128)1 << 100)) return false; }
  if (v > (u128)max_mag_pos + (neg ? 1 : 0)) return false;
  *out = neg ? -(i128)v : (i128)v; return true;
}
__global__ void __launch_bounds__(BLOCK) k_csv_parse_int(const uint8_t* __restrict__ d, const CsvField* __restrict__ f, int64_t n, int32_t type, int32_t width, void* __restrict__ vals, uint64_t* __restrict__ valid, uint32_t* flags) {
  const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; bool ok = false; i128 v = 0; bool bad = false;
  if (r < n) {
    const uint32_t len = f[r].len & 0x7FFFFFFFu; const uint8_t* s = d + f[r].start;
    if (len) {
      uint64_t mx; bool sg = true;
      switch (type) { case DFGPU_INT8: mx = 127; break; case DFGPU_INT16: mx = 32767; break; case DFGPU_INT32: mx = 2147483647ull; break; case DFGPU_INT64: mx = 9223372036854775807ull; break;
                      case DFGPU_UINT8: mx = 255; sg = false; break; case DFGPU_UINT16: mx = 65535; sg = false; break; case DFGPU_UINT32: mx = 4294967295ull; sg = false; break; default: mx = ~0ull; sg = false; break; }
      ok = csv_int(s, len, sg, mx, &v); bad = !ok;
      if (!sg && ok && v < 0) { ok = false; bad = true; }
    }
    switch (width) { case 1: ((uint8_t*)vals)[r] = (uint8_t)(int64_t)v; break; case 2: ((uint16_t*)vals)[r] = (uint16_t)(int64_t)v; break; case 4: ((uint32_t*)vals)[r] = (uint32_t)(int64_t)v; break; default: ((uint64_t*)vals)[r] = (uint64_t)v; break; }
  }
  const uint64_t m = ballot64(ok);
  if (lane_id() == 0 && (r >> 6) < ((n + 63) >> 6)) valid[r >> 6] = m;
  if (bad) atomicOr(flags, DFGPU_FLAG_CAST);
}
__device__ inline int64_t days_from_civil(int64_t y, unsigned m, unsigned dd) {
  y -= m <= 2; const int64_t era = (y >= 0 ? y : y - 399) / 400; const unsigned yoe = (unsigned)(y - era * 400);
  const unsigned doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + dd - 1, doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + (int64_t)doe - 719468;
}
__global__ void __launch_bounds__(BLOCK) k_csv_parse_date(const uint8_t* __restrict__ d, const CsvField* __restrict__ f, int64_t n, int32_t* __restrict__ vals, uint64_t* __restrict__ valid, uint32_t* flags) {
  const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; bool ok = false, bad = false; int32_t v = 0;
  if (r < n) {
    const uint32_t len = f[r].len & 0x7FFFFFFFu; const uint8_t* s = d + f[r].start;
    if (len) {
      bad = true;
      if (len == 10 && s[4] == '-' && s[7] == '-') {
        unsigned y = 0, mo = 0, dy = 0; bool dg = true;
        for (int i = 0; i < 4; i++) { unsigned c = s[i] - '0'; dg &= c <= 9; y = y * 10 + c; }
        for (int i = 5; i < 7; i++) { unsigned c = s[i] - '0'; dg &= c <= 9; mo = mo * 10 + c; }
        for (int i = 8; i < 10; i++) { unsigned c = s[i] - '0'; dg &= c <= 9; dy = dy * 10 + c; }
        const unsigned mdays[12] = {31, (y % 4 == 0 && (y % 100 != 0 || y % 400 == 0)) ? 29u : 28u, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
        if (dg && mo >= 1 && mo <= 12 && dy >= 1 && dy <= mdays[mo - 1]) { v = (int32_t)days_from_civil((int64_t)y, mo, dy); ok = true; bad = false; }
      }
    }
    vals[r] = v;
  }
  const uint64_t m = ballot64(ok);
  if (lane_id() == 0 && (r >> 6) < ((n + 63) >> 6)) valid[r >> 6] = m;
  if (bad) atomicOr(flags, DFGPU_FLAG_CAST);
}
// sign, digits, at most one '.', fraction digits beyond the scale are checked and dropped, fewer are padded (arrow-cast parse_decimal)
__global__ void __launch_bounds__(BLOCK) k_csv_parse_decimal(const uint8_t* __restrict__ d, const CsvField* __restrict__ f, int64_t n, int32_t precision, int32_t scale, uint64_t* __restrict__ vals, uint64_t* __restrict__ valid, uint32_t* flags) {
  const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; bool ok = false, bad = false; i128 v = 0;
  if (r < n) {
    const uint32_t len = f[r].len & 0x7FFFFFFFu; const uint8_t* s = d + f[r].start;
    if (len) {
      uint32_t i = 0; bool neg = false, dot = false, any = false; int32_t frac = 0; u128 m = 0; bad = false;
      if (s[0] == '-' || s[0] == '+') { neg = s[0] == '-'; i = 1; }
      for (; i < len && !bad; i++) {
        if (s[i] == '.') { if (dot) bad = true; dot = true; continue; }
        uint32_t c = s[i] - '0'; if (c > 9) { bad = true; break; }
        any = true;
        if (dot) { if (frac >= scale) continue; frac++; }
        m = m * 10 + c; if (m > ((u128)1 << 120)) bad = true;
      }
      if (!any) bad = true;
      for (; frac < scale && !bad; frac++) { m *= 10; if (m > ((u128)1 << 124)) bad = true; }
      if (!bad) { u128 lim = 1; for (int k = 0; k < precision; k++) lim *= 10; if (m >= lim) bad = true; }
      if (!bad) { v = neg ? -(i128)m : (i128)m; ok = true; }
    }
    vals[2 * r] = (uint64_t)v; vals[2 * r + 1] = (uint64_t)(v >> 64);
  }
  const uint64_t mm = ballot64(ok);
  if (lane_id() == 0 && (r >> 6) < ((n + 63) >> 6)) valid[r >> 6] = mm;
  if (bad) atomicOr(flags, DFGPU_FLAG_CAST);
}
__global__ void __launch_bounds__(BLOCK) k_csv_parse_bool(const uint8_t* __restrict__ d, const CsvField* __restrict__ f, int64_t n, uint64_t* __restrict__ vals, uint64_t* __restrict__ valid, uint32_t* flags) {
  const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; bool ok = false, bad = false, v = false;
  if (r < n) {
    const uint32_t len = f[r].len & 0x7FFFFFFFu; const uint8_t* s = d + f[r].start;
    if (len) {
      auto low = [](uint8_t c) { return (uint8_t)(c >= 'A' && c <= 'Z' ? c + 32 : c); };
      if (len == 4 && low(s[0]) == 't' && low(s[1]) == 'r' && low(s[2]) == 'u' && low(s[3]) == 'e') { v = true; ok = true; }
      else if (len == 5 && low(s[0]) == 'f' && low(s[1]) == 'a' && low(s[2]) == 'l' && low(s[3]) == 's' && low(s[4]) == 'e') ok = true;
      else bad = true;
    }
  }
  const uint64_t vm = ballot64(v), m = ballot64(ok);
  if (lane_id() == 0 && (r >> 6) < ((n + 63) >> 6)) { vals[r >> 6] = vm; valid[r >> 6] = m; }
  if (bad) atomicOr(flags, DFGPU_FLAG_CAST);
}
// decimal notation with an optional exponent; exact when the digits fit 2^53 and the power of ten is exact in a double (<= 10^22): one rounding
__global__ void __launch_bounds__(BLOCK) k_csv_parse_f64(const uint8_t* __restrict__ d, const CsvField* __restrict__ f, int64_t n, double* __restrict__ vals, uint64_t* __restrict__ valid, uint32_t* flags) {
  const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; bool ok = false, bad = false; double v = 0;
  if (r < n) {
    const uint32_t len = f[r].len & 0x7FFFFFFFu; const uint8_t* s = d + f[r].start;
    if (len) {
      uint32_t i = 0; bool neg = false, dot = false, any = false; uint64_t m = 0; int32_t e10 = 0; int digits = 0;
      if (s[0] == '-' || s[0] == '+') { neg = s[0] == '-'; i = 1; }
      for (; i < len; i++) {
        const uint8_t c = s[i];
        if (c == '.') { if (dot) { bad = true; break; } dot = true; continue; }
        if (c == 'e' || c == 'E') break;
        const uint32_t dg = c - '0'; if (dg > 9) { bad = true; break; }
        any = true;
        if (m == 0 && dg == 0) { if (dot) e10--; continue; }             // leading zeros carry no digits
        if (digits >= 15) { bad = true; break; }                          // beyond the exact path
        m = m * 10 + dg; digits++; if (dot) e10--;
      }
      if (!bad && i < len) {                                              // exponent
        i++; bool eneg = false; int32_t ex = 0; bool ed = false;
        if (i < len && (s[i] == '-' || s[i] == '+')) { eneg = s[i] == '-'; i++; }
        for (; i < len; i++) { const uint32_t dg = s[i] - '0'; if (dg > 9 || ex > 400) { bad = true; break; } ex = ex * 10 + (int32_t)dg; ed = true; }
        if (!ed) bad = true;
        e10 += eneg ? -ex : ex;
      }
      if (!any) bad = true;
      if (!bad) {
        const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
        if (m == 0) v = 0.0; else if (e10 >= 0 && e10 <= 22) v = (double)m * p10[e10]; else if (e10 < 0 && e10 >= -22) v = (double)m / p10[-e10]; else bad = true;
        if (!bad) { if (neg) v = -v; ok = true; }
      }
    }
    vals[r] = v;
  }
  const uint64_t m2 = ballot64(ok);
  if (lane_id() == 0 && (r >> 6) < ((n + 63) >> 6)) valid[r >> 6] = m2;
  if (bad) atomicOr(flags, DFGPU_FLAG_CAST);
}
// Utf8: byte length after collapsing doubled quotes, then the copy
__global__ void __launch_bounds__(BLOCK) k_csv_str_len(const uint8_t* __restrict__ d, const CsvField* __restrict__ f, int64_t n, uint8_t quote, int esc, uint32_t* __restrict__ lens) {
  const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (r >= n) return;
  const uint32_t raw = f[r].len, len = raw & 0x7FFFFFFFu; uint32_t out = len;
  if (raw >> 31) { const uint8_t* s = d + f[r].start; for (uint32_t i = 0; i + 1 < len; i++) if ((s[i] == quote && s[i + 1] == quote) || (esc >= 0 && s[i] == (uint8_t)esc)) { out--; i++; } }      // a doubled quote, or escape + byte, is one byte
  lens[r] = out;
}
__global__ void __launch_bounds__(BLOCK) k_csv_str_copy(const uint8_t* __restrict__ d, const CsvField* __restrict__ f, int64_t n, uint8_t quote, int esc, const int32_t* __restrict__ offsets, uint8_t* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; if (r >= n) return;
  const uint32_t raw = f[r].len, len = raw & 0x7FFFFFFFu; const uint8_t* s = d + f[r].start; uint8_t* o = out + offsets[r];
  if (raw >> 31) { for (uint32_t i = 0; i < len; i++) { if (esc >= 0 && s[i] == (uint8_t)esc && i + 1 < len) { *o++ = s[++i]; continue; } *o++ = s[i]; if (s[i] == quote && i + 1 < len && s[i + 1] == quote) i++; } }
  else for (uint32_t i = 0; i < len; i++) o[i] = s[i];
}
__global__ void k_csv_set_i32(int32_t* p, int32_t v) { *p = v; }

}  // namespace dfgpu
using namespace dfgpu;

extern "C" dfgpu_status dfgpu_csv_read(dfgpu_ctx* ctx, const uint8_t* bytes, int64_t len, int32_t bytes_on_device, int32_t delimiter, int32_t quote, int32_t escape, int32_t has_header, int32_t ncols_file,
                                       const int32_t* columns, const int32_t* types /* 3 per projected column: type, precision, scale */, int32_t ncols, dfgpu_array** out, int64_t* out_rows) {
  return guard(ctx, [&] {
    if (!bytes || len < 0 || !out || ncols < 1 || !columns || !types || ncols_file < 1) fail(DFGPU_INVALID_ARGUMENT, "csv_read: bad argument");
    if (len > 0xFFFFFFF0ll) fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: CSV images above 4 GB in one call (split the byte range)");
    for (int c = 0; c < ncols; c++) { if (columns[c] < 0 || columns[c] >= ncols_file || (c && columns[c] <= columns[c - 1])) fail(DFGPU_INVALID_ARGUMENT, "csv_read: projected columns must be ascending file column indices");
      int32_t t = types[3 * c]; if (!(type_width(t) || t == DFGPU_BOOL || t == DFGPU_UTF8) || t == DFGPU_FLOAT32) fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: CSV column of type %d on the device", t); }
    HIP_CHECK(hipSetDevice(ctx->device));
    BufferPtr img; const uint8_t* d = bytes;
    if (!bytes_on_device) { img = alloc_buffer(ctx, (size_t)len + 64); if (len) HIP_CHECK(hipMemcpyAsync(img->ptr, bytes, (size_t)len, hipMemcpyHostToDevice, ctx->stream)); d = (const uint8_t*)img->ptr; }
    const uint8_t dl = (uint8_t)delimiter, qt = (uint8_t)quote; const int esc = escape > 0 && escape < 256 && escape != quote ? escape : -1;
    // ---- records
    const int64_t nblk = (len + CSV_BLK - 1) / CSV_BLK;
    ArrayHolder ends;
    { KernelTimer kt(ctx, "csv_rows");
      BufferPtr qc = alloc_buffer(ctx, (size_t)(nblk + 1) * 4), qb = alloc_buffer(ctx, (size_t)(nblk + 1) * 8), bits = alloc_buffer(ctx, (size_t)nblk * (CSV_BLK / 8) + 16, true);
      if (nblk) {
        const dim3 g(grid_for(nblk, BLOCK / 64));
        hipLaunchKernelGGL(k_csv_quotes, g, dim3(BLOCK), 0, ctx->stream, d, len, qt, esc, nblk, (uint32_t*)qc->ptr);
        exclusive_scan_u32(ctx, (const uint32_t*)qc->ptr, (uint64_t*)qb->ptr, nblk, nullptr);
        hipLaunchKernelGGL(k_csv_rows, g, dim3(BLOCK), 0, ctx->stream, d, len, qt, esc, nblk, (const uint64_t*)qb->ptr, (uint16_t*)bits->ptr, ctx->d_flags);
        KERNEL_CHECK();
      }
      ends.a = mask_to_indices_impl(ctx, (const uint64_t*)bits->ptr, len); }
    const int64_t total = ends.get()->length, first = has_header && total ? 1 : 0, nrows = total - first;
    if (out_rows) *out_rows = nrows;
    // ---- fields
    BufferPtr fields = alloc_buffer(ctx, std::max<size_t>((size_t)ncols * (size_t)nrows * sizeof(CsvField), 16));
    BufferPtr want = alloc_buffer(ctx, (size_t)ncols * 4 + 16); HIP_CHECK(hipMemcpyAsync(want->ptr, columns, (size_t)ncols * 4, hipMemcpyHostToDevice, ctx->stream));
    if (nrows) { KernelTimer kt(ctx, "csv_fields");
      hipLaunchKernelGGL(k_csv_fields, dim3(grid_for(nrows, BLOCK)), dim3(BLOCK), 0, ctx->stream, d, (const uint32_t*)ends.get()->values->ptr, first, nrows, dl, qt, esc, (const int32_t*)want->ptr, ncols, ncols_file, (CsvField*)fields->ptr, ctx->d_flags);
      KERNEL_CHECK(); }
    // ---- values
    std::vector<ArrayHolder> res((size_t)ncols);
    KernelTimer kt(ctx, "csv_parse");
    for (int c = 0; c < ncols; c++) {
      const int32_t t = types[3 * c], pr = types[3 * c + 1], sc = types[3 * c + 2]; const CsvField* f = (const CsvField*)fields->ptr + (size_t)c * (size_t)nrows; dim3 g(grid_for(nrows, BLOCK));
      if (t == DFGPU_UTF8) {
        ArrayHolder a(new_array(ctx, DFGPU_UTF8, nrows)); BufferPtr off = alloc_buffer(ctx, (size_t)(nrows + 1) * 4 + 16);
        uint64_t tot = 0;
        if (nrows) {
          hipLaunchKernelGGL(k_csv_str_len, g, dim3(BLOCK), 0, ctx->stream, d, f, nrows, qt, esc, (uint32_t*)off->ptr);
          exclusive_scan_u32_inplace32(ctx, (uint32_t*)off->ptr, nrows, ctx->d_scratch64 + 41);
          tot = read_scratch(ctx, 41);
          if (tot > 0x7FFFFFFFull) fail(DFGPU_EXECUTION, "Arrow error: a Utf8 CSV column of %llu bytes overflows int32 offsets -- read a smaller byte range", (unsigned long long)tot);
        }
        hipLaunchKernelGGL(k_csv_set_i32, dim3(1), dim3(1), 0, ctx->stream, (int32_t*)off->ptr + nrows, (int32_t)tot);
        BufferPtr ch = alloc_buffer(ctx, std::max<size_t>((size_t)tot, 16));
        if (nrows && tot) hipLaunchKernelGGL(k_csv_str_copy, g, dim3(BLOCK), 0, ctx->stream, d, f, nrows, qt, esc, (const int32_t*)off->ptr, (uint8_t*)ch->ptr);
        a.get()->offsets = off; a.get()->values = ch; a.get()->values_bytes = (int64_t)tot; a.get()->null_count = 0;
        res[(size_t)c].a = a.release();
      } else {
        ArrayHolder a(new_fixed(ctx, t, nrows, t == DFGPU_DECIMAL128 ? pr : 0, t == DFGPU_DECIMAL128 ? sc : 0, true)); a.get()->null_count = -1;
        uint64_t* valid = (uint64_t*)a.get()->validity->ptr; void* vals = a.get()->values->ptr;
        if (nrows) {
          if (t == DFGPU_BOOL) hipLaunchKernelGGL(k_csv_parse_bool, g, dim3(BLOCK), 0, ctx->stream, d, f, nrows, (uint64_t*)vals, valid, ctx->d_flags);
          else if (t == DFGPU_DATE32) hipLaunchKernelGGL(k_csv_parse_date, g, dim3(BLOCK), 0, ctx->stream, d, f, nrows, (int32_t*)vals, valid, ctx->d_flags);
          else if (t == DFGPU_DECIMAL128) hipLaunchKernelGGL(k_csv_parse_decimal, g, dim3(BLOCK), 0, ctx->stream, d, f, nrows, pr, sc, (uint64_t*)vals, valid, ctx->d_flags);
          else if (t == DFGPU_FLOAT64) hipLaunchKernelGGL(k_csv_parse_f64, g, dim3(BLOCK), 0, ctx->stream, d, f, nrows, (double*)vals, valid, ctx->d_flags);
          else hipLaunchKernelGGL(k_csv_parse_int, g, dim3(BLOCK), 0, ctx->stream, d, f, nrows, t, type_width(t), vals, valid, ctx->d_flags);
        }
        res[(size_t)c].a = a.release();
      }
      KERNEL_CHECK();
    }
    check_flags(ctx, "CSV parse (a field does not parse as its column type, a record is short, or quoting is malformed)");
    for (int c = 0; c < ncols; c++) out[c] = res[(size_t)c].release();
  });
}
