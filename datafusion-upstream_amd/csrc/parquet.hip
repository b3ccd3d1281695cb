// parquet.hip -- Parquet column chunks -> Arrow columns in HBM (the scan below the operators).
//
// Replaces, for flat schemas, what ParquetExec's stream does per row group on the CPU (core/src/datasource/physical_plan/parquet/mod.rs:
// ParquetOpener :417-560 builds a ParquetRecordBatchStream of the `parquet` crate, arrow-rs 50 -- a dependency that is not part of
// /root/reference; the format itself is the published parquet-format specification, restated here).  The host side parses the Thrift
// compact footer and the page headers (bytes, no data); everything that touches values runs on the device:
//
//   k_pq_snappy   one wave per page: Snappy raw-format decompression; the last 64 KB of output live in an LDS ring (every back reference of
//                 the standard 64 KB-block compressor resolves there), tags are parsed out of an LDS window of the input
//   k_pq_decode   one workgroup per page: definition levels and dictionary indices (RLE / bit-packed hybrid, a batch of runs parsed by one
//                 lane, expanded by all), PLAIN values (fixed width, Boolean bits, length-prefixed byte arrays walked in an LDS window),
//                 NULL slots from the level prefix sums; writes Arrow values / dictionary keys / (length, source) of every string
//   k_pq_plain    wide path of PLAIN fixed-width pages without levels: (page, slice) grid
//   k_pq_chars    string bytes to their offsets after one device-wide scan of the lengths
//
// Utf8 columns keep their dictionary (option): the keys are the page indices plus the row group's base in the concatenated dictionary,
// so dictionary predicates and the canonical-id group-by take the column without ever expanding it.
#include <algorithm>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "device_utils.h"
#include "zstd_device.h"

namespace dfgpu {
namespace pq {

// ================================================================================ Thrift compact protocol (host)
struct TR {
  const uint8_t* p; const uint8_t* end;
  [[noreturn]] static void bad() { fail(DFGPU_EXECUTION, "Parquet error: truncated or malformed Thrift metadata"); }
  uint8_t u8() { if (p >= end) bad(); return *p++; }
  uint64_t varint() { uint64_t v = 0; for (int s = 0; s < 64; s += 7) { uint8_t b = u8(); v |= (uint64_t)(b & 0x7f) << s; if (!(b & 0x80)) return v; } bad(); }
  int64_t zz() { uint64_t v = varint(); return (int64_t)(v >> 1) ^ -(int64_t)(v & 1); }
  bool field(int16_t& id, int& type) { uint8_t b = u8(); if (b == 0) return false; type = b & 15; int d = b >> 4; if (d) id = (int16_t)(id + d); else id = (int16_t)zz(); return true; }
  std::string binary() { uint64_t n = varint(); if ((uint64_t)(end - p) < n) bad(); std::string s((const char*)p, (size_t)n); p += n; return s; }
  void list(int& elem_type, int64_t& n) { uint8_t b = u8(); elem_type = b & 15; n = b >> 4; if (n == 15) n = (int64_t)varint(); }
  void skip(int type, bool in_list = false) {
    switch (type) {
      case 1: case 2: if (in_list) u8(); break;
      case 3: u8(); break;
      case 4: case 5: case 6: varint(); break;
      case 7: if (end - p < 8) bad(); p += 8; break;
      case 8: { uint64_t n = varint(); if ((uint64_t)(end - p) < n) bad(); p += n; break; }
      case 9: case 10: { int et; int64_t n; list(et, n); for (int64_t i = 0; i < n; i++) skip(et, true); break; }
      case 11: { uint64_t n = varint(); if (n) { uint8_t kv = u8(); for (uint64_t i = 0; i < n; i++) { skip(kv >> 4, true); skip(kv & 15, true); } } break; }
      case 12: { int16_t id = 0; int t; while (field(id, t)) skip(t); break; }
      default: bad();
    }
  }
  template <typename F> void fields(F&& on) { int16_t id = 0; int t; while (field(id, t)) if (!on(id, t)) skip(t); }
};

enum { PT_BOOLEAN = 0, PT_INT32 = 1, PT_INT64 = 2, PT_INT96 = 3, PT_FLOAT = 4, PT_DOUBLE = 5, PT_BYTE_ARRAY = 6, PT_FLBA = 7 };
enum { ENC_PLAIN = 0, ENC_PLAIN_DICT = 2, ENC_RLE = 3, ENC_BIT_PACKED = 4, ENC_RLE_DICT = 8 };
enum { CODEC_NONE = 0, CODEC_SNAPPY = 1, CODEC_ZSTD = 6, CODEC_LZ4_RAW = 7 };
enum { PG_DATA = 0, PG_INDEX = 1, PG_DICT = 2, PG_DATA_V2 = 3 };

struct Leaf {
  std::string name; int phys = -1, type_len = 0, rep = 0, conv = -1, scale = 0, precision = 0, depth = 1;
  bool lt_string = false, lt_decimal = false, lt_date = false, lt_other = false; int lt_int_bits = 0; bool lt_int_signed = true;
  int32_t arrow = 0;          // DFGPU_* of the decoded column before the dictionary option; 0 = not supported by this slice
  std::string why;            // reason when arrow == 0
};
struct Chunk {
  int phys = -1, codec = 0; int64_t num_values = 0, data_off = -1, dict_off = -1, total_comp = 0, total_uncomp = 0, null_count = -1;
  bool has_minmax = false; std::string min_v, max_v;
};
struct RowGroup { int64_t rows = 0; std::vector<Chunk> cols; };

static void parse_statistics(TR& r, Chunk& c) {
  std::string mn, mx, mn_old, mx_old; bool a = false, b = false, ao = false, bo = false;
  r.fields([&](int id, int) {
    switch (id) {
      case 1: mx_old = r.binary(); bo = true; return true;
      case 2: mn_old = r.binary(); ao = true; return true;
      case 3: c.null_count = r.zz(); return true;
      case 5: mx = r.binary(); b = true; return true;
      case 6: mn = r.binary(); a = true; return true;
      default: return false;
    }
  });
  if (a && b) { c.has_minmax = true; c.min_v = mn; c.max_v = mx; }
  else if (ao && bo && (c.phys == PT_INT32 || c.phys == PT_INT64)) { c.has_minmax = true; c.min_v = mn_old; c.max_v = mx_old; }   // deprecated fields: signed order only
}
static void parse_column_meta(TR& r, Chunk& c) {
  r.fields([&](int id, int) {
    switch (id) {
      case 1: c.phys = (int)r.zz(); return true;
      case 4: c.codec = (int)r.zz(); return true;
      case 5: c.num_values = r.zz(); return true;
      case 6: c.total_uncomp = r.zz(); return true;
      case 7: c.total_comp = r.zz(); return true;
      case 9: c.data_off = r.zz(); return true;
      case 11: c.dict_off = r.zz(); return true;
      case 12: parse_statistics(r, c); return true;
      default: return false;
    }
  });
}
static void parse_logical_type(TR& r, Leaf& l) {
  r.fields([&](int id, int) {
    switch (id) {
      case 1: r.skip(12); l.lt_string = true; return true;
      case 5: r.fields([&](int f, int) { if (f == 1) { l.scale = (int)r.zz(); return true; } if (f == 2) { l.precision = (int)r.zz(); return true; } return false; }); l.lt_decimal = true; return true;
      case 6: r.skip(12); l.lt_date = true; return true;
      case 10: { int16_t fid = 0; int t; while (r.field(fid, t)) { if (fid == 1) l.lt_int_bits = (int8_t)r.u8(); else if (fid == 2) l.lt_int_signed = (t == 1); else r.skip(t); } return true; }
      case 11: r.skip(12); return true;                 // UNKNOWN (always null): falls back to the physical type
      default: r.skip(12); l.lt_other = true; return true;      // MAP / LIST / ENUM / TIME / TIMESTAMP / JSON / BSON / UUID / FLOAT16
    }
  });
}

// parquet -> arrow type of a leaf, the rules of the parquet crate's schema conversion (arrow-rs 50 parquet/src/arrow/schema/primitive.rs) for the supported subset
static void resolve_arrow_type(Leaf& l) {
  auto no = [&](const char* w) { l.arrow = 0; l.why = w; };
  if (l.depth != 1 || l.rep == 2) return no("nested or repeated column");
  if (l.lt_other) return no("logical type outside this slice (time / timestamp / list / map / enum / json / uuid)");
  bool dec = l.lt_decimal || l.conv == 5;
  switch (l.phys) {
    case PT_BOOLEAN: l.arrow = DFGPU_BOOL; return;
    case PT_INT32:
      if (dec) { l.arrow = DFGPU_DECIMAL128; return; }
      if (l.lt_date || l.conv == 6) { l.arrow = DFGPU_DATE32; return; }
      if (l.lt_int_bits) { int b = l.lt_int_bits; bool s = l.lt_int_signed; l.arrow = b == 8 ? (s ? DFGPU_INT8 : DFGPU_UINT8) : b == 16 ? (s ? DFGPU_INT16 : DFGPU_UINT16) : b == 32 ? (s ? DFGPU_INT32 : DFGPU_UINT32) : 0; if (!l.arrow) l.why = "integer width"; return; }
      switch (l.conv) { case 11: l.arrow = DFGPU_UINT8; return; case 12: l.arrow = DFGPU_UINT16; return; case 13: l.arrow = DFGPU_UINT32; return; case 15: l.arrow = DFGPU_INT8; return; case 16: l.arrow = DFGPU_INT16; return;
                        case 17: case -1: l.arrow = DFGPU_INT32; return; default: return no("converted type of an INT32 column outside this slice"); }
    case PT_INT64:
      if (dec) { l.arrow = DFGPU_DECIMAL128; return; }
      if (l.lt_int_bits) { l.arrow = l.lt_int_bits == 64 ? (l.lt_int_signed ? DFGPU_INT64 : DFGPU_UINT64) : 0; if (!l.arrow) l.why = "integer width"; return; }
      switch (l.conv) { case 14: l.arrow = DFGPU_UINT64; return; case 18: case -1: l.arrow = DFGPU_INT64; return; default: return no("converted type of an INT64 column outside this slice (timestamp / time)"); }
    case PT_FLOAT: l.arrow = DFGPU_FLOAT32; return;
    case PT_DOUBLE: l.arrow = DFGPU_FLOAT64; return;
    case PT_BYTE_ARRAY: if (l.lt_string || l.conv == 0) { l.arrow = DFGPU_UTF8; return; } return no("BYTE_ARRAY without a string annotation (Binary)");
    case PT_FLBA: if (dec && l.type_len >= 1 && l.type_len <= 16) { l.arrow = DFGPU_DECIMAL128; return; } return no("FIXED_LEN_BYTE_ARRAY that is not a decimal of at most 16 bytes");
    default: return no("INT96");
  }
}

struct PageHdr { int type = -1; int32_t usize = 0, csize = 0, nvals = 0; int enc = 0; int32_t def_len = 0, rep_len = 0, num_nulls = -1; bool v2_compressed = true; int hdr_bytes = 0; };
static PageHdr parse_page_header(const uint8_t* p, const uint8_t* end) {
  TR r{p, end}; PageHdr h;
  r.fields([&](int id, int) {
    switch (id) {
      case 1: h.type = (int)r.zz(); return true;
      case 2: h.usize = (int32_t)r.zz(); return true;
      case 3: h.csize = (int32_t)r.zz(); return true;
      case 5: r.fields([&](int f, int) { if (f == 1) { h.nvals = (int32_t)r.zz(); return true; } if (f == 2) { h.enc = (int)r.zz(); return true; } return false; }); return true;
      case 7: r.fields([&](int f, int) { if (f == 1) { h.nvals = (int32_t)r.zz(); return true; } if (f == 2) { h.enc = (int)r.zz(); return true; } return false; }); return true;
      case 8: { int16_t fid = 0; int t; while (r.field(fid, t)) { switch (fid) { case 1: h.nvals = (int32_t)r.zz(); break; case 2: h.num_nulls = (int32_t)r.zz(); break; case 4: h.enc = (int)r.zz(); break;
                 case 5: h.def_len = (int32_t)r.zz(); break; case 6: h.rep_len = (int32_t)r.zz(); break; case 7: h.v2_compressed = (t == 1); break; default: r.skip(t); } } return true; }
      default: return false;
    }
  });
  h.hdr_bytes = (int)(r.p - p);
  if (h.type < 0 || h.usize < 0 || h.csize < 0 || h.nvals < 0) fail(DFGPU_EXECUTION, "Parquet error: malformed page header");
  return h;
}

}  // namespace pq
}  // namespace dfgpu

using namespace dfgpu;
using namespace dfgpu::pq;

struct dfgpu_parquet {
  const uint8_t* host = nullptr; int64_t len = 0;
  void* map = nullptr; size_t map_len = 0;            // open_file: the mapping this handle owns
  const uint8_t* dev = nullptr; BufferPtr dev_owned;  // the file image in HBM (caller's, or staged by open_file)
  std::vector<Leaf> leaves; std::vector<RowGroup> rgs; int64_t num_rows = 0; std::string created_by;
  bool utf8_dictionary = true;
  bool registered = false;                            // open_file without device staging: the mapping is page-locked, so column chunks cross PCIe by DMA from where they lie
  hipEvent_t last_copy = nullptr;                     // recorded on the copy stream behind the last staged chunk of every read: DMA out of the mapping is over once it has fired
  std::mutex copy_mu;
  // the mapping must outlive every copy that reads it: wait for the last one before the pages are unlocked and unmapped (not left to hipHostUnregister's own waiting)
  ~dfgpu_parquet() { if (last_copy) { (void)hipEventSynchronize(last_copy); (void)hipEventDestroy(last_copy); } if (registered) (void)hipHostUnregister(map); if (map) munmap(map, map_len); }
};

namespace dfgpu {
namespace pq {

static void parse_footer(dfgpu_parquet* f) {
  if (f->len < 12 || memcmp(f->host, "PAR1", 4) != 0 || memcmp(f->host + f->len - 4, "PAR1", 4) != 0)
    fail(DFGPU_EXECUTION, "Parquet error: Invalid Parquet file. Corrupt footer");          // parquet crate file/footer.rs
  uint32_t mlen; memcpy(&mlen, f->host + f->len - 8, 4);
  if ((int64_t)mlen + 12 > f->len) fail(DFGPU_EXECUTION, "Parquet error: Invalid Parquet file. Reported metadata length of %u + 8 byte footer, but file is only %lld bytes", mlen, (long long)f->len);
  TR r{f->host + f->len - 8 - mlen, f->host + f->len - 8};
  std::vector<std::pair<Leaf, int>> elems;     // (element, num_children)
  r.fields([&](int id, int) {
    if (id == 2) {
      int et; int64_t n; r.list(et, n);
      for (int64_t i = 0; i < n; i++) {
        Leaf l; int nch = 0;
        r.fields([&](int fid, int) {
          switch (fid) {
            case 1: l.phys = (int)r.zz(); return true;
            case 2: l.type_len = (int)r.zz(); return true;
            case 3: l.rep = (int)r.zz(); return true;
            case 4: l.name = r.binary(); return true;
            case 5: nch = (int)r.zz(); return true;
            case 6: l.conv = (int)r.zz(); return true;
            case 7: l.scale = (int)r.zz(); return true;
            case 8: l.precision = (int)r.zz(); return true;
            case 10: parse_logical_type(r, l); return true;
            default: return false;
          }
        });
        elems.emplace_back(l, nch);
      }
      return true;
    }
    if (id == 3) { f->num_rows = r.zz(); return true; }
    if (id == 4) {
      int et; int64_t n; r.list(et, n);
      for (int64_t i = 0; i < n; i++) {
        RowGroup g;
        r.fields([&](int fid, int) {
          if (fid == 1) {
            int e2; int64_t nc; r.list(e2, nc);
            for (int64_t c = 0; c < nc; c++) { Chunk ch; r.fields([&](int cf, int) { if (cf == 3) { parse_column_meta(r, ch); return true; } return false; }); g.cols.push_back(ch); }
            return true;
          }
          if (fid == 3) { g.rows = r.zz(); return true; }
          return false;
        });
        f->rgs.push_back(std::move(g));
      }
      return true;
    }
    if (id == 6) { f->created_by = r.binary(); return true; }
    return false;
  });
  if (elems.empty()) fail(DFGPU_EXECUTION, "Parquet error: file metadata holds no schema");
  // depth-first schema: element 0 is the root; leaves in order are the column chunks of every row group
  std::vector<int> open; open.push_back(elems[0].second);
  for (size_t i = 1; i < elems.size(); i++) {
    while (!open.empty() && open.back() == 0) open.pop_back();
    if (open.empty()) fail(DFGPU_EXECUTION, "Parquet error: schema tree is malformed");
    open.back()--;
    Leaf l = elems[i].first; int nch = elems[i].second;
    if (nch > 0) { open.push_back(nch); continue; }
    l.depth = (int)open.size();
    resolve_arrow_type(l);
    f->leaves.push_back(l);
  }
  for (auto& g : f->rgs) if (g.cols.size() != f->leaves.size()) fail(DFGPU_EXECUTION, "Parquet error: row group with %zu column chunks, schema has %zu leaves", g.cols.size(), f->leaves.size());
}

// ================================================================================ device side
constexpr int PQ_NT = 256, PQ_TILE = 2048, PQ_MAXR = 256;
enum { MODE_FIXED = 0, MODE_KEYS = 1, MODE_STRING = 2 };
enum { CONV_COPY4 = 0, CONV_COPY8 = 1, CONV_4TO1 = 2, CONV_4TO2 = 3, CONV_I32_DEC = 4, CONV_I64_DEC = 5, CONV_FLBA_DEC = 6, CONV_BOOL = 7 };

struct PqPage {
  const uint8_t* data; const uint8_t* dict_data;     // uncompressed page payload; PLAIN values of the row group's dictionary page (fixed width)
  int64_t row_start;
  uint32_t size; int32_t num_values; int32_t dict_enc; int32_t lvl_mode /*0 none, 1 u32-length prefixed (v1), 2 lvl_len bytes (v2)*/; int32_t lvl_len; int32_t decode_levels;
  int32_t dict_base, dict_count;
  uint32_t str_base;                                  // PLAIN byte arrays: first slot of this page in PqCol::str_pos, its index in PqCol::str_cnt is the page's own
};
struct PqCol {
  int32_t mode, conv, wp, wo;
  void* out; uint8_t* vbytes; uint32_t* slen; uint64_t* ssrc; const int32_t* dict_offsets; const uint8_t* dict_chars;
  const uint32_t* str_pos; const uint32_t* str_cnt;   // PLAIN byte arrays: where each value's bytes start in its page, values found per page (k_pq_str_walk)
};

__device__ inline uint32_t ld32u(const uint8_t* p) {
  uintptr_t a = (uintptr_t)p; const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3); uint32_t sh = (uint32_t)(a & 3);
  uint32_t lo = q[0]; if (!sh) return lo;
  return __builtin_amdgcn_alignbyte(q[1], lo, sh);
}
__device__ inline uint64_t ld64u(const uint8_t* p) { return (uint64_t)ld32u(p) | ((uint64_t)ld32u(p + 4) << 32); }

__device__ inline void store_fixed(const PqCol& c, int64_t row, const uint8_t* src, bool valid) {
  switch (c.conv) {
    case CONV_COPY4: ((uint32_t*)c.out)[row] = valid ? ld32u(src) : 0u; break;
    case CONV_COPY8: ((uint64_t*)c.out)[row] = valid ? ld64u(src) : 0ull; break;
    case CONV_4TO1: ((uint8_t*)c.out)[row] = valid ? (uint8_t)ld32u(src) : (uint8_t)0; break;
    case CONV_4TO2: ((uint16_t*)c.out)[row] = valid ? (uint16_t)ld32u(src) : (uint16_t)0; break;
    case CONV_I32_DEC: { int64_t v = valid ? (int64_t)(int32_t)ld32u(src) : 0; ((uint64_t*)c.out)[2 * row] = (uint64_t)v; ((uint64_t*)c.out)[2 * row + 1] = (uint64_t)(v >> 63); break; }
    case CONV_I64_DEC: { int64_t v = valid ? (int64_t)ld64u(src) : 0; ((uint64_t*)c.out)[2 * row] = (uint64_t)v; ((uint64_t*)c.out)[2 * row + 1] = (uint64_t)(v >> 63); break; }
    case CONV_FLBA_DEC: {                              // big-endian two's complement of wp bytes
      uint64_t lo = 0, hi = 0;
      if (valid) {
        bool neg = src[0] & 0x80; lo = hi = neg ? ~0ull : 0ull;
        for (int b = 0; b < c.wp; b++) { hi = (hi << 8) | (lo >> 56); lo = (lo << 8) | src[b]; }
      }
      ((uint64_t*)c.out)[2 * row] = lo; ((uint64_t*)c.out)[2 * row + 1] = hi; break;
    }
    default: break;
  }
}

// ---- RLE / bit-packed hybrid: one lane parses a batch of runs (clipped to what the tile wants), every lane expands
struct RleState { const uint8_t* p; const uint8_t* end; const uint8_t* run_ptr; uint32_t run_left, run_val, run_done, run_packed, bw, bad; };
struct RunTable { uint32_t start[PQ_MAXR + 1]; uint32_t first[PQ_MAXR]; uint64_t src[PQ_MAXR]; uint32_t nr, filled; };

__device__ inline void rle_parse(RleState& s, uint32_t want, RunTable& t) {
  uint32_t nr = 0, filled = 0;
  while (filled < want && nr < (uint32_t)PQ_MAXR) {
    if (s.run_left == 0) {
      uint32_t h = 0; int sh = 0; bool ok = false;
      while (s.p < s.end && sh < 35) { uint8_t b = *s.p++; h |= (uint32_t)(b & 0x7f) << sh; sh += 7; if (!(b & 0x80)) { ok = true; break; } }
      if (!ok) { s.bad = 1; break; }
      if (h & 1) {
        uint32_t groups = h >> 1; uint64_t bytes = (uint64_t)groups * s.bw;
        if (groups == 0) { s.bad = 1; break; }
        if ((uint64_t)(s.end - s.p) < bytes) { uint64_t avail = (uint64_t)(s.end - s.p); groups = s.bw ? (uint32_t)(avail / s.bw) : groups; bytes = (uint64_t)groups * s.bw; if (!groups) { s.bad = 1; break; } }   // a last run may be cut short by the writer
        s.run_packed = 1; s.run_left = groups * 8; s.run_ptr = s.p; s.run_done = 0; s.p += bytes;
      } else {
        uint32_t cnt = h >> 1, nb = (s.bw + 7) >> 3, v = 0;
        if (cnt == 0 || (uint32_t)(s.end - s.p) < nb) { s.bad = 1; break; }
        for (uint32_t b = 0; b < nb; b++) v |= (uint32_t)s.p[b] << (8 * b);
        s.p += nb; s.run_packed = 0; s.run_left = cnt; s.run_val = v; s.run_done = 0;
      }
    }
    uint32_t take = min(s.run_left, want - filled);
    t.start[nr] = filled; t.first[nr] = s.run_packed ? s.run_done : 0xFFFFFFFFu; t.src[nr] = s.run_packed ? (uint64_t)(uintptr_t)s.run_ptr : (uint64_t)s.run_val;
    nr++; filled += take; s.run_left -= take; s.run_done += take;
  }
  t.start[nr] = filled; t.nr = nr; t.filled = filled;
}
// decode `want` values of the stream into dst[0..want); returns false when the stream ends early
__device__ inline bool rle_fill(RleState& s, RunTable& t, uint32_t want, uint32_t* dst) {
  uint32_t got = 0;
  while (got < want) {
    __syncthreads();
    if (threadIdx.x == 0) rle_parse(s, want - got, t);
    __syncthreads();
    uint32_t filled = t.filled, nr = t.nr, bw = s.bw;
    if (filled == 0) return false;
    uint32_t mask = bw >= 32 ? 0xFFFFFFFFu : ((1u << bw) - 1u);
    for (uint32_t i = threadIdx.x; i < filled; i += PQ_NT) {
      uint32_t lo = 0, hi = nr - 1;
      while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (t.start[mid] <= i) lo = mid; else hi = mid - 1; }
      uint32_t first = t.first[lo], v;
      if (first == 0xFFFFFFFFu) v = (uint32_t)t.src[lo];
      else {
        uint64_t bit = (uint64_t)(first + (i - t.start[lo])) * bw; const uint8_t* q = (const uint8_t*)(uintptr_t)t.src[lo] + (bit >> 3); uint32_t sh = (uint32_t)(bit & 7);
        uint32_t nb = (sh + bw + 7) >> 3; uint64_t w = 0;
        for (uint32_t b = 0; b < nb; b++) w |= (uint64_t)q[b] << (8 * b);
        v = (uint32_t)(w >> sh) & mask;
      }
      dst[got + i] = v;
    }
    got += filled;
  }
  __syncthreads();
  return true;
}

__global__ void __launch_bounds__(PQ_NT) k_pq_decode(const PqPage* __restrict__ pages, PqCol col, uint32_t* flags) {
  __shared__ RleState lv, ix; __shared__ RunTable rt;
  __shared__ uint32_t vals[PQ_TILE]; __shared__ uint16_t pos16[PQ_TILE];
  __shared__ uint32_t wstate[4];                        // [0] offset of the values in the page, [2] bad
  __shared__ uint32_t scan_lds[4]; __shared__ const uint8_t* s_vptr;
  const PqPage pg = pages[blockIdx.x];
  const int tid = threadIdx.x;
  if (tid == 0) {
    const uint8_t* p = pg.data; const uint8_t* end = pg.data + pg.size; uint32_t bad = 0;
    lv.run_left = ix.run_left = 0; lv.bad = ix.bad = 0; lv.bw = 1; ix.bw = 0;
    if (pg.lvl_mode == 1) {
      if (pg.size < 4) bad = 1; else { uint32_t L = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); if (L > pg.size - 4) bad = 1; else { lv.p = p + 4; lv.end = p + 4 + L; p += 4 + L; } }
    } else if (pg.lvl_mode == 2) { if ((uint32_t)pg.lvl_len > pg.size) bad = 1; else { lv.p = p; lv.end = p + pg.lvl_len; p += pg.lvl_len; } }
    if (!bad && pg.dict_enc == 1) { if (p >= end) { if (pg.num_values) bad = 1; } else { ix.bw = *p++; if (ix.bw > 32) bad = 1; } ix.p = p; ix.end = end; }
    if (!bad && pg.dict_enc == 2) { if (end - p < 4) bad = 1; else { p += 4; ix.bw = 1; ix.p = p; ix.end = end; } }        // RLE Boolean values: u32 length, then the hybrid runs at width 1
    s_vptr = p; wstate[0] = (uint32_t)(p - pg.data); wstate[2] = bad;
  }
  __syncthreads();
  if (wstate[2]) { if (tid == 0) atomicOr(flags, DFGPU_FLAG_OOB); return; }
  const uint8_t* vptr = s_vptr; const uint32_t vbytes_avail = pg.size - (uint32_t)(vptr - pg.data);
  uint32_t consumed = 0; bool bad = false;
  for (uint32_t row0 = 0; row0 < (uint32_t)pg.num_values && !bad; row0 += PQ_TILE) {
    const uint32_t tl = min((uint32_t)PQ_TILE, (uint32_t)pg.num_values - row0);
    uint32_t nn = tl;
    if (pg.decode_levels) {
      if (!rle_fill(lv, rt, tl, vals)) { bad = true; break; }
      uint32_t cnt = 0, f8 = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) { uint32_t idx = tid * 8 + j; if (idx < tl && vals[idx] == 1u) { f8 |= 1u << j; cnt++; } }
      uint32_t tot; uint32_t base = block_exclusive_sum<uint32_t>(cnt, scan_lds, &tot); nn = tot;
#pragma unroll
      for (int j = 0; j < 8; j++) { uint32_t idx = tid * 8 + j; if (idx < tl) pos16[idx] = (f8 >> j) & 1 ? (uint16_t)base++ : (uint16_t)0xFFFF; }
      __syncthreads();
    }
    if (pg.dict_enc) {
      if (nn && !rle_fill(ix, rt, nn, vals)) { bad = true; break; }
    } else if (col.mode == MODE_STRING) {
      // PLAIN byte arrays: k_pq_str_walk has found where every value starts; fewer values than the levels ask for = the stream ended early
      if ((uint64_t)consumed + nn > col.str_cnt[blockIdx.x]) { bad = true; break; }
    } else if (col.conv == CONV_BOOL) {
      if ((uint64_t)(consumed + nn + 7) / 8 > vbytes_avail) { bad = true; break; }
    } else if ((uint64_t)(consumed + nn) * (uint32_t)col.wp > vbytes_avail) { bad = true; break; }
    __syncthreads();
    for (uint32_t i = tid; i < tl; i += PQ_NT) {
      const int64_t row = pg.row_start + row0 + i;
      const uint32_t p16 = pg.decode_levels ? (uint32_t)pos16[i] : i; const bool valid = p16 != 0xFFFFu;
      if (col.vbytes) col.vbytes[row] = valid;
      uint32_t idx = 0;
      if (pg.dict_enc && valid) { idx = vals[p16]; if (pg.dict_enc == 1 && idx >= (uint32_t)pg.dict_count) { atomicOr(flags, DFGPU_FLAG_OOB); idx = 0; if (!pg.dict_count) continue; } }
      if (col.mode == MODE_KEYS) ((int32_t*)col.out)[row] = valid ? (int32_t)(idx + (uint32_t)pg.dict_base) : 0;
      else if (col.mode == MODE_FIXED) {
        if (col.conv == CONV_BOOL) { uint32_t q = consumed + p16; ((uint8_t*)col.out)[row] = !valid ? (uint8_t)0 : pg.dict_enc ? (uint8_t)(idx & 1) : (uint8_t)((vptr[q >> 3] >> (q & 7)) & 1); }      // PLAIN bits or RLE values; never a dictionary
        else store_fixed(col, row, pg.dict_enc ? pg.dict_data + (size_t)idx * col.wp : vptr + (size_t)(consumed + (valid ? p16 : 0)) * col.wp, valid);
      } else {
        uint32_t L = 0; uint64_t src = 0;
        if (valid) {
          if (pg.dict_enc) { int32_t k = (int32_t)idx + pg.dict_base; int32_t o = col.dict_offsets[k]; L = (uint32_t)(col.dict_offsets[k + 1] - o); src = (uint64_t)(uintptr_t)(col.dict_chars + o); }
          else { const uint32_t at = col.str_pos[pg.str_base + consumed + p16]; L = ld32u(pg.data + at - 4); src = (uint64_t)(uintptr_t)(pg.data + at); }
        }
        col.slen[row] = L; col.ssrc[row] = src;
      }
    }
    consumed += nn;
    __syncthreads();
  }
  if (bad || lv.bad || ix.bad) { if (tid == 0) atomicOr(flags, DFGPU_FLAG_OOB); }
}

// PLAIN byte arrays (u32 length + bytes, back to back) are a linked list: where value i starts is known only after value i - 1.  One lane walking a 1 MB page of short strings
// takes ~70 000 dependent loads.  Here the page goes through LDS in windows of PQ_WIN bytes (read once, coalesced); a window is cut into one segment per lane and every lane walks
// its own: lane 0 from the position the previous window ended on, the others from a GUESS -- the first position of the segment from which PQ_CHAIN headers in a row stay inside the
// page (text bytes read as a length point far outside; the byte before a header reads as a small length, which is why one or three in a row are not enough).  The guesses are then
// checked against the true path, all lanes at once: a walk stands if it started on the position its predecessor's walk ended on; a lane whose guess was wrong walks again from
// there, and its successor looks again.  So the result never depends on a guess being right; only the time does.
// Counts are scanned and the lanes walk once more, out of LDS, to write where each value's bytes start.
// Tried on the way: the same scheme straight from global memory with 256 segments per 1 MB page (1.85 ms for 36 M values: every hop re-fetches its 128-byte line, the lanes' lines
// do not fit L1) and with 1024 (3.0 ms: they do not fit L2 either).
constexpr int PQ_WIN = 65536, PQ_SUB = PQ_WIN / PQ_NT, PQ_CHAIN = 8; constexpr uint32_t PQ_NOPOS = 0xFFFFFFFFu;
static_assert(PQ_SUB >= 64 && PQ_SUB * PQ_NT == PQ_WIN, "one segment per lane");
// word w of the window lives at LDS word w + w / 64: the lanes' segments are 64 words apart, which unpadded is one bank for the whole wave
__device__ inline uint32_t pq_win_at(uint32_t w) { return w + (w >> 6); }
// a length prefix at page offset p, read from the window (win holds the page's bytes from offset wb on, whole words, one word more than the window)
__device__ inline bool pq_str_hdr(const uint32_t* win, uint32_t wb, uint32_t wend, uint32_t size, uint32_t p, uint32_t& next) {
  if (p >= wend || (uint64_t)p + 4 > size) return false;             // wend: headers from here on belong to the next window
  const uint32_t r = p - wb, w = r >> 2, lo = win[pq_win_at(w)], sh = r & 3;
  const uint32_t L = sh ? __builtin_amdgcn_alignbyte(win[pq_win_at(w + 1)], lo, sh) : lo;
  if ((uint64_t)p + 4 + L > size) return false;
  next = p + 4 + L; return true;
}
__global__ void __launch_bounds__(PQ_NT) k_pq_str_walk(const PqPage* __restrict__ pages, uint32_t* __restrict__ str_pos, uint32_t* __restrict__ str_cnt) {
  __shared__ __attribute__((aligned(16))) uint32_t win[PQ_WIN / 4 + PQ_WIN / 256 + 8];
  __shared__ uint32_t s_exit[PQ_NT], s_wave[PQ_NT / WAVE], s_next; __shared__ uint8_t s_stale[PQ_NT];
  const PqPage pg = pages[blockIdx.x]; const int t = threadIdx.x;
  uint32_t vstart = 0; bool bad = false;
  if (pg.lvl_mode == 1) { if (pg.size < 4) bad = true; else { const uint32_t L = ld32u(pg.data); if (L > pg.size - 4) bad = true; else vstart = 4 + L; } }
  else if (pg.lvl_mode == 2) { if ((uint32_t)pg.lvl_len > pg.size) bad = true; else vstart = (uint32_t)pg.lvl_len; }
  if (bad || pg.dict_enc || pg.num_values <= 0) { if (t == 0) str_cnt[blockIdx.x] = 0; return; }        // block-uniform; k_pq_decode reports a bad page itself
  // positions below are byte offsets from `words`, the word at or below the page's first byte (a page need not start on a word): page offset + mis
  const uint32_t* words = (const uint32_t*)((uintptr_t)pg.data & ~(uintptr_t)3); const uint32_t mis = (uint32_t)((uintptr_t)pg.data & 3);
  const uint32_t size = pg.size + mis, want = (uint32_t)pg.num_values;
  uint32_t cur = vstart + mis, found = 0;                                                                 // block-uniform: the true path's position, values written so far
  while (cur < size && found < want) {
    const uint32_t wb = cur & ~3u, wend = (uint32_t)min((uint64_t)wb + PQ_WIN, (uint64_t)size);             // window [wb, wend)
    const uint32_t nwords = (wend - wb + 3) / 4 + 1;                                                       // one more: a header may straddle the last word (buffers are padded by 16 bytes)
    __syncthreads();
    for (uint32_t i = t; i < nwords; i += PQ_NT) win[pq_win_at(i)] = words[(wb >> 2) + i];
    const uint32_t s0 = max(cur, wb + (uint32_t)t * PQ_SUB), s1 = min(wend, wb + (uint32_t)(t + 1) * PQ_SUB);
    __syncthreads();
    uint32_t entry = PQ_NOPOS, exitp = PQ_NOPOS, cnt = 0;                                                 // exit NOPOS: the list ends inside the segment
    if (s0 < s1) {
      if (t == 0) entry = cur;
      else for (uint32_t p = s0; p < s1; p++) {
        // the chain may leave the window (one long value does), and then goes on in global memory: accepting it for what it saw inside is not enough -- five-byte values
        // ("A", "N", "R" with their prefixes) read one byte early as a length of 0x14100 + 4, a multiple of five, which lands on the same wrong phase 82 KB further on
        uint32_t q = p, nx; int h = 0;
        while (h < PQ_CHAIN && q != size) {
          if (q < wend) { if (!pq_str_hdr(win, wb, wend, size, q, nx)) break; }
          else { if ((uint64_t)q + 4 > size) break; const uint32_t L = ld32u((const uint8_t*)words + q); if ((uint64_t)q + 4 + L > size) break; nx = q + 4 + L; }
          q = nx; h++;
        }
        if (h == PQ_CHAIN || (h > 0 && q == size)) { entry = p; break; }
      }
      if (entry != PQ_NOPOS) { uint32_t c = entry, nx; while (c < s1 && pq_str_hdr(win, wb, wend, size, c, nx)) { cnt++; c = nx; } if (c >= s1) exitp = c; }
    }
    s_exit[t] = exitp;
    __syncthreads();
    // Repair, all lanes at once: a lane's walk stands if it started where its predecessor's walk arrives.  One that did not walks again from there (or lets a value that jumps
    // over its whole segment pass through) -- but only when the predecessor itself stands: starting from a wrong walk's exit would carry the error down the window, one
    // segment per round (tried: 11 ms).  The first lane that does not stand always has a standing predecessor, so every round settles at least one more segment; isolated
    // wrong guesses, the usual case, all settle in the first.
    const int tl = (int)((wend - 1 - wb) / PQ_SUB);                                                       // the last segment with bytes
    uint32_t from = t == 0 ? cur : entry != PQ_NOPOS ? entry : 0xFFFFFFFEu;                               // the arrival this lane's (entry, cnt, exit) were computed for
    for (;;) {
      const uint32_t arrive = t == 0 ? cur : s_exit[t - 1];
      const bool stale = t > 0 && t <= tl && from != arrive;
      s_stale[t] = stale ? 1 : 0;
      if (!__syncthreads_or(stale ? 1 : 0)) break;
      if (stale && !s_stale[t - 1]) {
        from = arrive; entry = PQ_NOPOS; cnt = 0; exitp = arrive;                                        // the list is over (NOPOS), or jumps over this segment
        if (arrive < s1) { entry = arrive; exitp = PQ_NOPOS; uint32_t c = arrive, nx; while (c < s1 && pq_str_hdr(win, wb, wend, size, c, nx)) { cnt++; c = nx; } if (c >= s1) exitp = c; }
        s_exit[t] = exitp;
      }
      __syncthreads();
    }
    if (t == tl) s_next = exitp;
    const uint32_t inc = wave_inclusive_sum(cnt);
    if (lane_id() == WAVE - 1) s_wave[t >> 6] = inc;
    __syncthreads();
    uint32_t base = found + inc - cnt, total = 0;
#pragma unroll
    for (int w = 0; w < PQ_NT / WAVE; w++) { const uint32_t x = s_wave[w]; if (w < (t >> 6)) base += x; total += x; }
    if (entry != PQ_NOPOS) {
      uint32_t c = entry, nx;
      for (uint32_t j = 0; j < cnt && base + j < want; j++) { pq_str_hdr(win, wb, wend, size, c, nx); str_pos[pg.str_base + base + j] = c + 4 - mis; c = nx; }      // a page holds at most num_values values
    }
    found += total;
    const uint32_t nxt = s_next;
    if (nxt == PQ_NOPOS || total == 0) break;                                                             // the list ended (a length that leaves the page, or fewer than 4 bytes left)
    cur = nxt;
  }
  if (t == 0) str_cnt[blockIdx.x] = min(found, want);
}

// PLAIN fixed-width pages without levels: (page, slice of 8192 values)
constexpr int PQ_SLICE = 8192;
struct PqSlice { uint32_t page; uint32_t first; };
__global__ void __launch_bounds__(PQ_NT) k_pq_plain(const PqPage* __restrict__ pages, const PqSlice* __restrict__ slices, PqCol col, uint32_t* flags) {
  const PqSlice sl = slices[blockIdx.x]; const PqPage pg = pages[sl.page];
  const uint8_t* vptr = pg.data + (pg.lvl_mode == 2 ? pg.lvl_len : 0);
  if (pg.lvl_mode == 1) { uint32_t L = ld32u(pg.data); vptr = pg.data + 4 + L; if ((uint64_t)L + 4 > pg.size) { if (threadIdx.x == 0) atomicOr(flags, DFGPU_FLAG_OOB); return; } }
  uint32_t avail = pg.size - (uint32_t)(vptr - pg.data);
  if (col.conv == CONV_BOOL ? ((uint64_t)pg.num_values + 7) / 8 > avail : (uint64_t)pg.num_values * (uint32_t)col.wp > avail) { if (threadIdx.x == 0) atomicOr(flags, DFGPU_FLAG_OOB); return; }
  uint32_t last = min(sl.first + (uint32_t)PQ_SLICE, (uint32_t)pg.num_values);
  for (uint32_t i = sl.first + threadIdx.x; i < last; i += PQ_NT) {
    int64_t row = pg.row_start + i;
    if (col.vbytes) col.vbytes[row] = 1;
    if (col.conv == CONV_BOOL) ((uint8_t*)col.out)[row] = (vptr[i >> 3] >> (i & 7)) & 1;
    else store_fixed(col, row, vptr + (size_t)i * col.wp, true);
  }
}

// string bytes: 8 lanes per row
__global__ void __launch_bounds__(BLOCK) k_pq_chars(const uint64_t* __restrict__ ssrc, const int32_t* __restrict__ offsets, int64_t n, uint8_t* __restrict__ out) {
  int64_t row = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 3; int sub = threadIdx.x & 7;
  if (row >= n) return;
  int32_t o = offsets[row], L = offsets[row + 1] - o; const uint8_t* s = (const uint8_t*)(uintptr_t)ssrc[row];
  for (int32_t b = sub; b < L; b += 8) out[o + b] = s[b];
}
__global__ void k_pq_set_i32(int32_t* p, int32_t v) { *p = v; }
__global__ void __launch_bounds__(BLOCK) k_pq_bytes_to_bits(const uint8_t* in, int64_t n, uint64_t* bits) {
  int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; uint64_t m = ballot64(i < n && in[i] != 0);
  if (lane_id() == 0 && (i >> 6) < ((n + 63) >> 6)) bits[i >> 6] = m;
}

// ---- Snappy (raw format): one wave per page.  The element chain is inherently serial (a tag's position follows from the tag before it, a copy may
// read what the element before it wrote), so the loop keeps that chain short: tags are decoded on the scalar unit from an LDS window of the input, the
// next tag's bytes are requested before the current element's bytes move, output goes to an LDS ring that holds the last 64 KB (every reference of
// the standard 64 KB-block compressor resolves there) and is flushed to HBM in 16-byte stores.  One wave per workgroup: LDS operations of a wave
// execute in order, no barrier is needed between an element's write and the next element's read.
constexpr int SN_RING = 32768, SN_WIN = 8192, SN_FLUSH = 8192;        // ring + window = 40 KB: four waves per CU; references further back than the ring are read from HBM
struct SnJob { const uint8_t* src; uint8_t* dst; uint32_t csize, usize; int32_t raw; int32_t codec; };

__device__ inline void sn_order() { __builtin_amdgcn_wave_barrier(); }          // LDS operations of one wave execute in issue order: only the compiler must not move them across
__device__ inline uint64_t sn_peek(const uint32_t* win, uint32_t rel) { uint32_t a = rel >> 2, sh = (rel & 3) * 8; return ((uint64_t)win[a] | ((uint64_t)win[a + 1] << 32)) >> sh; }   // >= 5 bytes at win + rel

// Block mode (blks != nullptr): the standard compressor works on 64 KB fragments of the input one at a time (a fresh hash table per fragment), so elements
// never straddle a multiple of 64 KB of OUTPUT and copies never reach before it: the 64 KB blocks of a page decode independently.  Workgroup b takes
// block blks[b].blk of page blks[b].job: it first skips the tags of the blocks before its own (tags only, out of a 512-byte register window read with
// v_readlane -- no bytes move), then decodes its block.  A page that breaks either assumption is marked in page_bad and decoded front to back by the
// repair launch (blks == nullptr), which also is where corrupt input raises the error flag.
struct SnBlk { uint32_t job, blk; };
__global__ void __launch_bounds__(64) k_pq_snappy(const SnJob* __restrict__ jobs, const SnBlk* __restrict__ blks, uint32_t* page_bad, uint32_t* flags) {
  __shared__ __attribute__((aligned(16))) uint8_t ring[SN_RING];
  __shared__ __attribute__((aligned(16))) uint32_t win[SN_WIN / 4 + 4];
  const uint32_t job = blks ? blks[blockIdx.x].job : blockIdx.x, blk = blks ? blks[blockIdx.x].blk : 0u;
  if (!blks && !page_bad[job]) return;
  SnJob jb = jobs[job]; const uint32_t lane = threadIdx.x;
  if (jb.raw) { for (uint32_t i = lane; i < jb.usize; i += 64) jb.dst[i] = jb.src[i]; return; }
  uint32_t pin = 0, ulen = 0; { int sh = 0; bool ok = false; while (pin < jb.csize && sh < 35) { uint8_t b = jb.src[pin++]; ulen |= (uint32_t)(b & 0x7f) << sh; sh += 7; if (!(b & 0x80)) { ok = true; break; } } if (!ok) ulen = 0xFFFFFFFFu; }
  // every cursor below is wave-uniform; sn_u() pins it to the scalar unit, so the element loop branches on SCC instead of masking lanes
#define sn_u(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
  auto give_up = [&]() { if (lane == 0) { if (blks) page_bad[job] = 1; else atomicOr(flags, DFGPU_FLAG_OOB); } };
  if (ulen != jb.usize) { give_up(); return; }
  pin = sn_u(pin);
  const uint32_t csize = sn_u(jb.csize);
  uint32_t usize = sn_u(jb.usize);
  if (blks) {
    const uint32_t target = blk * 65536u;
    if (target) {           // skip the elements of the blocks in front
      const uint32_t A = (uint32_t)((uintptr_t)jb.src & 3), vend = csize + A; const uint32_t* base32 = (const uint32_t*)(jb.src - A);
      auto loadw = [&](uint32_t wv) -> uint32_t { uint32_t o = wv + 4 * lane; return o < vend ? base32[o >> 2] : 0u; };
      uint32_t v = pin + A, wv = v & ~255u, cur = loadw(wv), nxt = loadw(wv + 256), out = 0; bool bad0 = false;
      // element (output length, bytes to the next tag) if a tag started at the low byte of t (>= 5 bytes of input); len 0 = malformed
      auto tagdec = [](uint64_t t, uint32_t& len, uint32_t& step) {
        const uint32_t lo = (uint32_t)t, b0 = lo & 0xff, kind = lo & 3;
        if (kind == 0) { len = (b0 >> 2) + 1; step = 1; if (len > 60) { uint32_t nb = len - 60; uint32_t ext = (uint32_t)(t >> 8); len = (nb == 4 ? ext : (ext & ((1u << (8 * nb)) - 1u))) + 1; step = 1 + nb; } step += len; }
        else if (kind == 1) { len = 4 + ((b0 >> 2) & 7); step = 2; } else { len = 1 + (b0 >> 2); step = kind == 2 ? 3 : 5; }
      };
      // Tags are skipped a 256-byte chunk at a time by pointer jumping: every byte position of the chunk is decoded as if a tag started there (4 positions per
      // lane), then 8 doubling rounds over (next position, output bytes, overshoot past the chunk) give, for the true entry position, the chunk's output and
      // where the next chunk is entered.  Only the chunk in which the output reaches the block boundary is walked tag by tag.
      __shared__ uint16_t pjN[260]; __shared__ uint32_t pjO[260], pjX[260];
      while (out < target) {
        if (v >= vend) { bad0 = true; break; }
        if (v - wv >= 256u) { if (v - wv < 512u) { cur = nxt; wv = sn_u(wv + 256u); } else { wv = sn_u(v & ~255u); cur = loadw(wv); } nxt = loadw(wv + 256u); }
        {
          const uint32_t up = (uint32_t)__shfl_down((int)cur, 1, 64), hi = lane == 63 ? (uint32_t)__builtin_amdgcn_readlane((int)nxt, 0) : up;
          const uint64_t w = ((uint64_t)hi << 32) | cur;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            uint32_t len, step; tagdec(w >> (8 * j), len, step);
            const uint32_t p = 4 * lane + j; uint32_t np = p + step; if (len == 0 || step == 0) np = 0x7FFFFFFFu;        // malformed: runs out of the stream, caught by v >= vend
            pjN[p] = (uint16_t)(np < 256u ? np : 256u); pjO[p] = len; pjX[p] = np < 256u ? 0u : np - 256u;
          }
          if (lane == 0) { pjN[256] = 256; pjO[256] = 0; pjX[256] = 0; }
          sn_order();
#pragma unroll 1
          for (int r = 0; r < 8; r++) {
            uint32_t n[4], o[4], x[4], nn[4];
#pragma unroll
            for (int j = 0; j < 4; j++) n[j] = pjN[4 * lane + j];
#pragma unroll
            for (int j = 0; j < 4; j++) { o[j] = pjO[n[j]]; x[j] = pjX[n[j]]; nn[j] = pjN[n[j]]; }
            sn_order();
#pragma unroll
            for (int j = 0; j < 4; j++) if (n[j] < 256u) { const uint32_t p = 4 * lane + j; pjO[p] += o[j]; pjX[p] = x[j]; pjN[p] = (uint16_t)nn[j]; }
            sn_order();
          }
          const uint32_t e = v - wv, co = sn_u(pjO[e]), cx = sn_u(pjX[e]), cn = sn_u(pjN[e]);
          sn_order();
          if (cn == 256u && co <= target - out && cx < 0x40000000u) { out = sn_u(out + co); v = sn_u(wv + 256u + cx); continue; }
        }
        while (out < target && v - wv < 256u && v < vend) {          // the chunk that holds the boundary (or one the jump could not finish): tag by tag
          const uint32_t idx = (v - wv) >> 2, sh = (v & 3u) * 8u;
          const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)idx), a1 = idx == 63u ? (uint32_t)__builtin_amdgcn_readlane((int)nxt, 0) : (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(idx + 1u));
          uint32_t len, step; tagdec((((uint64_t)a1 << 32) | a0) >> sh, len, step);
          if (len == 0 || len > target - out || step > vend - v) { bad0 = true; break; }          // an element across the 64 KB boundary: not the standard block structure
          out = sn_u(out + len); v = sn_u(v + step);
        }
        if (bad0) break;
      }
      if (bad0) { give_up(); return; }
      pin = sn_u(v - A);
    }
    if (target >= usize) { give_up(); return; }
    usize = sn_u(min(65536u, usize - target)); jb.dst += target;
  }
  // ---- decode: the input is taken a 256-byte chunk at a time.  Pointer jumping (as in the skip above, with the jump table of every doubling level kept) finds which
  // byte positions of the chunk start an element and where each element's output goes; the chunk's elements are then listed in order.  Literals of up to 64
  // bytes do not depend on earlier output: every lane moves one of them (input window -> ring).  Only copies (and long literals) run one after the other, all
  // lanes moving one element's bytes.
  uint32_t pout = 0, flushed = 0, wbv = 0xFFFFFFFFu, wlen = 0; bool bad = false;
  const bool dst16 = ((uintptr_t)jb.dst & 15) == 0;
  const uint32_t A = (uint32_t)((uintptr_t)jb.src & 3), vend = csize + A; const uint32_t* base32 = (const uint32_t*)(jb.src - A);
  const uint8_t* winb = (const uint8_t*)win;
  __shared__ uint16_t dN[260], dL[8][256], dTS[136]; __shared__ uint32_t dO[260], dX[260]; __shared__ uint8_t dR[264];
  struct SnSym { uint32_t opos, len, off, src; };
  __shared__ SnSym dT[136];
  auto flush = [&](uint32_t upto) {            // ring -> dst for [flushed, upto)
    uint32_t a = flushed; flushed = sn_u(upto);
    if (!dst16) { for (uint32_t o = a + lane; o < upto; o += 64) jb.dst[o] = ring[o & (SN_RING - 1)]; return; }
    uint32_t head = min(upto, (a + 15u) & ~15u);
    for (uint32_t o = a + lane; o < head; o += 64) jb.dst[o] = ring[o & (SN_RING - 1)];
    uint32_t n16 = (upto - head) >> 4;
    for (uint32_t i = lane; i < n16; i += 64) { uint32_t o = head + (i << 4); *(uint4*)(jb.dst + o) = *(const uint4*)(ring + (o & (SN_RING - 1))); }
    for (uint32_t o = head + (n16 << 4) + lane; o < upto; o += 64) jb.dst[o] = ring[o & (SN_RING - 1)];
  };
  uint32_t v = pin + A;
  while (pout < usize) {
    if (v >= vend) { bad = true; break; }
    const uint32_t wv = sn_u(v & ~255u);
    if (!(wv >= wbv && (wv + 384u <= wbv + wlen || wbv + wlen >= vend))) {       // LDS window of the input: the chunk and the 128 bytes behind it (short literal bodies)
      sn_order();
      wbv = wv; wlen = sn_u(min((uint32_t)SN_WIN, ((vend - wbv) + 3u) & ~3u));
      const uint32_t nw = wlen >> 2;
      for (uint32_t i = lane; i < nw; i += 64) win[i] = base32[(wbv >> 2) + i];
      for (uint32_t i = nw + lane; i < nw + 4; i += 64) win[i] = 0;
      sn_order();
    }
    const uint32_t wi = (wv - wbv) >> 2, nwv = wlen >> 2;
    const uint32_t cur = wi + lane < nwv ? win[wi + lane] : 0u, nx0 = wi + 64 < nwv ? win[wi + 64] : 0u;
    const uint32_t up = (uint32_t)__shfl_down((int)cur, 1, 64);
    const uint64_t w = ((uint64_t)(lane == 63 ? nx0 : up) << 32) | cur;
    uint32_t len4[4], step4[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint64_t t = w >> (8 * j); const uint32_t lo = (uint32_t)t, b0 = lo & 0xff, kind = lo & 3; uint32_t len, step;
      if (kind == 0) { len = (b0 >> 2) + 1; step = 1; if (len > 60) { uint32_t nb = len - 60; uint32_t ext = (uint32_t)(t >> 8); len = (nb == 4 ? ext : (ext & ((1u << (8 * nb)) - 1u))) + 1; step = 1 + nb; } step += len; }
      else if (kind == 1) { len = 4 + ((b0 >> 2) & 7); step = 2; } else { len = 1 + (b0 >> 2); step = kind == 2 ? 3 : 5; }
      len4[j] = len; step4[j] = step;
      const uint32_t p = 4 * lane + j; uint32_t np = p + step; if (len == 0 || (kind == 0 && step < len) || step > 0x3FFFFFFFu) np = 0x7FFFFFFFu;        // malformed / absurd lengths: leave the stream, caught below
      dN[p] = (uint16_t)(np < 256u ? np : 256u); dO[p] = len; dX[p] = np < 256u ? 0u : np - 256u; dR[p] = 0;
    }
    if (lane == 0) { dN[256] = 256; dO[256] = 0; dX[256] = 0; }
    sn_order();
#pragma unroll 1
    for (int r = 0; r < 8; r++) {
      uint32_t n[4], o[4], x[4], nn[4];
#pragma unroll
      for (int j = 0; j < 4; j++) { n[j] = dN[4 * lane + j]; dL[r][4 * lane + j] = (uint16_t)n[j]; }
#pragma unroll
      for (int j = 0; j < 4; j++) { o[j] = dO[n[j]]; x[j] = dX[n[j]]; nn[j] = dN[n[j]]; }
      sn_order();
#pragma unroll
      for (int j = 0; j < 4; j++) if (n[j] < 256u) { const uint32_t p = 4 * lane + j; dO[p] += o[j]; dX[p] = x[j]; dN[p] = (uint16_t)nn[j]; }
      sn_order();
    }
    const uint32_t e = v - wv, c_out = sn_u(dO[e]), c_x = sn_u(dX[e]), c_n = sn_u(dN[e]);
    if (c_n != 256u || c_x >= 0x40000000u) { bad = true; break; }
    // positions on the path from e: top-down over the doubling levels (a node at distance d is reached through the set bits of d, high to low)
    if (lane == 0) dR[e] = 1;
    sn_order();
#pragma unroll 1
    for (int r = 7; r >= 0; r--) {
      uint32_t rp[4], t[4];
#pragma unroll
      for (int j = 0; j < 4; j++) { rp[j] = dR[4 * lane + j]; t[j] = dL[r][4 * lane + j]; }
      sn_order();
#pragma unroll
      for (int j = 0; j < 4; j++) if (rp[j] && t[j] < 256u) dR[t[j]] = 1;
      sn_order();
    }
    // the chunk's elements in order: output position, length, copy offset (0 = literal) and literal source; those past the end of the block are left out
    uint32_t cnt = 0, scnt = 0, flag = 0, sflag = 0; SnSym sy[4]; bool mal = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t p = 4 * lane + j;
      if (!dR[p]) continue;
      const uint64_t t = w >> (8 * j); const uint32_t lo = (uint32_t)t, b0 = lo & 0xff, kind = lo & 3;
      const uint32_t opos = pout + (c_out - dO[p]), len = len4[j];
      if (opos >= usize) continue;
      uint32_t off = 0;
      if (kind == 1) off = ((b0 >> 5) << 8) | ((lo >> 8) & 0xff); else if (kind == 2) off = (lo >> 8) & 0xffff; else if (kind == 3) off = (uint32_t)(t >> 8);
      const uint32_t ipos = wv + p;          // virtual input position of the tag
      if (len > usize - opos || ipos >= vend || step4[j] > vend - ipos || (kind != 0 && (off == 0 || off > opos))) { mal = true; continue; }
      sy[j] = SnSym{opos, len, off, kind == 0 ? ipos + (step4[j] - len) : 0u};       // literal: first byte of the body (virtual position)
      flag |= 1u << j; cnt++;
      if (kind != 0 || len > 64u) { sflag |= 1u << j; scnt++; }
    }
    if (ballot64(mal)) { bad = true; break; }
    const uint32_t kinc = wave_inclusive_sum(cnt), sinc = wave_inclusive_sum(scnt);
    const uint32_t K = (uint32_t)__builtin_amdgcn_readlane((int)kinc, 63), KS = (uint32_t)__builtin_amdgcn_readlane((int)sinc, 63);
    if (K > 136u) { bad = true; break; }
    { uint32_t k = kinc - cnt, ks = sinc - scnt;
#pragma unroll
      for (int j = 0; j < 4; j++) if ((flag >> j) & 1) { dT[k] = sy[j]; if ((sflag >> j) & 1) dTS[ks++] = (uint16_t)k; k++; } }
    sn_order();
    const uint32_t chunk_end = min(pout + c_out, usize);
    // short literals: one lane each, input window -> ring
    for (uint32_t k = lane; k < K; k += 64) {
      const SnSym q = dT[k];
      if (q.off == 0 && q.len <= 64u) { const uint32_t bi = q.src - wbv; for (uint32_t b = 0; b < q.len; b++) ring[(q.opos + b) & (SN_RING - 1)] = winb[bi + b]; }
    }
    sn_order();
    // copies and long literals, in order
    for (uint32_t ks = 0; ks < KS && !bad; ks++) {
      const uint32_t k = dTS[ks];
      const uint32_t opos = sn_u(dT[k].opos), len = sn_u(dT[k].len), off = sn_u(dT[k].off), src = sn_u(dT[k].src);
      if (off == 0) {               // long literal: from HBM in pieces, flushing as the ring fills
        uint32_t from = src - A, left = len, at = opos;
        while (left) {
          const uint32_t piece = min(left, (uint32_t)SN_FLUSH);
          if (at + piece - flushed > (uint32_t)(SN_RING - 64)) { flush(at); sn_order(); }
          for (uint32_t i = lane; i < piece; i += 64) ring[(at + i) & (SN_RING - 1)] = jb.src[from + i];
          at = sn_u(at + piece); from = sn_u(from + piece); left = sn_u(left - piece);
          sn_order();
        }
        continue;
      }
      uint32_t rel = lane;
      if (off < len) { uint32_t q = (uint32_t)((float)lane * __frcp_rn((float)off)); int32_t r = (int32_t)lane - (int32_t)(q * off); rel = r < 0 ? (uint32_t)(r + (int32_t)off) : ((uint32_t)r >= off ? (uint32_t)r - off : (uint32_t)r); }
      uint8_t bt = 0;
      if (off + (chunk_end - opos) <= (uint32_t)(SN_RING - 64)) { if (lane < len) bt = ring[(opos - off + rel) & (SN_RING - 1)]; }        // the source is still in the ring even with this chunk's literals written ahead
      else { flush(opos); sn_order(); __threadfence_block(); if (lane < len) bt = jb.dst[opos - off + rel]; }
      if (lane < len) ring[(opos + lane) & (SN_RING - 1)] = bt;
      sn_order();
    }
    if (bad) break;
    pout = sn_u(chunk_end); v = sn_u(wv + 256u + c_x);
    if (pout - flushed >= (uint32_t)SN_FLUSH) { flush(pout & ~15u); sn_order(); }
  }
  if (bad || pout != usize) { give_up(); return; }
  sn_order();
  flush(pout);
#undef sn_u
}

// ================================================================================ host orchestration
struct ColPlan {
  int leaf; int32_t out_type; bool as_dict;
  std::vector<PqPage> pages;          // data pages, device pointers filled
  std::vector<PqPage> dict_pages;     // string dictionaries as pages of a REQUIRED PLAIN string column
  bool any_levels = false, all_dict = true, any_dict = false;
};

template <typename T> static BufferPtr upload(dfgpu_ctx* ctx, const std::vector<T>& v) {
  BufferPtr b = alloc_buffer(ctx, std::max<size_t>(v.size() * sizeof(T), 16));
  if (!v.empty()) HIP_CHECK(hipMemcpyAsync(b->ptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return b;
}

// decode PLAIN / dictionary string pages into a Utf8 array (values + int32 offsets [+ validity bytes])
struct StrOut { BufferPtr offsets, chars; int64_t chars_bytes = 0; };
static StrOut finish_strings(dfgpu_ctx* ctx, int64_t n, BufferPtr offsets, BufferPtr ssrc) {
  StrOut o; o.offsets = offsets;
  uint64_t* d_total = ctx->d_scratch64 + 40;
  exclusive_scan_u32_inplace32(ctx, (uint32_t*)offsets->ptr, n, d_total);
  fetch_to_pinned(ctx, 40, d_total, 8); ctx->count_sync("parquet string bytes");
  uint64_t total = ctx->h_pinned[40];
  if (total > 0x7FFFFFFFull) fail(DFGPU_EXECUTION, "Parquet error: a Utf8 column of one read holds %llu bytes, more than int32 offsets address -- read fewer row groups per call", (unsigned long long)total);
  hipLaunchKernelGGL(k_pq_set_i32, dim3(1), dim3(1), 0, ctx->stream, (int32_t*)offsets->ptr + n, (int32_t)total);
  o.chars = alloc_buffer(ctx, std::max<size_t>((size_t)total, 16)); o.chars_bytes = (int64_t)total;
  if (n && total) { KernelTimer kt(ctx, "pq_chars"); hipLaunchKernelGGL(k_pq_chars, dim3(grid_for(n * 8, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint64_t*)ssrc->ptr, (const int32_t*)offsets->ptr, n, (uint8_t*)o.chars->ptr); }
  KERNEL_CHECK();
  return o;
}

static void launch_decode(dfgpu_ctx* ctx, const std::vector<PqPage>& pages, const PqCol& col, bool wide_ok) {
  if (pages.empty()) return;
  if (col.mode == MODE_STRING) {                 // every page goes through k_pq_decode; the PLAIN ones get their values' positions from k_pq_str_walk first
    std::vector<PqPage> pg(pages); uint64_t slots = 0; bool any_plain = false;
    for (auto& p : pg) { p.str_base = (uint32_t)slots; if (!p.dict_enc) { slots += (uint64_t)std::max(p.num_values, 0); any_plain = true; } }
    if (slots > 0xFFFFFFF0ull) fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: more than 2^32 PLAIN byte-array values in one Parquet read");
    BufferPtr dpg = upload(ctx, pg), pos, cnt; PqCol c2 = col;
    if (any_plain) {
      pos = alloc_buffer(ctx, (size_t)slots * 4 + 16); cnt = alloc_buffer(ctx, pg.size() * 4 + 16);
      KernelTimer kt(ctx, "pq_str_walk");
      hipLaunchKernelGGL(k_pq_str_walk, dim3((unsigned)pg.size()), dim3(PQ_NT), 0, ctx->stream, (const PqPage*)dpg->ptr, (uint32_t*)pos->ptr, (uint32_t*)cnt->ptr); KERNEL_CHECK();
      c2.str_pos = (const uint32_t*)pos->ptr; c2.str_cnt = (const uint32_t*)cnt->ptr;
    }
    KernelTimer kt(ctx, "pq_decode");
    hipLaunchKernelGGL(k_pq_decode, dim3((unsigned)pg.size()), dim3(PQ_NT), 0, ctx->stream, (const PqPage*)dpg->ptr, c2, ctx->d_flags); KERNEL_CHECK();
    return;                                       // pos / cnt go back to the stream-ordered cache behind the kernel
  }
  BufferPtr dpages = upload(ctx, pages);
  std::vector<PqPage> slow; std::vector<PqSlice> slices; std::vector<uint32_t> slow_idx;
  for (size_t i = 0; i < pages.size(); i++) {
    const PqPage& p = pages[i];
    if (wide_ok && !p.dict_enc && !p.decode_levels && col.mode == MODE_FIXED) { for (uint32_t f = 0; f < (uint32_t)p.num_values; f += PQ_SLICE) slices.push_back({(uint32_t)i, f}); }
    else slow.push_back(p);
  }
  if (!slices.empty()) {
    BufferPtr ds = upload(ctx, slices); KernelTimer kt(ctx, "pq_plain");
    hipLaunchKernelGGL(k_pq_plain, dim3((unsigned)slices.size()), dim3(PQ_NT), 0, ctx->stream, (const PqPage*)dpages->ptr, (const PqSlice*)ds->ptr, col, ctx->d_flags); KERNEL_CHECK();
  }
  if (!slow.empty()) {
    BufferPtr dslow = slow.size() == pages.size() ? dpages : upload(ctx, slow); KernelTimer kt(ctx, "pq_decode");
    hipLaunchKernelGGL(k_pq_decode, dim3((unsigned)slow.size()), dim3(PQ_NT), 0, ctx->stream, (const PqPage*)dslow->ptr, col, ctx->d_flags); KERNEL_CHECK();
  }
}

static BufferPtr bytes_to_validity(dfgpu_ctx* ctx, const BufferPtr& vbytes, int64_t n) {
  BufferPtr v = alloc_buffer(ctx, bitmap_bytes(n));
  if (n) hipLaunchKernelGGL(k_pq_bytes_to_bits, dim3(grid_for(((n + 63) / 64) * 64, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const uint8_t*)vbytes->ptr, n, (uint64_t*)v->ptr);
  KERNEL_CHECK(); return v;
}

// One column of one read: the host's walk over the page headers (plan_column; stages the chunks, lists the Snappy jobs of ALL columns so that one
// launch decompresses every page of the read) and the decode launches (decode_column).
struct ColumnRead {
  hipEvent_t copied = nullptr;                        // host image: recorded on the copy stream behind this column's last chunk
  ColumnRead() = default; ColumnRead(const ColumnRead&) = delete; ColumnRead& operator=(const ColumnRead&) = delete;
  ~ColumnRead() { if (copied) (void)hipEventDestroy(copied); }
  int leaf_idx = 0; int64_t total_rows = 0; std::vector<PqPage> pages, dict_str_pages; std::vector<BufferPtr> keep;
  bool any_levels = false, all_dict = true; int32_t dict_total = 0; int wp = 0;
};
// Zstandard pages: one wave per page (zstd_device.h); lit = 128 KB + 64 of literal scratch per job
constexpr size_t ZS_LIT = 128 * 1024 + 64;
__global__ void __launch_bounds__(64) k_pq_zstd(const SnJob* __restrict__ jobs, uint8_t* lit, uint32_t* flags) {
  __shared__ zs::Lds L; __shared__ __attribute__((aligned(16))) uint8_t ring[zs::ZS_RING]; __shared__ __attribute__((aligned(16))) uint8_t litl[zs::ZS_LIT_LDS];
  const SnJob jb = jobs[blockIdx.x]; const uint32_t lane = threadIdx.x;
  if (jb.raw) { for (uint32_t i = lane; i < jb.usize; i += 64) jb.dst[i] = jb.src[i]; return; }
  const bool ok = zs::decode_frame(&L, ring, litl, jb.src, jb.csize, jb.dst, jb.usize, lit + (size_t)blockIdx.x * ZS_LIT, lane);
  if (!ok && lane == 0) atomicOr(flags, DFGPU_FLAG_OOB);
}

// LZ4_RAW pages: a kernel of their own (sharing k_pq_zstd cost the Zstandard path a third of its speed: register pressure)
__global__ void __launch_bounds__(64) k_pq_lz4(const SnJob* __restrict__ jobs, uint32_t* flags) {
  __shared__ __attribute__((aligned(16))) uint8_t ring[zs::ZS_RING]; __shared__ __attribute__((aligned(16))) uint8_t win[16384];
  const SnJob jb = jobs[blockIdx.x]; const uint32_t lane = threadIdx.x;
  if (jb.raw) { for (uint32_t i = lane; i < jb.usize; i += 64) jb.dst[i] = jb.src[i]; return; }
  if (!zs::lz4_decode(ring, win, jb.src, jb.csize, jb.dst, jb.usize, lane) && lane == 0) atomicOr(flags, DFGPU_FLAG_OOB);
}

static void plan_column(dfgpu_ctx* ctx, dfgpu_parquet* f, int leaf_idx, int rg0, int nrg, ColumnRead& cr, std::vector<SnJob>& jobs) {
  const Leaf& leaf = f->leaves[(size_t)leaf_idx];
  if (!leaf.arrow) fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: Parquet column '%s': %s", leaf.name.c_str(), leaf.why.c_str());
  const bool is_str = leaf.arrow == DFGPU_UTF8;
  const int max_def = leaf.rep == 1 ? 1 : 0;             // OPTIONAL
  int64_t total_rows = 0; for (int g = rg0; g < rg0 + nrg; g++) total_rows += f->rgs[(size_t)g].rows;

  std::vector<PqPage>& pages = cr.pages; std::vector<PqPage>& dict_str_pages = cr.dict_str_pages; std::vector<BufferPtr>& keep = cr.keep;
  bool& any_levels = cr.any_levels; bool& all_dict = cr.all_dict; int64_t row = 0; int32_t& dict_total = cr.dict_total;
  cr.leaf_idx = leaf_idx; cr.total_rows = total_rows;
  int& wp = cr.wp; wp = leaf.phys == PT_INT32 || leaf.phys == PT_FLOAT ? 4 : leaf.phys == PT_INT64 || leaf.phys == PT_DOUBLE ? 8 : leaf.phys == PT_FLBA ? leaf.type_len : 0;
  bool staged = false;
  for (int g = rg0; g < rg0 + nrg; g++) {
    const RowGroup& rg = f->rgs[(size_t)g]; const Chunk& ch = rg.cols[(size_t)leaf_idx];
    if (ch.codec != CODEC_NONE && ch.codec != CODEC_SNAPPY && ch.codec != CODEC_ZSTD && ch.codec != CODEC_LZ4_RAW) fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: Parquet compression codec %d of column '%s' (UNCOMPRESSED, SNAPPY, ZSTD and LZ4_RAW are decoded on the device)", ch.codec, leaf.name.c_str());
    if (ch.num_values != rg.rows) fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: Parquet column '%s' holds %lld values for %lld rows (repeated values)", leaf.name.c_str(), (long long)ch.num_values, (long long)rg.rows);
    if (ch.num_values == 0) continue;                       // a row group without rows: nothing to walk
    int64_t start = ch.dict_off > 0 && ch.dict_off < ch.data_off ? ch.dict_off : ch.data_off;
    if (start < 4 || ch.total_comp < 0 || start + ch.total_comp > f->len) fail(DFGPU_EXECUTION, "Parquet error: column chunk of '%s' lies outside the file", leaf.name.c_str());
    const uint8_t* dsrc;
    if (f->dev) dsrc = f->dev + start;
    else { BufferPtr st = alloc_buffer(ctx, (size_t)ch.total_comp + 16); HIP_CHECK(hipMemcpyAsync(st->ptr, f->host + start, (size_t)ch.total_comp, hipMemcpyHostToDevice, ctx->copy_stream)); keep.push_back(st); dsrc = (const uint8_t*)st->ptr; staged = true; }
    // page headers (host), sizes of the uncompressed images
    struct P { PageHdr h; int64_t payload; }; std::vector<P> ps; int64_t pos = start, seen = 0, ubytes = 0;
    while (seen < ch.num_values) {
      if (pos >= start + ch.total_comp) fail(DFGPU_EXECUTION, "Parquet error: column chunk of '%s' ends after %lld of %lld values", leaf.name.c_str(), (long long)seen, (long long)ch.num_values);
      PageHdr h = parse_page_header(f->host + pos, f->host + start + ch.total_comp);
      int64_t payload = pos + h.hdr_bytes;
      if (payload + h.csize > start + ch.total_comp) fail(DFGPU_EXECUTION, "Parquet error: page of '%s' runs past its column chunk", leaf.name.c_str());
      if (h.type == PG_DATA || h.type == PG_DATA_V2) seen += h.nvals;
      if (h.type != PG_INDEX) { ps.push_back({h, payload}); ubytes += ((int64_t)h.usize + 31) & ~15ll; }
      pos = payload + h.csize;
    }
    uint8_t* ubase = nullptr;
    if (ch.codec != CODEC_NONE) { BufferPtr ub = alloc_buffer(ctx, (size_t)ubytes + 16); keep.push_back(ub); ubase = (uint8_t*)ub->ptr; }
    const uint8_t* dict_data = nullptr; int32_t dict_count = 0, dict_base = dict_total;
    for (auto& p : ps) {
      const PageHdr& h = p.h; const uint8_t* src = dsrc + (p.payload - start); const uint8_t* data = src;
      int32_t lvl = h.type == PG_DATA_V2 ? h.def_len + h.rep_len : 0;
      if (h.type == PG_DATA_V2 && (h.rep_len != 0 || lvl > h.csize || lvl > h.usize)) fail(DFGPU_EXECUTION, "Parquet error: level bytes of a v2 page of '%s'", leaf.name.c_str());
      if (ch.codec != CODEC_NONE) {
        bool comp = h.type != PG_DATA_V2 || h.v2_compressed;
        if (comp) {
          if (lvl) jobs.push_back({src, ubase, (uint32_t)lvl, (uint32_t)lvl, 1, ch.codec});
          if (h.usize - lvl > 0) jobs.push_back({src + lvl, ubase + lvl, (uint32_t)(h.csize - lvl), (uint32_t)(h.usize - lvl), 0, ch.codec});
          data = ubase; ubase += ((int64_t)h.usize + 31) & ~15ll;
        } else if (h.csize != h.usize) fail(DFGPU_EXECUTION, "Parquet error: uncompressed v2 page of '%s' with different sizes", leaf.name.c_str());      // decoded in place: only csize bytes were bounds-checked
      } else if (h.csize != h.usize) fail(DFGPU_EXECUTION, "Parquet error: uncompressed page of '%s' with different sizes", leaf.name.c_str());
      if (h.type == PG_DICT) {
        if (h.enc != ENC_PLAIN && h.enc != ENC_PLAIN_DICT) fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: dictionary page encoding %d", h.enc);
        dict_count = h.nvals; dict_data = data;
        if (is_str) { PqPage d{}; d.data = data; d.size = (uint32_t)h.usize; d.num_values = h.nvals; d.row_start = dict_total; dict_str_pages.push_back(d); dict_total += h.nvals; }
        else if ((int64_t)h.nvals * wp > h.usize) fail(DFGPU_EXECUTION, "Parquet error: dictionary page of '%s' is shorter than its %d values", leaf.name.c_str(), h.nvals);
        continue;
      }
      PqPage d{}; d.data = data; d.size = (uint32_t)h.usize; d.num_values = h.nvals; d.row_start = row; row += h.nvals;
      if (h.enc == ENC_PLAIN) { d.dict_enc = 0; all_dict = false; }
      else if (h.enc == ENC_RLE && leaf.phys == PT_BOOLEAN) { d.dict_enc = 2; all_dict = false; }
      else if (h.enc == ENC_RLE_DICT || h.enc == ENC_PLAIN_DICT) { d.dict_enc = 1; if (!dict_data && h.nvals) fail(DFGPU_EXECUTION, "Parquet error: dictionary-encoded page of '%s' without a dictionary page", leaf.name.c_str()); }
      else fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: Parquet encoding %d of column '%s' (PLAIN, RLE_DICTIONARY and RLE Booleans are decoded on the device)", h.enc, leaf.name.c_str());
      if (leaf.phys == PT_BOOLEAN && d.dict_enc == 1) fail(DFGPU_EXECUTION, "Parquet error: dictionary-encoded Boolean page");
      d.dict_data = dict_data; d.dict_count = dict_count; d.dict_base = dict_base;
      if (max_def) {
        if (h.type == PG_DATA_V2) { d.lvl_mode = 2; d.lvl_len = h.def_len; d.decode_levels = h.num_nulls != 0; }
        else { d.lvl_mode = 1; d.decode_levels = ch.null_count != 0; }
      } else if (h.type == PG_DATA_V2) { d.lvl_mode = 2; d.lvl_len = h.def_len; }
      any_levels |= d.decode_levels != 0;
      pages.push_back(d);
    }
  }
  if (row != total_rows) fail(DFGPU_EXECUTION, "Parquet error: pages of '%s' hold %lld values for %lld rows", leaf.name.c_str(), (long long)row, (long long)total_rows);
  if (staged) { HIP_CHECK(hipEventCreateWithFlags(&cr.copied, hipEventDisableTiming)); HIP_CHECK(hipEventRecord(cr.copied, ctx->copy_stream)); }
}
static dfgpu_array* decode_column(dfgpu_ctx* ctx, dfgpu_parquet* f, ColumnRead& cr) {
  const Leaf& leaf = f->leaves[(size_t)cr.leaf_idx]; const bool is_str = leaf.arrow == DFGPU_UTF8;
  const int64_t total_rows = cr.total_rows; std::vector<PqPage>& pages = cr.pages; std::vector<PqPage>& dict_str_pages = cr.dict_str_pages;
  const bool any_levels = cr.any_levels, all_dict = cr.all_dict; const int32_t dict_total = cr.dict_total; const int wp = cr.wp;
  BufferPtr vbytes; if (any_levels) vbytes = alloc_buffer(ctx, (size_t)total_rows + 64);
  PqCol col{}; col.vbytes = vbytes ? (uint8_t*)vbytes->ptr : nullptr; col.wp = wp;
  ArrayHolder out;
  if (!is_str) {
    int32_t t = leaf.arrow; col.mode = MODE_FIXED; col.wo = t == DFGPU_BOOL ? 1 : type_width(t);
    col.conv = t == DFGPU_BOOL ? CONV_BOOL : t == DFGPU_DECIMAL128 ? (leaf.phys == PT_INT32 ? CONV_I32_DEC : leaf.phys == PT_INT64 ? CONV_I64_DEC : CONV_FLBA_DEC) : col.wo == 8 ? CONV_COPY8 : col.wo == 4 ? CONV_COPY4 : col.wo == 2 ? CONV_4TO2 : CONV_4TO1;
    int prec = leaf.precision, scale = leaf.scale;
    if (t == DFGPU_DECIMAL128 && (prec < 1 || prec > 38)) fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: decimal precision %d of column '%s'", prec, leaf.name.c_str());
    ArrayHolder a(new_array(ctx, t, total_rows, t == DFGPU_DECIMAL128 ? prec : 0, t == DFGPU_DECIMAL128 ? scale : 0));
    BufferPtr vals = alloc_buffer(ctx, std::max<size_t>((size_t)total_rows * (size_t)col.wo, 16));
    col.out = vals->ptr;
    launch_decode(ctx, pages, col, true);
    if (t == DFGPU_BOOL) a.get()->values = bytes_to_validity(ctx, vals, total_rows); else a.get()->values = vals;
    if (vbytes) { a.get()->validity = bytes_to_validity(ctx, vbytes, total_rows); a.get()->null_count = -1; } else a.get()->null_count = 0;
    out.a = a.release();
  } else {
    // the concatenated dictionary of the row groups read (a Utf8 array), when any page is dictionary encoded
    ArrayHolder dict; StrOut dso;
    if (!dict_str_pages.empty()) {
      BufferPtr doff = alloc_buffer(ctx, (size_t)(dict_total + 1) * 4 + 16), dsrc = alloc_buffer(ctx, std::max<size_t>((size_t)dict_total * 8, 16));
      PqCol dc{}; dc.mode = MODE_STRING; dc.slen = (uint32_t*)doff->ptr; dc.ssrc = (uint64_t*)dsrc->ptr;
      launch_decode(ctx, dict_str_pages, dc, false);
      dso = finish_strings(ctx, dict_total, doff, dsrc);
      ArrayHolder d(new_array(ctx, DFGPU_UTF8, dict_total)); d.get()->offsets = dso.offsets; d.get()->values = dso.chars; d.get()->values_bytes = dso.chars_bytes; d.get()->null_count = 0;
      dict.a = d.release();
    }
    const bool keys_only = f->utf8_dictionary && all_dict && dict.get();
    if (keys_only) {
      BufferPtr keys = alloc_buffer(ctx, std::max<size_t>((size_t)total_rows * 4, 16));
      col.mode = MODE_KEYS; col.out = keys->ptr;
      launch_decode(ctx, pages, col, false);
      ArrayHolder a(new_array(ctx, DFGPU_DICTIONARY, total_rows)); a.get()->key_type = DFGPU_INT32; a.get()->values = keys;
      if (vbytes) { a.get()->validity = bytes_to_validity(ctx, vbytes, total_rows); a.get()->null_count = -1; } else a.get()->null_count = 0;
      a.get()->dictionary = dict.release();
      out.a = a.release();
    } else {
      BufferPtr off = alloc_buffer(ctx, (size_t)(total_rows + 1) * 4 + 16), ssrc = alloc_buffer(ctx, std::max<size_t>((size_t)total_rows * 8, 16));
      col.mode = MODE_STRING; col.slen = (uint32_t*)off->ptr; col.ssrc = (uint64_t*)ssrc->ptr;
      if (dict.get()) { col.dict_offsets = (const int32_t*)dso.offsets->ptr; col.dict_chars = (const uint8_t*)dso.chars->ptr; }
      launch_decode(ctx, pages, col, false);
      StrOut so = finish_strings(ctx, total_rows, off, ssrc);
      ArrayHolder a(new_array(ctx, DFGPU_UTF8, total_rows)); a.get()->offsets = so.offsets; a.get()->values = so.chars; a.get()->values_bytes = so.chars_bytes;
      if (vbytes) { a.get()->validity = bytes_to_validity(ctx, vbytes, total_rows); a.get()->null_count = -1; } else a.get()->null_count = 0;
      if (f->utf8_dictionary) {           // schema stability: the column is Dictionary(Int32, Utf8) in every batch; a chunk that fell back to PLAIN pages gets identity keys
        BufferPtr keys = alloc_buffer(ctx, std::max<size_t>((size_t)total_rows * 4, 16)); launch_iota_u32(ctx, (uint32_t*)keys->ptr, total_rows, 0);
        ArrayHolder d(new_array(ctx, DFGPU_DICTIONARY, total_rows)); d.get()->key_type = DFGPU_INT32; d.get()->values = keys; d.get()->validity = a.get()->validity; d.get()->null_count = a.get()->null_count;
        a.get()->validity = nullptr; a.get()->null_count = 0;          // NULL slots are empty strings in the value column; the keys carry the validity
        d.get()->dictionary = a.release(); out.a = d.release();
      } else out.a = a.release();
    }
  }
  return out.release();
}

}  // namespace pq
}  // namespace dfgpu

extern "C" {

dfgpu_status dfgpu_parquet_open(dfgpu_ctx* ctx, const uint8_t* file_bytes, int64_t len, const uint8_t* device_bytes, dfgpu_parquet** out) {
  return guard(ctx, [&] {
    if (!file_bytes || !out) fail(DFGPU_INVALID_ARGUMENT, "parquet_open: null argument");
    std::unique_ptr<dfgpu_parquet> f(new dfgpu_parquet()); f->host = file_bytes; f->len = len; f->dev = device_bytes;
    parse_footer(f.get()); *out = f.release();
  }, true);          // ctx may be NULL: the footer alone is read (metadata)
}
dfgpu_status dfgpu_parquet_open_file(dfgpu_ctx* ctx, const char* path, int32_t stage_on_device, dfgpu_parquet** out) {
  return guard(ctx, [&] {
    if (!path || !out) fail(DFGPU_INVALID_ARGUMENT, "parquet_open_file: null argument");
    int fd = open(path, O_RDONLY); if (fd < 0) fail(DFGPU_EXECUTION, "Object Store error: cannot open %s", path);
    struct stat st; if (fstat(fd, &st) != 0 || st.st_size < 12) { close(fd); fail(DFGPU_EXECUTION, "Parquet error: Invalid Parquet file. Size is smaller than footer"); }
    // host image: a private writable mapping (never written) can be page-locked, a read-only one cannot
    void* m = mmap(nullptr, (size_t)st.st_size, stage_on_device ? PROT_READ : PROT_READ | PROT_WRITE, MAP_PRIVATE, fd, 0); close(fd);
    if (m == MAP_FAILED) fail(DFGPU_EXECUTION, "Object Store error: cannot map %s", path);
    std::unique_ptr<dfgpu_parquet> f(new dfgpu_parquet()); f->map = m; f->map_len = (size_t)st.st_size; f->host = (const uint8_t*)m; f->len = st.st_size;
    parse_footer(f.get());
    if (stage_on_device) {
      if (!ctx) fail(DFGPU_INVALID_ARGUMENT, "parquet_open_file: staging on the device needs a ctx");
      HIP_CHECK(hipSetDevice(ctx->device));
      f->dev_owned = alloc_buffer(ctx, (size_t)f->len + 16); HIP_CHECK(hipMemcpyAsync(f->dev_owned->ptr, f->host, (size_t)f->len, hipMemcpyHostToDevice, ctx->stream)); HIP_CHECK(hipStreamSynchronize(ctx->stream));
      f->dev = (const uint8_t*)f->dev_owned->ptr;
    } else if (ctx) {
      // page-lock the image once: column chunks then move by DMA (~55 GB/s over PCIe 5 x16) and asynchronously; from pageable memory every copy goes through the runtime's bounce
      // buffer and holds the calling thread.  A mapping that cannot be locked (RLIMIT_MEMLOCK, a filesystem that refuses) is read as before.
      HIP_CHECK(hipSetDevice(ctx->device));
      if (hipHostRegister(m, (size_t)st.st_size, hipHostRegisterDefault) == hipSuccess) f->registered = true; else (void)hipGetLastError();
    }
    *out = f.release();
  }, true);          // ctx may be NULL: metadata only, read through the host mapping
}
void dfgpu_parquet_close(dfgpu_parquet* f) { delete f; }
dfgpu_status dfgpu_parquet_set_option(dfgpu_parquet* f, const char* key, int64_t value) {
  if (!f || !key) return DFGPU_INVALID_ARGUMENT;
  if (!strcmp(key, "utf8_dictionary")) { f->utf8_dictionary = value != 0; return DFGPU_OK; }
  return DFGPU_INVALID_ARGUMENT;
}
int64_t dfgpu_parquet_num_rows(const dfgpu_parquet* f) { return f ? f->num_rows : -1; }
int32_t dfgpu_parquet_num_row_groups(const dfgpu_parquet* f) { return f ? (int32_t)f->rgs.size() : -1; }
int32_t dfgpu_parquet_num_columns(const dfgpu_parquet* f) { return f ? (int32_t)f->leaves.size() : -1; }
int64_t dfgpu_parquet_row_group_rows(const dfgpu_parquet* f, int32_t rg) { return f && rg >= 0 && (size_t)rg < f->rgs.size() ? f->rgs[(size_t)rg].rows : -1; }
const char* dfgpu_parquet_column_name(const dfgpu_parquet* f, int32_t c) { return f && c >= 0 && (size_t)c < f->leaves.size() ? f->leaves[(size_t)c].name.c_str() : nullptr; }
dfgpu_status dfgpu_parquet_column_type(const dfgpu_parquet* f, int32_t c, int32_t* type, int32_t* value_type, int32_t* precision, int32_t* scale, int32_t* nullable) {
  if (!f || c < 0 || (size_t)c >= f->leaves.size()) return DFGPU_INVALID_ARGUMENT;
  const Leaf& l = f->leaves[(size_t)c];
  int32_t t = l.arrow == DFGPU_UTF8 && f->utf8_dictionary ? DFGPU_DICTIONARY : l.arrow;
  if (type) *type = t;
  if (value_type) *value_type = l.arrow;
  if (precision) *precision = l.arrow == DFGPU_DECIMAL128 ? l.precision : 0;
  if (scale) *scale = l.arrow == DFGPU_DECIMAL128 ? l.scale : 0;
  if (nullable) *nullable = l.rep == 1;
  return DFGPU_OK;
}
dfgpu_status dfgpu_parquet_column_stats(const dfgpu_parquet* f, int32_t rg, int32_t c, int64_t* min_v, int64_t* max_v, int64_t* null_count, int32_t* has_min_max) {
  if (!f || rg < 0 || (size_t)rg >= f->rgs.size() || c < 0 || (size_t)c >= f->leaves.size()) return DFGPU_INVALID_ARGUMENT;
  const Chunk& ch = f->rgs[(size_t)rg].cols[(size_t)c];
  if (null_count) *null_count = ch.null_count;
  int32_t has = 0; int64_t mn = 0, mx = 0;
  if (ch.has_minmax && (ch.phys == PT_INT32 || ch.phys == PT_INT64) && f->leaves[(size_t)c].arrow != DFGPU_DECIMAL128) {
    size_t w = ch.phys == PT_INT32 ? 4 : 8;
    if (ch.min_v.size() == w && ch.max_v.size() == w) {
      if (w == 4) { int32_t a, b; memcpy(&a, ch.min_v.data(), 4); memcpy(&b, ch.max_v.data(), 4); mn = a; mx = b; } else { memcpy(&mn, ch.min_v.data(), 8); memcpy(&mx, ch.max_v.data(), 8); }
      int32_t at = f->leaves[(size_t)c].arrow; has = !(at == DFGPU_UINT32 || at == DFGPU_UINT64);      // unsigned orders are not the physical order
    }
  }
  if (min_v) *min_v = mn; if (max_v) *max_v = mx; if (has_min_max) *has_min_max = has;
  return DFGPU_OK;
}
int64_t dfgpu_parquet_column_chunk_bytes(const dfgpu_parquet* f, int32_t rg, int32_t c, int32_t uncompressed) {
  if (!f || rg < 0 || (size_t)rg >= f->rgs.size() || c < 0 || (size_t)c >= f->leaves.size()) return -1;
  const Chunk& ch = f->rgs[(size_t)rg].cols[(size_t)c]; return uncompressed ? ch.total_uncomp : ch.total_comp;
}

dfgpu_status dfgpu_parquet_read(dfgpu_ctx* ctx, dfgpu_parquet* f, int32_t first_row_group, int32_t num_row_groups, const int32_t* columns, int32_t ncols, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!f || !out || (ncols > 0 && !columns)) fail(DFGPU_INVALID_ARGUMENT, "parquet_read: null argument");
    if (first_row_group < 0 || num_row_groups < 0 || (size_t)first_row_group + (size_t)num_row_groups > f->rgs.size()) fail(DFGPU_INVALID_ARGUMENT, "parquet_read: row groups [%d, %d) of %zu", first_row_group, first_row_group + num_row_groups, f->rgs.size());
    HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<ArrayHolder> res; std::vector<ColumnRead> reads((size_t)ncols); std::vector<SnJob> jobs;
    // A file image in host memory: the column chunks cross PCIe on the copy stream, one event per column, while `stream` decodes the columns that have arrived (ParquetExec's reader
    // fetches the projected chunks' byte ranges ahead of the decoder the same way, parquet/mod.rs ParquetOpener -> AsyncFileReader::get_byte_ranges).  The staging buffers come from
    // the stream-ordered cache of `stream`, so the copies start behind everything `stream` has been given so far; an error on the way waits for the copies before the buffers go back.
    struct CopyDrain { dfgpu_ctx* c; bool armed = true; ~CopyDrain() { if (armed && c->copy_stream) (void)hipStreamSynchronize(c->copy_stream); } } drain{ctx};
    if (!f->dev) {
      { static std::mutex create_mu; std::lock_guard<std::mutex> l(create_mu);       // plan partitions read concurrently on one ctx: one of them creates the stream
        if (!ctx->copy_stream) HIP_CHECK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking)); }
      hipEvent_t e0; HIP_CHECK(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
      hipError_t e1 = hipEventRecord(e0, ctx->stream), e2 = e1 == hipSuccess ? hipStreamWaitEvent(ctx->copy_stream, e0, 0) : e1; (void)hipEventDestroy(e0); HIP_CHECK(e2);
    }
    // byte-array columns first: their decode is the long one (a walk over length prefixes), so it should start while the fixed-width chunks are still on the wire
    std::vector<int32_t> order; order.reserve((size_t)ncols);
    for (int32_t i = 0; i < ncols; i++) if (columns[i] < 0 || (size_t)columns[i] >= f->leaves.size()) fail(DFGPU_INVALID_ARGUMENT, "parquet_read: column %d of %zu", columns[i], f->leaves.size());
    for (int pass = 0; pass < 2; pass++) for (int32_t i = 0; i < ncols; i++) if ((f->leaves[(size_t)columns[i]].arrow == DFGPU_UTF8) == (pass == 0)) order.push_back(i);
    for (int32_t i : order) plan_column(ctx, f, columns[i], first_row_group, num_row_groups, reads[(size_t)i], jobs);
    const bool compressed = !jobs.empty();
    if (compressed) for (int32_t i = 0; i < ncols; i++) if (reads[(size_t)i].copied) HIP_CHECK(hipStreamWaitEvent(ctx->stream, reads[(size_t)i].copied, 0));      // the decompressors take every page of the read in one launch
    std::vector<SnJob> zjobs, ljobs; { std::vector<SnJob> sj; for (auto& j : jobs) (j.codec == CODEC_ZSTD ? zjobs : j.codec == CODEC_LZ4_RAW ? ljobs : sj).push_back(j); jobs.swap(sj); }
    if (!ljobs.empty()) {
      std::stable_sort(ljobs.begin(), ljobs.end(), [](const SnJob& x, const SnJob& y) { return x.usize > y.usize; });
      BufferPtr dl = upload(ctx, ljobs);
      KernelTimer kt(ctx, "pq_lz4");
      hipLaunchKernelGGL(k_pq_lz4, dim3((unsigned)ljobs.size()), dim3(64), 0, ctx->stream, (const SnJob*)dl->ptr, ctx->d_flags);
      KERNEL_CHECK();
    }
    BufferPtr zlit;
    if (!zjobs.empty()) {                         // Zstandard: a frame is sequential, the pages of the read are the parallelism -- one wave each
      std::stable_sort(zjobs.begin(), zjobs.end(), [](const SnJob& x, const SnJob& y) { return x.usize > y.usize; });          // the longest pages start first
      BufferPtr dz = upload(ctx, zjobs); zlit = alloc_buffer(ctx, zjobs.size() * ZS_LIT);
      KernelTimer kt(ctx, "pq_zstd");
      hipLaunchKernelGGL(k_pq_zstd, dim3((unsigned)zjobs.size()), dim3(64), 0, ctx->stream, (const SnJob*)dz->ptr, (uint8_t*)zlit->ptr, ctx->d_flags);
      KERNEL_CHECK();
#ifdef ZS_PROFILE
      { long long t[4]; HIP_CHECK(hipMemcpy(t, zlit->ptr, 32, hipMemcpyDeviceToHost)); fprintf(stderr, "zstd job0: literals %.2f ms, tables/headers %.2f ms, sequences %.2f ms, usize %lld, jobs %zu\n", t[0] / 1e5, t[1] / 1e5, t[2] / 1e5, t[3], zjobs.size()); }
#endif
    }
    if (!jobs.empty()) {                          // every compressed page of the read in one launch: the pages are the parallelism
      BufferPtr dj = upload(ctx, jobs);
      std::vector<SnBlk> blks;                   // one workgroup per 64 KB output block of every page, the largest pages' blocks first in launch order
      for (size_t j = 0; j < jobs.size(); j++) { uint32_t nb = jobs[j].raw ? 1u : std::max(1u, (jobs[j].usize + 65535u) / 65536u); for (uint32_t k = 0; k < nb; k++) blks.push_back(SnBlk{(uint32_t)j, k}); }
      std::stable_sort(blks.begin(), blks.end(), [](const SnBlk& x, const SnBlk& y) { return x.blk > y.blk; });       // blocks with the longest tag skip start first
      BufferPtr db = upload(ctx, blks), bad = alloc_buffer(ctx, jobs.size() * 4 + 16, true);
      { KernelTimer kt(ctx, "pq_snappy");
        hipLaunchKernelGGL(k_pq_snappy, dim3((unsigned)blks.size()), dim3(64), 0, ctx->stream, (const SnJob*)dj->ptr, (const SnBlk*)db->ptr, (uint32_t*)bad->ptr, ctx->d_flags); }
      { KernelTimer kt(ctx, "pq_snappy_repair");
        hipLaunchKernelGGL(k_pq_snappy, dim3((unsigned)jobs.size()), dim3(64), 0, ctx->stream, (const SnJob*)dj->ptr, (const SnBlk*)nullptr, (uint32_t*)bad->ptr, ctx->d_flags); }
      KERNEL_CHECK();
    }
    res.resize((size_t)ncols);
    for (int32_t i : order) {
      if (!compressed && reads[(size_t)i].copied) HIP_CHECK(hipStreamWaitEvent(ctx->stream, reads[(size_t)i].copied, 0));
      res[(size_t)i].a = decode_column(ctx, f, reads[(size_t)i]);
    }
    if (!f->dev && ctx->copy_stream) {         // the handle's own marker behind this read's copies (closing the file waits for it)
      std::lock_guard<std::mutex> l(f->copy_mu);
      if (!f->last_copy) HIP_CHECK(hipEventCreateWithFlags(&f->last_copy, hipEventDisableTiming));
      HIP_CHECK(hipEventRecord(f->last_copy, ctx->copy_stream));
    }
    drain.armed = false;                       // every copy is ordered before a kernel of `stream` from here on
    // the staged / decompressed bytes (reads[].keep) outlive the kernels: frees are stream ordered through the caching allocator
    check_flags(ctx, "Parquet page decode (malformed page, run or dictionary index)");
    for (int32_t i = 0; i < ncols; i++) out[i] = res[(size_t)i].release();
  });
}

}  // extern "C"
