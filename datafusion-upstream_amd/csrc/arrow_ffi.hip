// arrow_ffi.hip -- Arrow C Data Interface bridge: this is what a Rust shim hands over via arrow::ffi
// (FFI_ArrowArray / FFI_ArrowSchema, arrow-rs 50) and what pyarrow's _export_to_c / _import_from_c speak.
// Import copies host buffers to HBM over PCIe; export copies back and transfers ownership through `release`.
#include "dfgpu_internal.h"
#include <cstdlib>

namespace dfgpu {

static int32_t parse_format(const char* f, int32_t* p, int32_t* s) {
  *p = 0; *s = 0;
  if (!f) return 0;
  std::string x(f);
  if (x == "b") return DFGPU_BOOL; if (x == "c") return DFGPU_INT8; if (x == "C") return DFGPU_UINT8;
  if (x == "s") return DFGPU_INT16; if (x == "S") return DFGPU_UINT16; if (x == "i") return DFGPU_INT32;
  if (x == "I") return DFGPU_UINT32; if (x == "l") return DFGPU_INT64; if (x == "L") return DFGPU_UINT64;
  if (x == "f") return DFGPU_FLOAT32; if (x == "g") return DFGPU_FLOAT64; if (x == "tdD") return DFGPU_DATE32;
  if (x == "u") return DFGPU_UTF8;
  if (x.rfind("d:", 0) == 0) {
    int pp = 0, ss = 0, bw = 128;
    int k = sscanf(f, "d:%d,%d,%d", &pp, &ss, &bw);
    if (k >= 2 && bw == 128) { *p = pp; *s = ss; return DFGPU_DECIMAL128; }
  }
  return 0;
}
static std::string format_of(int32_t t, int32_t p, int32_t s) {
  switch (t) {
    case DFGPU_BOOL: return "b"; case DFGPU_INT8: return "c"; case DFGPU_UINT8: return "C"; case DFGPU_INT16: return "s"; case DFGPU_UINT16: return "S";
    case DFGPU_INT32: return "i"; case DFGPU_UINT32: return "I"; case DFGPU_INT64: return "l"; case DFGPU_UINT64: return "L";
    case DFGPU_FLOAT32: return "f"; case DFGPU_FLOAT64: return "g"; case DFGPU_DATE32: return "tdD"; case DFGPU_UTF8: return "u";
    case DFGPU_DECIMAL128: return "d:" + std::to_string(p) + "," + std::to_string(s);
    default: return "";
  }
}

static void fill_desc(struct ArrowArray* a, struct ArrowSchema* sc, dfgpu_array_desc* d, std::vector<std::unique_ptr<dfgpu_array_desc>>& keep) {
  if (!a || !sc || !a->release || !sc->release) fail(DFGPU_INVALID_ARGUMENT, "import_arrow: released or null ArrowArray/ArrowSchema");
  if (a->offset != 0) fail(DFGPU_NOT_IMPLEMENTED, "import_arrow: arrays with a non-zero offset (slice) are not supported; copy the slice first");
  memset(d, 0, sizeof *d);
  int32_t p, s; int32_t t = parse_format(sc->format, &p, &s);
  if (!t) fail(DFGPU_NOT_IMPLEMENTED, "import_arrow: unsupported Arrow format '%s'", sc->format ? sc->format : "(null)");
  d->length = a->length; d->null_count = a->null_count;
  if (sc->dictionary) {
    if (!a->dictionary) fail(DFGPU_INVALID_ARGUMENT, "import_arrow: dictionary schema without dictionary array");
    if (!(t >= DFGPU_INT8 && t <= DFGPU_UINT64)) fail(DFGPU_INVALID_ARGUMENT, "import_arrow: dictionary index format '%s'", sc->format);
    d->type = DFGPU_DICTIONARY; d->key_type = t;
    if (a->n_buffers < 2) fail(DFGPU_INVALID_ARGUMENT, "import_arrow: dictionary array needs 2 buffers");
    d->validity = (const uint8_t*)a->buffers[0]; d->values = a->buffers[1];
    keep.emplace_back(new dfgpu_array_desc());
    dfgpu_array_desc* dd = keep.back().get();
    fill_desc(a->dictionary, sc->dictionary, dd, keep);
    d->dictionary = dd;
    return;
  }
  d->type = t; d->precision = p; d->scale = s;
  if (t == DFGPU_UTF8) {
    if (a->n_buffers < 3) fail(DFGPU_INVALID_ARGUMENT, "import_arrow: utf8 array needs 3 buffers");
    d->validity = (const uint8_t*)a->buffers[0]; d->offsets = (const int32_t*)a->buffers[1]; d->values = a->buffers[2];
    d->values_bytes = a->length ? d->offsets[a->length] : 0;
    static const char empty = 0; if (!d->values) d->values = &empty;
  } else {
    if (a->n_buffers < 2) fail(DFGPU_INVALID_ARGUMENT, "import_arrow: primitive array needs 2 buffers");
    d->validity = (const uint8_t*)a->buffers[0]; d->values = a->buffers[1];
    static const uint64_t zero[2] = {0, 0}; if (!d->values) d->values = zero;    // empty arrays may carry null buffers
  }
  if (a->null_count == 0) d->validity = nullptr;
}

struct ExportPrivate { std::vector<void*> bufs; const void* ptrs[3]; std::string format; struct ArrowArray* dict_arr = nullptr; struct ArrowSchema* dict_sc = nullptr; };
static void release_array(struct ArrowArray* a) {
  auto* pr = (ExportPrivate*)a->private_data;
  if (a->dictionary) { if (a->dictionary->release) a->dictionary->release(a->dictionary); delete a->dictionary; }
  if (pr) { for (void* b : pr->bufs) free(b); delete pr; }
  a->release = nullptr;
}
static void release_schema(struct ArrowSchema* s) {
  if (s->dictionary) { if (s->dictionary->release) s->dictionary->release(s->dictionary); delete s->dictionary; }
  delete (std::string*)s->private_data;
  s->release = nullptr;
}

static void export_one(dfgpu_ctx* ctx, const dfgpu_array* a, struct ArrowArray* oa, struct ArrowSchema* os) {
  memset(oa, 0, sizeof *oa); memset(os, 0, sizeof *os);
  auto* pr = new ExportPrivate();
  int64_t n = a->length;
  int32_t vt = a->type == DFGPU_DICTIONARY ? a->key_type : a->type;
  size_t vbytes = a->type == DFGPU_UTF8 ? (size_t)a->values_bytes : (vt == DFGPU_BOOL ? (size_t)(n + 7) / 8 : (size_t)n * type_width(vt));
  uint8_t* hval = nullptr; int32_t* hoff = nullptr; void* hv = nullptr;
  if (a->validity) { hval = (uint8_t*)calloc(1, (size_t)(n + 7) / 8 + 64); pr->bufs.push_back(hval); }
  if (a->type == DFGPU_UTF8) { hoff = (int32_t*)calloc((size_t)n + 1 + 16, 4); pr->bufs.push_back(hoff); }
  dfgpu_status st;
  if (a->type == DFGPU_UTF8) {
    // the bytes of THIS array are offsets[0] .. offsets[n] of its value buffer (a slice shares its parent's buffer; values_bytes is the parent's): offsets first, then that range
    st = dfgpu_array_export_host(ctx, a, nullptr, hval, hoff);
    const int32_t first = n ? hoff[0] : 0, last = n ? hoff[n] : 0;
    vbytes = (size_t)(last - first); hv = calloc(1, vbytes + 64); pr->bufs.push_back(hv);
    if (st == DFGPU_OK && vbytes) { if (hipMemcpy(hv, (const char*)a->values->ptr + first, vbytes, hipMemcpyDeviceToHost) != hipSuccess) { st = DFGPU_INTERNAL; ctx->err = "export_arrow: copy of string bytes failed"; } }
    for (int64_t i = 0; i <= n && first; i++) hoff[i] -= first;
  } else {
    hv = calloc(1, vbytes + 64); pr->bufs.push_back(hv);
    st = dfgpu_array_export_host(ctx, a, hv, hval, hoff);
  }
  if (st != DFGPU_OK) { for (void* b : pr->bufs) free(b); delete pr; fail(st, "%s", ctx->err.c_str()); }
  int64_t nulls = 0;
  if (hval) { for (int64_t i = 0; i < n; i++) nulls += !((hval[i >> 3] >> (i & 7)) & 1); if (n & 7) hval[(n - 1) >> 3] &= (uint8_t)((1u << (n & 7)) - 1); }
  if (vt == DFGPU_BOOL && a->type != DFGPU_UTF8 && (n & 7)) ((uint8_t*)hv)[(n - 1) >> 3] &= (uint8_t)((1u << (n & 7)) - 1);
  oa->length = n; oa->null_count = nulls; oa->offset = 0; oa->n_children = 0; oa->children = nullptr;
  if (a->type == DFGPU_UTF8) { pr->ptrs[0] = hval; pr->ptrs[1] = hoff; pr->ptrs[2] = hv; oa->n_buffers = 3; }
  else { pr->ptrs[0] = hval; pr->ptrs[1] = hv; oa->n_buffers = 2; }
  oa->buffers = pr->ptrs; oa->private_data = pr; oa->release = release_array;
  auto* fmt = new std::string(format_of(vt, a->precision, a->scale));
  os->format = fmt->c_str(); os->name = ""; os->metadata = nullptr; os->flags = 2 /* ARROW_FLAG_NULLABLE */; os->private_data = fmt; os->release = release_schema;
  if (a->type == DFGPU_DICTIONARY) {
    oa->dictionary = new ArrowArray(); os->dictionary = new ArrowSchema();
    export_one(ctx, a->dictionary, oa->dictionary, os->dictionary);
  }
}

}  // namespace dfgpu

using namespace dfgpu;
extern "C" {

dfgpu_status dfgpu_array_import_arrow(dfgpu_ctx* ctx, struct ArrowArray* array, struct ArrowSchema* schema, dfgpu_array** out) {
  return guard(ctx, [&] {
    if (!out) fail(DFGPU_INVALID_ARGUMENT, "import_arrow: null out");
    dfgpu_array_desc d; std::vector<std::unique_ptr<dfgpu_array_desc>> keep;
    fill_desc(array, schema, &d, keep);
    dfgpu_status st = dfgpu_array_import_host(ctx, &d, out);
    if (st != DFGPU_OK) fail(st, "%s", ctx->err.c_str());
  });
}
dfgpu_status dfgpu_array_export_arrow(dfgpu_ctx* ctx, const dfgpu_array* a, struct ArrowArray* out_array, struct ArrowSchema* out_schema) {
  return guard(ctx, [&] {
    if (!a || !out_array || !out_schema) fail(DFGPU_INVALID_ARGUMENT, "export_arrow: null argument");
    export_one(ctx, a, out_array, out_schema);
  });
}

}  // extern "C"
