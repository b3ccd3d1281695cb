// zstd_device.h -- Zstandard frames (RFC 8878) decoded on the device: the codec DataFusion's own Parquet writer defaults to (`datafusion.execution.parquet.compression =
// zstd(3)`, common/src/config.rs) and that the `parquet` crate (arrow-rs 50, not part of the reference tree) hands to the `zstd` crate on the CPU.  Restated from the
// published format: frame header, raw / RLE / compressed blocks, literals (raw, RLE, Huffman with 1 or 4 streams, tree reuse), Huffman weights (direct or FSE-coded),
// sequences (predefined / RLE / FSE-coded / repeated tables for literal lengths, offsets and match lengths, repeat offsets), no dictionaries.
//
// One wave per frame (= per Parquet page).  A frame is sequential by construction -- blocks share the window, the repeat offsets and reusable tables, and a block's
// sequences are one FSE bit stream read back to front -- so the parallelism is across the pages of a read, not inside one.  Inside the wave: every lane runs the
// same parsing code on the same values (uniform branches, table reads broadcast from LDS), the four Huffman streams of a literals section are decoded by four lanes,
// and the byte movement of every sequence (its literals, then its match, overlapping matches included) is done by all 64 lanes.
#pragma once
#include "device_utils.h"

namespace dfgpu {
namespace zs {

struct FseEnt { uint16_t base; uint8_t nbits; uint8_t sym; };
struct SeqEnt { uint16_t base; uint8_t nbits; uint8_t addbits; uint32_t value; };      // sequence tables: the code's base value and number of extra bits ride in the entry -- one LDS read per state
constexpr int ZS_HUF_LOG = 11, ZS_LL_LOG = 9, ZS_ML_LOG = 9, ZS_OF_LOG = 8, ZS_BLOCK_MAX = 128 * 1024;
struct Lds {                              // ~17 KB per wave
  uint16_t huf[1 << ZS_HUF_LOG];          // sym | nbits << 8
  SeqEnt ll[1 << ZS_LL_LOG], ml[1 << ZS_ML_LOG], of[1 << ZS_OF_LOG]; FseEnt tmp[1 << ZS_LL_LOG], wt[64];
  int16_t norm[256]; uint16_t next[256]; uint8_t weight[256];
  int32_t ll_log, ml_log, of_log, huf_log, huf_ok;
};

__device__ inline uint64_t ld64(const uint8_t* p) {             // 8 bytes at any address, little endian, through two aligned loads
  const uintptr_t a = (uintptr_t)p; const uint64_t* q = (const uint64_t*)(a & ~(uintptr_t)7); const unsigned sh = (unsigned)(a & 7) * 8;
  const uint64_t w0 = q[0]; if (!sh) return w0;
  return (w0 >> sh) | (q[1] << (64 - sh));
}
// bits [pos, pos + n) of a stream (bit 0 = LSB of byte 0), n <= 32; bits below 0 read as zero (the format allows the last reads of a backward stream to run out)
__device__ inline uint32_t bits_at(const uint8_t* s, int64_t pos, int n) {
  if (n <= 0) return 0;
  if (pos < 0) { const int m = n + (int)pos; if (m <= 0) return 0; return (uint32_t)((ld64(s) & ((1ull << m) - 1ull)) << (-(int)pos)); }
  return (uint32_t)((ld64(s + (pos >> 3)) >> (pos & 7)) & ((1ull << n) - 1ull));
}
// backward stream: unread bits are [0, bit) (bit 0 = LSB of the stream's first byte).  The reader keeps the two 8-byte-aligned words around the read position in registers
// (hi = word w, lo = word w - 1: 128 bits) and the word below them (nx = word w - 2) already loaded: when the position leaves `lo`, the words shift down and the load of the
// next one is only ISSUED -- it has a whole word of reads (two or more sequences) to arrive, so a refill never waits on memory.  Words below the stream's first byte read as
// zero (the format lets the last reads of a stream run out).
struct Back { const uint64_t* q; int64_t bit; int64_t w; int64_t lead; uint64_t hi, lo, nx; };       // q: aligned base, lead: bits of q[0] in front of the stream, w: index of `hi`
__device__ inline uint64_t back_word(const Back* b, int64_t w) { if (w < 0) return 0ull; const uint64_t v = b->q[w]; return w == 0 && b->lead ? (v >> b->lead) << b->lead : v; }
__device__ inline bool back_init(Back* b, const uint8_t* s, uint32_t len) {
  if (!len) return false; const uint8_t last = s[len - 1]; if (!last) return false;
  const uintptr_t a = (uintptr_t)s; b->q = (const uint64_t*)(a & ~(uintptr_t)7); b->lead = (int64_t)(a & 7) * 8;
  b->bit = b->lead + (int64_t)(len - 1) * 8 + (31 - __clz((int)last));           // positions are kept relative to q[0]: the stream's bit 0 is at `lead`
  b->w = b->bit >> 6; b->hi = back_word(b, b->w); b->lo = back_word(b, b->w - 1); b->nx = back_word(b, b->w - 2);
  return true;
}
__device__ inline uint32_t back_get(Back* b, int64_t pos, int n) {    // bits [pos, pos + n), pos + n <= bit, n <= 32; positions below `lead` are zero
  while (pos < (b->w - 1) * 64) { b->hi = b->lo; b->lo = b->nx; b->w -= 1; b->nx = back_word(b, b->w - 2); }
  const int64_t rel = pos - (b->w - 1) * 64;                          // 0 .. 127 inside lo:hi
  uint64_t v;
  if (rel >= 64) v = b->hi >> (rel - 64);
  else v = rel ? (b->lo >> rel) | (b->hi << (64 - rel)) : b->lo;
  return (uint32_t)(v & ((1ull << n) - 1ull));
}
__device__ inline uint32_t back_read(Back* b, int n) { if (n <= 0) return 0; b->bit -= n; return back_get(b, b->bit, n); }
__device__ inline uint32_t back_peek(Back* b, int n) { return back_get(b, b->bit - n, n); }
__device__ inline int64_t back_left(const Back* b) { return b->bit - b->lead; }      // unread bits; negative = the stream ran out

// FSE table description (forward bits) -> norm[]; returns bytes consumed, 0 on error
__device__ inline uint32_t fse_read_norm(const uint8_t* s, uint32_t len, int max_log, int max_sym, int16_t* norm, int* out_log, int* out_nsym) {
  if (len < 1) return 0;
  int64_t bp = 0; const int64_t lim = (int64_t)len * 8;
  const int al = (int)bits_at(s, 0, 4) + 5; bp = 4; if (al > max_log) return 0;
  int remaining = (1 << al) + 1, threshold = 1 << al, nb = al + 1, sym = 0; bool prev0 = false;
  while (remaining > 1 && sym <= max_sym) {
    if (prev0) { int n0 = sym; for (;;) { if (bp + 2 > lim + 16) return 0; const uint32_t r = bits_at(s, bp, 2); bp += 2; n0 += (int)r; if (r != 3) break; }
      if (n0 > max_sym + 1) return 0; while (sym < n0) norm[sym++] = 0; if (sym > max_sym) break; }
    const int mx = (2 * threshold - 1) - remaining; const uint32_t v = bits_at(s, bp, nb); int count;
    if ((int)(v & (uint32_t)(threshold - 1)) < mx) { count = (int)(v & (uint32_t)(threshold - 1)); bp += nb - 1; }
    else { count = (int)(v & (uint32_t)(2 * threshold - 1)); if (count >= threshold) count -= mx; bp += nb; }
    count--;
    remaining -= count < 0 ? -count : count;
    norm[sym++] = (int16_t)count; prev0 = count == 0;
    while (remaining < threshold) { nb--; threshold >>= 1; }
  }
  if (remaining != 1 || bp > lim + 7) return 0;
  *out_log = al; *out_nsym = sym;
  return (uint32_t)((bp + 7) >> 3);
}
// norm[] -> decoding table of 1 << al entries
__device__ inline void fse_build(FseEnt* t, const int16_t* norm, int nsym, int al, uint16_t* next) {
  const int size = 1 << al; int high = size - 1;
  for (int s = 0; s < nsym; s++) { if (norm[s] == -1) { t[high--].sym = (uint8_t)s; next[s] = 1; } else next[s] = (uint16_t)norm[s]; }
  const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1; int pos = 0;
  for (int s = 0; s < nsym; s++) for (int i = 0; i < norm[s]; i++) { t[pos].sym = (uint8_t)s; pos = (pos + step) & mask; while (pos > high) pos = (pos + step) & mask; }
  for (int u = 0; u < size; u++) { const int s = t[u].sym; const int ns = next[s]++; const int nbits = al - (31 - __clz(ns)); t[u].nbits = (uint8_t)nbits; t[u].base = (uint16_t)((ns << nbits) - size); }
}
__device__ inline void fse_rle(FseEnt* t, uint8_t sym) { t[0].sym = sym; t[0].nbits = 0; t[0].base = 0; }

__device__ const int16_t ZS_LL_DEF[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__device__ const int16_t ZS_ML_DEF[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__device__ const int16_t ZS_OF_DEF[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
__device__ const uint32_t ZS_LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
__device__ const uint8_t ZS_LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__device__ const uint32_t ZS_ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34,
                                            35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
__device__ const uint8_t ZS_ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

// Huffman tree description at s -> L->huf, L->huf_log; returns bytes consumed, 0 on error.  Serial: call from one lane.
__device__ inline uint32_t huf_read_tree(Lds* L, const uint8_t* s, uint32_t len) {
  if (len < 1) return 0;
  const uint32_t hb = s[0]; uint32_t used; int nw = 0;
  if (hb >= 128) {                        // direct: 4 bits per weight
    nw = (int)hb - 127; used = 1 + (uint32_t)(nw + 1) / 2; if (used > len) return 0;
    for (int i = 0; i < nw; i++) { const uint8_t b = s[1 + i / 2]; L->weight[i] = (i & 1) ? (b & 15) : (b >> 4); }
  } else {                                // FSE-coded weights: table description, then a backward stream with two interleaved states
    used = 1 + hb; if (used > len || hb < 2) return 0;
    int al = 0, nsym = 0; const uint32_t h = fse_read_norm(s + 1, hb, 6, 255, L->norm, &al, &nsym); if (!h || h >= hb) return 0;
    fse_build(L->wt, L->norm, nsym, al, L->next);
    Back b; if (!back_init(&b, s + 1 + h, hb - h)) return 0;
    uint32_t s1 = back_read(&b, al), s2 = back_read(&b, al); if (back_left(&b) < 0) return 0;
    for (;;) {
      if (nw > 253) return 0;
      L->weight[nw++] = L->wt[s1].sym; s1 = L->wt[s1].base + back_read(&b, L->wt[s1].nbits);
      if (back_left(&b) < 0) { L->weight[nw++] = L->wt[s2].sym; break; }
      if (nw > 253) return 0;
      L->weight[nw++] = L->wt[s2].sym; s2 = L->wt[s2].base + back_read(&b, L->wt[s2].nbits);
      if (back_left(&b) < 0) { L->weight[nw++] = L->wt[s1].sym; break; }
    }
  }
  uint32_t total = 0; for (int i = 0; i < nw; i++) { if (L->weight[i] > ZS_HUF_LOG) return 0; if (L->weight[i]) total += 1u << (L->weight[i] - 1); }
  if (!total) return 0;
  const int maxbits = 32 - __clz((int)total); if (maxbits > ZS_HUF_LOG) return 0;          // floor(log2(total)) + 1
  const uint32_t rest = (1u << maxbits) - total; if (!rest || (rest & (rest - 1))) return 0;
  L->weight[nw++] = (uint8_t)(32 - __clz((int)rest));                                       // log2(rest) + 1: the last symbol's weight is implied
  uint32_t pos = 0;
  for (int w = 1; w <= maxbits; w++) for (int sym = 0; sym < nw; sym++) if (L->weight[sym] == w) {
    const uint32_t n = 1u << (w - 1); const uint16_t e = (uint16_t)(sym | ((maxbits + 1 - w) << 8));
    for (uint32_t i = 0; i < n; i++) L->huf[pos + i] = e; pos += n; }
  if (pos != (1u << maxbits)) return 0;
  L->huf_log = maxbits; L->huf_ok = 1;
  return used;
}
// one Huffman stream -> n symbols; false on error.  Serial per stream (the bit reader's register window serves 5-8 symbols per refill).
__device__ inline bool huf_stream(const Lds* L, const uint8_t* s, uint32_t len, uint8_t* out, uint32_t n) {
  Back b; if (!back_init(&b, s, len)) return false; const int hl = L->huf_log;
  for (uint32_t i = 0; i < n; i++) { const uint16_t e = L->huf[back_peek(&b, hl)]; out[i] = (uint8_t)e; b.bit -= e >> 8; }
  return back_left(&b) == 0;
}

// one of the three sequence tables (kind 0 literal lengths, 1 offsets, 2 match lengths): mode 0 predefined, 1 RLE, 2 FSE description, 3 repeat; advances *p.  Serial: one lane.
__device__ inline bool seq_table(Lds* L, int kind, int mode, const uint8_t** p, const uint8_t* end, SeqEnt* t, int32_t* log, const int16_t* def, int def_n, int def_log, int max_log, int max_sym) {
  int al;
  if (mode == 0) { for (int i = 0; i < def_n; i++) L->norm[i] = def[i]; fse_build(L->tmp, L->norm, def_n, def_log, L->next); al = def_log; }
  else if (mode == 1) { if (*p >= end || **p > max_sym) return false; fse_rle(L->tmp, **p); (*p)++; al = 0; }
  else if (mode == 2) { int ns = 0; const uint32_t h = fse_read_norm(*p, (uint32_t)(end - *p), max_log, max_sym, L->norm, &al, &ns); if (!h) return false; fse_build(L->tmp, L->norm, ns, al, L->next); *p += h; }
  else return *log >= 0;                  // repeat: a table must exist
  for (int u = 0; u < (1 << al); u++) { const FseEnt e = L->tmp[u]; if (e.sym > max_sym) return false;
    SeqEnt o; o.base = e.base; o.nbits = e.nbits;
    if (kind == 1) { o.addbits = e.sym; o.value = 1u << e.sym; } else if (kind == 0) { o.addbits = ZS_LL_BITS[e.sym]; o.value = ZS_LL_BASE[e.sym]; } else { o.addbits = ZS_ML_BITS[e.sym]; o.value = ZS_ML_BASE[e.sym]; }
    t[u] = o; }
  *log = al; return true;
}

constexpr uint32_t ZS_LIT_LDS = 65536;
constexpr uint32_t ZS_RING = 32768;       // the last 32 KB of output also live in LDS: a match that ends inside them is read there (LDS operations of a wave execute in order), no
                                          // store -> load round trip through the memory system per sequence; farther matches wait for the wave's stores and read HBM
// Decode one frame src[0, csize) into dst[0, usize).  ring = ZS_RING bytes of LDS; a block's literals are decoded into lit_lds (ZS_LIT_LDS bytes of LDS: every sequence then
// takes its literals at LDS latency) or, when a block regenerates more than that, into lit_hbm (128 KB + 32 of scratch for this wave).  true when exactly usize bytes came out.
// Every value that steers control flow is the same in all 64 lanes; `lane` only selects which bytes a lane moves.
__device__ inline bool decode_frame(Lds* L, uint8_t* ring, uint8_t* lit_lds, const uint8_t* src, uint32_t csize, uint8_t* dst, uint32_t usize, uint8_t* lit_hbm, uint32_t lane) {
  constexpr uint32_t RM = ZS_RING - 1;
  if (csize < 6) return false;
  if (src[0] != 0x28u || src[1] != 0xB5u || src[2] != 0x2Fu || src[3] != 0xFDu) return false;
  const uint32_t fhd = src[4]; const uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3; if (fhd & 8) return false;
  uint32_t p = 5; if (!single) p += 1;
  if (did) return false;                                     // dictionaries are not part of a Parquet page
  const uint32_t fcs_bytes = fcs_flag == 0 ? (single ? 1u : 0u) : fcs_flag == 1 ? 2u : fcs_flag == 2 ? 4u : 8u;
  p += fcs_bytes; if (p > csize) return false;
  uint32_t rep1 = 1, rep2 = 4, rep3 = 8, out = 0, fenced = 0;
#ifdef ZS_PROFILE
  long long t_lit = 0, t_tab = 0, t_seq = 0, t_mark = wall_clock64();        // -DZS_PROFILE: time per phase of job 0, printed by dfgpu_parquet_read
#define ZS_T(acc) { const long long now_ = wall_clock64(); acc += now_ - t_mark; t_mark = now_; }
#else
#define ZS_T(acc)
#endif
  __shared__ uint32_t sh_bad, sh_used; __shared__ const uint8_t* sh_ptr;
  if (lane == 0) { L->ll_log = L->ml_log = L->of_log = -1; L->huf_ok = 0; }
  __syncthreads();
  // Output goes to the LDS ring only; the ring is copied to HBM a few KB at a time with 16-byte stores (no byte stores, and no store between the loads of the bit reader).
  // Measured with -DZS_PROFILE on a 1 MB page of sorted Int64 keys (131 462 sequences, one per value): ~128 ms in the sequence loop, ~1 us per sequence, split evenly over
  // decoding the sequence, copying its literals and copying its match, and the same with per-sequence stores, with literals in HBM, with table constants in global memory:
  // the compiler keeps the (uniform) parse on the vector unit behind exec masks -- ~1000 instructions per sequence on a lone wave.  Variants tried and dropped because they
  // were slower on the device: a reader with one window move per group of reads and v_readfirstlane on everything loaded (129 ms), LZ4 sharing this kernel (159 ms).
  uint32_t flushed = 0;
  const bool dst16 = ((uintptr_t)dst & 15) == 0;
  auto flush = [&](uint32_t upto, bool all) {                  // ring[flushed, upto) -> dst; without `all` only whole 16-byte groups leave
    __builtin_amdgcn_wave_barrier();
    uint32_t a = flushed;
    if (!dst16) { if (all || upto - a >= 4096) { for (uint32_t q = a + lane; q < upto; q += 64) dst[q] = ring[q & RM]; flushed = upto; } return; }
    const uint32_t a16 = (a + 15u) & ~15u, e16 = upto & ~15u;
    if (a16 > a) { const uint32_t h = a16 < upto ? a16 : upto; for (uint32_t q = a + lane; q < h; q += 64) dst[q] = ring[q & RM]; a = h; }
    if (e16 > a) { for (uint32_t q = a + lane * 16; q < e16; q += 1024) *(uint4*)(dst + q) = *(const uint4*)(ring + (q & RM)); a = e16; }
    if (all && upto > a) { for (uint32_t q = a + lane; q < upto; q += 64) dst[q] = ring[q & RM]; a = upto; }
    flushed = a;
  };
  constexpr uint32_t CH = 8192;                                // at most this much is written between two looks at the ring's fill
  auto room = [&](uint32_t c) { if (out + c - flushed > ZS_RING - 64) flush(out, false); };
  auto put_copy = [&](const uint8_t* from, uint32_t n) {      // n bytes from `from` (input or literal buffer) to the output
    while (n) { const uint32_t c = n < CH ? n : CH; room(c);
      for (uint32_t i = lane; i < c; i += 64) ring[(out + i) & RM] = from[i];
      out += c; from += c; n -= c; }
  };
  auto put_fill = [&](uint8_t v, uint32_t n) {
    while (n) { const uint32_t c = n < CH ? n : CH; room(c); for (uint32_t i = lane; i < c; i += 64) ring[(out + i) & RM] = v; out += c; n -= c; }
  };
  for (;;) {
    if (p + 3 > csize) return false;
    const uint32_t bh = (uint32_t)src[p] | ((uint32_t)src[p + 1] << 8) | ((uint32_t)src[p + 2] << 16); p += 3;
    const uint32_t last = bh & 1, btype = (bh >> 1) & 3, bsize = bh >> 3;
    if (btype == 0) { if (p + bsize > csize || out + bsize > usize) return false; put_copy(src + p, bsize); p += bsize; }
    else if (btype == 1) { if (p + 1 > csize || out + bsize > usize) return false; put_fill(src[p], bsize); p += 1; }
    else if (btype == 2) {
      if (bsize > (uint32_t)ZS_BLOCK_MAX || p + bsize > csize || bsize < 2) return false;
      const uint8_t* b = src + p; const uint8_t* const bend = b + bsize; p += bsize;
      ZS_T(t_tab)
      // ---- literals section
      const uint32_t lt = b[0] & 3, sf = (b[0] >> 2) & 3; uint32_t regen = 0, comp = 0, hdr = 0, streams = 1;
      if (lt < 2) {
        if (sf == 0 || sf == 2) { regen = b[0] >> 3; hdr = 1; }
        else if (sf == 1) { regen = ((uint32_t)b[0] >> 4) | ((uint32_t)b[1] << 4); hdr = 2; }
        else { if (bsize < 3) return false; regen = ((uint32_t)b[0] >> 4) | ((uint32_t)b[1] << 4) | ((uint32_t)b[2] << 12); hdr = 3; }
      } else {
        if (bsize < 3) return false;
        const uint64_t h = (uint64_t)b[0] | ((uint64_t)b[1] << 8) | ((uint64_t)b[2] << 16) | ((uint64_t)(bsize > 3 ? b[3] : 0) << 24) | ((uint64_t)(bsize > 4 ? b[4] : 0) << 32);
        if (sf == 0) { regen = (uint32_t)(h >> 4) & 0x3FF; comp = (uint32_t)(h >> 14) & 0x3FF; hdr = 3; streams = 1; }
        else if (sf == 1) { regen = (uint32_t)(h >> 4) & 0x3FF; comp = (uint32_t)(h >> 14) & 0x3FF; hdr = 3; streams = 4; }
        else if (sf == 2) { regen = (uint32_t)(h >> 4) & 0x3FFF; comp = (uint32_t)(h >> 18) & 0x3FFF; hdr = 4; streams = 4; }
        else { regen = (uint32_t)(h >> 4) & 0x3FFFF; comp = (uint32_t)(h >> 22) & 0x3FFFF; hdr = 5; streams = 4; }
      }
      if (regen > (uint32_t)ZS_BLOCK_MAX || b + hdr > bend) return false;
      uint8_t* const lit = regen <= ZS_LIT_LDS ? lit_lds : lit_hbm;
      const uint8_t* lsrc = lit; b += hdr;
      if (lt == 0) { if (b + regen > bend) return false; if (regen <= ZS_LIT_LDS) { for (uint32_t i = lane; i < regen; i += 64) lit[i] = b[i]; } else lsrc = b; b += regen; }      // raw: staged in LDS, or read in place
      else if (lt == 1) { if (b + 1 > bend) return false; const uint8_t v = b[0]; for (uint32_t i = lane; i < regen; i += 64) lit[i] = v; b += 1; }
      else {
        if (b + comp > bend) return false;
        const uint8_t* hs = b; uint32_t hl = comp; b += comp;
        if (lane == 0) { sh_bad = 0; sh_used = 0; if (lt == 2) { sh_used = huf_read_tree(L, hs, hl); if (!sh_used) sh_bad = 1; } else if (!L->huf_ok) sh_bad = 1; }
        __syncthreads();
        if (sh_bad) return false;
        hs += sh_used; hl -= sh_used;
        __syncthreads();
        if (streams == 1) { if (lane == 0 && !huf_stream(L, hs, hl, lit, regen)) sh_bad = 1; }
        else {
          if (hl < 6) return false;
          const uint32_t l1 = (uint32_t)hs[0] | ((uint32_t)hs[1] << 8), l2 = (uint32_t)hs[2] | ((uint32_t)hs[3] << 8), l3 = (uint32_t)hs[4] | ((uint32_t)hs[5] << 8);
          if (6 + l1 + l2 + l3 > hl) return false;
          const uint32_t l4 = hl - 6 - l1 - l2 - l3, per = (regen + 3) / 4; if (3 * per > regen) return false;
          if (lane < 4) {
            const uint8_t* ss = hs + 6 + (lane > 0 ? l1 : 0) + (lane > 1 ? l2 : 0) + (lane > 2 ? l3 : 0);
            const uint32_t sl = lane == 0 ? l1 : lane == 1 ? l2 : lane == 2 ? l3 : l4, cnt = lane < 3 ? per : regen - 3 * per;
            if (!huf_stream(L, ss, sl, lit + lane * per, cnt)) sh_bad = 1;
          }
        }
        __syncthreads();
        if (sh_bad) return false;
      }
      __syncthreads();                                      // literals written by some lanes are read by all
      ZS_T(t_lit)
      // ---- sequences section
      if (b >= bend) return false;
      uint32_t nseq = b[0]; b += 1;
      if (nseq >= 128) { if (nseq < 255) { if (b >= bend) return false; nseq = ((nseq - 128) << 8) + b[0]; b += 1; } else { if (b + 2 > bend) return false; nseq = (uint32_t)b[0] + ((uint32_t)b[1] << 8) + 0x7F00; b += 2; } }
      uint32_t lpos = 0;
      if (nseq) {
        if (b >= bend) return false;
        const uint32_t modes = b[0]; b += 1; if (modes & 3) return false;
        if (lane == 0) { const uint8_t* q = b; bool ok = seq_table(L, 0, (modes >> 6) & 3, &q, bend, L->ll, &L->ll_log, ZS_LL_DEF, 36, 6, ZS_LL_LOG, 35);
          ok = ok && seq_table(L, 1, (modes >> 4) & 3, &q, bend, L->of, &L->of_log, ZS_OF_DEF, 29, 5, ZS_OF_LOG, 31);
          ok = ok && seq_table(L, 2, (modes >> 2) & 3, &q, bend, L->ml, &L->ml_log, ZS_ML_DEF, 53, 6, ZS_ML_LOG, 52);
          sh_bad = ok ? 0u : 1u; sh_ptr = q; }
        __syncthreads();
        if (sh_bad) return false;
        b = sh_ptr;
        __syncthreads();
        ZS_T(t_tab)
        Back s; if (b >= bend || !back_init(&s, b, (uint32_t)(bend - b))) return false;
        uint32_t sl = back_read(&s, L->ll_log), so = back_read(&s, L->of_log), sm = back_read(&s, L->ml_log);
        if (back_left(&s) < 0) return false;
        for (uint32_t i = 0; i < nseq; i++) {
          const SeqEnt el = L->ll[sl], eo = L->of[so], em = L->ml[sm];
          const uint32_t ov = eo.value + back_read(&s, eo.addbits);
          const uint32_t mlen = em.value + back_read(&s, em.addbits);
          const uint32_t llen = el.value + back_read(&s, el.addbits);
          if (i + 1 < nseq) { sl = el.base + back_read(&s, el.nbits); sm = em.base + back_read(&s, em.nbits); so = eo.base + back_read(&s, eo.nbits); }
          if (back_left(&s) < 0) return false;
          uint32_t off;
          if (ov > 3) { off = ov - 3; rep3 = rep2; rep2 = rep1; rep1 = off; }
          else { const uint32_t idx = ov + (llen == 0 ? 1u : 0u);
            if (idx == 1) off = rep1;
            else if (idx == 2) { off = rep2; rep2 = rep1; rep1 = off; }
            else if (idx == 3) { off = rep3; rep3 = rep2; rep2 = rep1; rep1 = off; }
            else { off = rep1 - 1; if (!off) return false; rep3 = rep2; rep2 = rep1; rep1 = off; } }
          if (lpos + llen > regen || out + llen + mlen > usize || off > out + llen) return false;
          if (llen) { put_copy(lsrc + lpos, llen); lpos += llen; }
          __builtin_amdgcn_wave_barrier();
          if (off + CH <= ZS_RING) {              // the source of every chunk lies inside the ring
            uint32_t n = mlen;
            while (n) { const uint32_t c = n < CH ? n : CH; room(c); const uint32_t from = out - off;
              if (off >= c) { for (uint32_t k = lane; k < c; k += 64) ring[(out + k) & RM] = ring[(from + k) & RM]; }
              else for (uint32_t k = lane; k < c; k += 64) { const uint8_t v = ring[(from + k % off) & RM]; __builtin_amdgcn_wave_barrier(); ring[(out + k) & RM] = v; }
              out += c; n -= c; __builtin_amdgcn_wave_barrier(); }
          } else {                                // further back than the ring: everything written so far leaves for HBM, one wait, then the source is read there
            if (out - off + (mlen < off ? mlen : off) > fenced) { flush(out, true); __threadfence_block(); fenced = out; }      // one wait covers everything written so far
            const uint32_t o0 = out; uint32_t n = mlen, done = 0;                    // source positions are all in front of o0: nothing this match writes is read back
            while (n) { const uint32_t c = n < CH ? n : CH; room(c);
              for (uint32_t k = lane; k < c; k += 64) { const uint32_t j = done + k; ring[(out + k) & RM] = dst[o0 - off + (off >= mlen ? j : j % off)]; }
              out += c; done += c; n -= c; }
          }
          __builtin_amdgcn_wave_barrier();
        }
        if (back_left(&s) != 0) return false;
        ZS_T(t_seq)
      }
      const uint32_t restl = regen - lpos; if (out + restl > usize) return false;
      put_copy(lsrc + lpos, restl);
      __syncthreads();
    } else return false;
    if (last) break;
  }
  flush(out, true);
#ifdef ZS_PROFILE
  if (lane == 0) { long long* d = (long long*)lit_hbm; d[0] = t_lit; d[1] = t_tab; d[2] = t_seq; d[3] = (long long)usize; }
#endif
  return out == usize;
}


// ---- LZ4 block format (Parquet codec LZ4_RAW): token (literal length : match length - 4), extension bytes of 255, literals, 2-byte offset.  One wave per page like the
// Zstandard frames above and for the same reason (every match may reach back to the start of the page); the input is parsed out of a 16 KB LDS window, output goes to
// the same kind of LDS ring, flushed to HBM in 16-byte stores.
__device__ inline bool lz4_decode(uint8_t* ring, uint8_t* win, const uint8_t* src, uint32_t csize, uint8_t* dst, uint32_t usize, uint32_t lane) {
  constexpr uint32_t RM = ZS_RING - 1, WIN = 16384, CH = 8192;
  uint32_t out = 0, flushed = 0, fenced = 0, ip = 0, wbase = 0, wlen = 0;
  const bool dst16 = ((uintptr_t)dst & 15) == 0;
  auto flush = [&](uint32_t upto, bool all) {
    __builtin_amdgcn_wave_barrier();
    uint32_t a = flushed;
    if (!dst16) { if (all || upto - a >= 4096) { for (uint32_t q = a + lane; q < upto; q += 64) dst[q] = ring[q & RM]; flushed = upto; } return; }
    const uint32_t a16 = (a + 15u) & ~15u, e16 = upto & ~15u;
    if (a16 > a) { const uint32_t h = a16 < upto ? a16 : upto; for (uint32_t q = a + lane; q < h; q += 64) dst[q] = ring[q & RM]; a = h; }
    if (e16 > a) { for (uint32_t q = a + lane * 16; q < e16; q += 1024) *(uint4*)(dst + q) = *(const uint4*)(ring + (q & RM)); a = e16; }
    if (all && upto > a) { for (uint32_t q = a + lane; q < upto; q += 64) dst[q] = ring[q & RM]; a = upto; }
    flushed = a;
  };
  auto room = [&](uint32_t c) { if (out + c - flushed > ZS_RING - 64) flush(out, false); };
  auto need = [&](uint32_t n) {                    // input [ip, ip + n) readable from the window (n <= WIN)
    if (ip + n > wbase + wlen) { __builtin_amdgcn_wave_barrier(); wbase = ip; wlen = csize - ip < WIN ? csize - ip : WIN;
      for (uint32_t i = lane; i < wlen; i += 64) win[i] = src[ip + i];
      __builtin_amdgcn_wave_barrier(); }
  };
  if (!csize) return usize == 0;
  for (;;) {
    if (ip >= csize) return false;
    need(csize - ip < 64 ? csize - ip : 64);
    const uint32_t token = win[ip - wbase]; ip += 1;
    uint32_t ll = token >> 4;
    if (ll == 15) { for (;;) { if (ip >= csize) return false; need(1); const uint32_t b = win[ip - wbase]; ip += 1; ll += b; if (b != 255) break; } }
    if (ip + ll > csize || out + ll > usize) return false;
    { uint32_t n = ll, at = ip;                      // literals: short ones out of the window, long ones straight from the page
      while (n) { const uint32_t c = n < CH ? n : CH; room(c);
        if (at >= wbase && at + c <= wbase + wlen) { for (uint32_t i = lane; i < c; i += 64) ring[(out + i) & RM] = win[at - wbase + i]; }
        else for (uint32_t i = lane; i < c; i += 64) ring[(out + i) & RM] = src[at + i];
        out += c; at += c; n -= c; } }
    ip += ll;
    if (ip >= csize) break;                          // the last sequence is literals only
    if (ip + 2 > csize) return false;
    need(2);
    const uint32_t off = (uint32_t)win[ip - wbase] | ((uint32_t)win[ip - wbase + 1] << 8); ip += 2;
    uint32_t ml = token & 15;
    if (ml == 15) { for (;;) { if (ip >= csize) return false; need(1); const uint32_t b = win[ip - wbase]; ip += 1; ml += b; if (b != 255) break; } }
    ml += 4;
    if (!off || off > out || out + ml > usize) return false;
    __builtin_amdgcn_wave_barrier();
    if (off + CH <= ZS_RING) {
      uint32_t n = ml;
      while (n) { const uint32_t c = n < CH ? n : CH; room(c); const uint32_t from = out - off;
        if (off >= c) { for (uint32_t k = lane; k < c; k += 64) ring[(out + k) & RM] = ring[(from + k) & RM]; }
        else for (uint32_t k = lane; k < c; k += 64) { const uint8_t v = ring[(from + k % off) & RM]; __builtin_amdgcn_wave_barrier(); ring[(out + k) & RM] = v; }
        out += c; n -= c; __builtin_amdgcn_wave_barrier(); }
    } else {
      if (out - off + (ml < off ? ml : off) > fenced) { flush(out, true); __threadfence_block(); fenced = out; }
      const uint32_t o0 = out; uint32_t n = ml, done = 0;
      while (n) { const uint32_t c = n < CH ? n : CH; room(c);
        for (uint32_t k = lane; k < c; k += 64) { const uint32_t j = done + k; ring[(out + k) & RM] = dst[o0 - off + (off >= ml ? j : j % off)]; }
        out += c; done += c; n -= c; }
    }
    __builtin_amdgcn_wave_barrier();
  }
  flush(out, true);
  return out == usize;
}

}  // namespace zs
}  // namespace dfgpu
