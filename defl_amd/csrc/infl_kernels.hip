// infl_kernels.hip -- DEFLATE / zlib block decode for gfx950 (MI355X, CDNA4).
//
// One group of T lanes of a wavefront (T = 64, 32, 16 or 8) decodes one
// independent stream with the semantics of recp/defl's one-shot infl() on a
// single contiguous input (reference src/infl/infl.c:1507-1551 dispatcher,
// :94-267 stored-only path, :1002-1104 block loop, :903-1000 dynamic header,
// :452-578 table build, :717-901 symbol loop).  Nothing here is translated from
// that file: the reference is a scalar CPU loop over a 64-bit reservoir, this is
// a SIMT kernel whose lanes share the decode tables in LDS, replicate the bit
// reservoir in registers, build the tables cooperatively and copy LZ77 matches
// and stored blocks with one byte (or 16 bytes) per lane.
//
// All arithmetic is integer; the only floating point instruction class that
// hipcc could pack is therefore absent, but the file is still compiled with
// packed-fp32 ops disabled (see the build script).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "infl_device.h"

namespace defl {

// ---------------------------------------------------------------------------
// Decode-table entry, same packing as the reference's fast tables
// (infl.c:385-397): bits 0-4 total = codelen + xbits, 5-8 codelen, 9-12 xbits,
// 13 LITERAL, 14 END, 15 SUBTABLE, 16-31 base value or subtable offset.
// ---------------------------------------------------------------------------
constexpr uint32_t E_LITERAL = 1u << 13;
constexpr uint32_t E_END     = 1u << 14;
constexpr uint32_t E_SUB     = 1u << 15;

constexpr int LIT_BITS  = 10;     // infl.c:378
constexpr int DIST_BITS = 8;      // infl.c:379
constexpr int LIT_CAP   = 2048;   // infl.c:382
constexpr int DIST_CAP  = 512;    // infl.c:383

__device__ __forceinline__ uint32_t mk_entry(uint32_t base, uint32_t xbits, uint32_t codelen,
                                             uint32_t flags) {
  return (base << 16) | flags | (xbits << 9) | (codelen << 5) | (codelen + xbits);
}

// RFC 1951 length / distance bases computed instead of looked up
// (values of apicommon.h:36-49).
__device__ __forceinline__ uint32_t litlen_entry(uint32_t sym, uint32_t len) {
  if (sym < 256) return mk_entry(sym, 0, len, E_LITERAL);
  if (sym == 256) return mk_entry(0, 0, len, E_END);
  if (sym > 285) return 0;                         // 286, 287 never get an entry
  uint32_t i = sym - 257;
  if (i < 8) return mk_entry(3 + i, 0, len, 0);
  if (i == 28) return mk_entry(258, 0, len, 0);
  uint32_t xb = (i - 4) >> 2;
  return mk_entry(3 + ((4 + (i & 3)) << xb), xb, len, 0);
}

__device__ __forceinline__ uint32_t dist_entry(uint32_t sym, uint32_t len) {
  if (sym > 29) return 0;                          // 30, 31 never get an entry
  if (sym < 4) return mk_entry(1 + sym, 0, len, 0);
  uint32_t xb = (sym >> 1) - 1;
  return mk_entry(1 + ((2 + (sym & 1)) << xb), xb, len, 0);
}

// Per-stream working set in LDS.
struct StreamLds {
  uint32_t lit[LIT_CAP];      // 8 KiB
  uint32_t dist[DIST_CAP];    // 2 KiB
  uint16_t rev[320];          // bit-reversed canonical code per symbol; the
                              // code-length alphabet's 256-entry table lives
                              // here while the header is being read
  uint8_t  lens[320];         // code lengths of the block being set up
  uint32_t count[16];
  uint32_t next[16];
  uint32_t used;
};

// Lanes of a group run in lockstep inside one wavefront; LDS and global
// operations of a wavefront are issued in program order, so the only thing a
// group-level "barrier" has to stop is the compiler moving memory operations.
__device__ __forceinline__ void group_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------
// Bit reader.  Every lane of the group keeps an identical copy: 64-bit
// LSB-first reservoir that holds only real input bits (everything above `n` is
// zero, like the reference's reservoir, infl.c:580-633) and one prefetched
// 32-bit word so that the global load latency overlaps with symbol decode.
// ---------------------------------------------------------------------------
struct Reader {
  const uint8_t *p;      // next byte not yet handed to `nextw`'s successor
  const uint8_t *end;
  uint64_t       bits;
  uint32_t       n;
  uint32_t       nextw;  // bytes p[0..3] (zero padded at the end of input)
};

__device__ __forceinline__ uint32_t load_word(const uint8_t *p, const uint8_t *end) {
  size_t k = (size_t)(end - p);
  if (k >= 4) {
    uint32_t w;
    __builtin_memcpy(&w, p, 4);                    // unaligned global_load_dword
    return w;
  }
  uint32_t w = 0;
  for (size_t i = 0; i < k; i++) w |= (uint32_t)p[i] << (8 * i);
  return w;
}

__device__ __forceinline__ void rd_seek(Reader &r, const uint8_t *p) {
  r.p     = p;
  r.bits  = 0;
  r.n     = 0;
  r.nextw = load_word(p, r.end);
}

// After this call n >= 32 unless the input is exhausted (then n = all that is left).
__device__ __forceinline__ void rd_fill(Reader &r) {
  if (r.n < 32) {
    size_t k = (size_t)(r.end - r.p);
    if (k > 4) k = 4;
    if (k) {
      r.bits |= (uint64_t)r.nextw << r.n;
      r.n    += (uint32_t)k * 8;
      r.p    += k;
      r.nextw = load_word(r.p, r.end);
    }
  }
}

__device__ __forceinline__ void rd_drop(Reader &r, uint32_t k) {
  r.bits >>= k;
  r.n    -= k;
}

// byte offset (from `base`) of the first byte that still has unread bits
__device__ __forceinline__ const uint8_t *rd_byte_ptr(const Reader &r) {
  return r.p - (r.n >> 3);
}

__device__ __forceinline__ uint32_t tab_lookup(const uint32_t *tab, int mainbits, uint32_t bits) {
  uint32_t e = tab[bits & ((1u << mainbits) - 1u)];
  if (e & E_SUB)
    e = tab[(e >> 16) + ((bits >> mainbits) & ((1u << ((e >> 5) & 15u)) - 1u))];
  return e;
}

// ---------------------------------------------------------------------------
// Cooperative copies
// ---------------------------------------------------------------------------
template <int T>
__device__ __forceinline__ void coop_copy(uint8_t *d, const uint8_t *s, uint32_t len, int lig) {
  uint32_t body = len & ~15u;
  for (uint32_t k = (uint32_t)lig * 16; k < body; k += T * 16) {
    uint4 v;
    __builtin_memcpy(&v, s + k, 16);
    __builtin_memcpy(d + k, &v, 16);
  }
  for (uint32_t k = body + lig; k < len; k += T) d[k] = s[k];
}

// dst[pos + i] = dst[pos - dist + i] for i < len with sequential semantics
// (what the copy kernels of infl.c:269-376 compute).  Every source byte lies
// in [pos - dist, pos), which earlier instructions of this wavefront wrote.
template <int T>
__device__ __forceinline__ void coop_match(uint8_t *dst, uint32_t pos, uint32_t dist, uint32_t len,
                                           int lig) {
  const uint8_t *s = dst + pos - dist;
  uint8_t       *d = dst + pos;
  if (dist >= len) {
    for (uint32_t k = lig; k < len; k += T) d[k] = s[k];
  } else {
    uint32_t m    = (uint32_t)lig % dist;          // source index of byte k is k mod dist
    uint32_t step = (uint32_t)T % dist;
    for (uint32_t k = lig; k < len; k += T) {
      d[k] = s[m];
      m += step;
      if (m >= dist) m -= dist;
    }
  }
}

// ---------------------------------------------------------------------------
// Two-level table build for one alphabet (acceptance rules of infl.c:452-578:
// a length set is rejected when over-subscribed or when main table plus
// subtables exceed `cap`; incomplete sets are accepted; symbols without an
// entry still take part in code assignment and subtable sizing).
// `lens` is in LDS and complete; returns false on rejection (group-uniform).
// ---------------------------------------------------------------------------
template <int T>
__device__ __noinline__ bool build_table(uint32_t *table, const uint8_t *lens, uint32_t nsyms, int tablebits,
                            uint32_t cap, bool litlen, StreamLds &L, int lig, int gshift) {
  const uint32_t main_size = 1u << tablebits;
  const uint64_t gmask     = (T == 64) ? ~0ull : (((1ull << T) - 1ull) << gshift);

  group_sync();
  for (uint32_t i = lig; i < cap; i += T) table[i] = 0;
  for (int i = lig; i < 16; i += T) L.count[i] = 0;
  if (lig == 0) L.used = main_size;
  group_sync();

  for (uint32_t s = lig; s < nsyms; s += T) atomicAdd(&L.count[lens[s]], 1u);
  group_sync();

  {
    int      left = 1;
    uint32_t code = 0, prev_count = 0;             // count[0] is left out: canonical codes
    for (int l = 1; l <= 15; l++) {
      uint32_t c = L.count[l];
      left = (left << 1) - (int)c;
      if (left < 0) return false;
      code = (code + prev_count) << 1;
      prev_count = c;
      if (lig == 0) L.next[l] = code;
    }
  }
  group_sync();

  // canonical codes in symbol order, T symbols per round
  for (uint32_t base = 0; base < nsyms; base += T) {
    uint32_t s   = base + lig;
    uint32_t len = s < nsyms ? lens[s] : 0;
    uint32_t rank = 0, cnt = 0;
    for (uint32_t l = 1; l <= 15; l++) {
      uint64_t m = __ballot(len == l) & gmask;
      if (len == l) {
        rank = __popcll(m & ((1ull << (gshift + lig)) - 1ull));
        cnt  = __popcll(m);
      }
    }
    uint32_t code = 0;
    if (len) code = L.next[len] + rank;
    group_sync();
    if (len) {
      if (rank + 1 == cnt) L.next[len] = code + 1;
      uint32_t r = __brev(code) >> (32 - len);
      L.rev[s] = (uint16_t)r;
      if (len > (uint32_t)tablebits)               // subtable width wanted under this prefix
        atomicMax(&table[r & (main_size - 1u)], len - tablebits);
    }
    group_sync();
  }

  // turn the widths into subtable pointers
  for (uint32_t p = lig; p < main_size; p += T) {
    uint32_t w = table[p];
    if (w) {
      uint32_t at = atomicAdd(&L.used, 1u << w);
      table[p] = (at << 16) | E_SUB | (w << 5) | (uint32_t)tablebits;
    }
  }
  group_sync();
  if (L.used > cap) return false;

  for (uint32_t s = lig; s < nsyms; s += T) {
    uint32_t len = lens[s];
    if (!len) continue;
    uint32_t e = litlen ? litlen_entry(s, len) : dist_entry(s, len);
    if (!e) continue;
    uint32_t r = L.rev[s];
    if (len <= (uint32_t)tablebits) {
      for (uint32_t i = r; i < main_size; i += 1u << len) table[i] = e;
    } else {
      uint32_t me  = table[r & (main_size - 1u)];
      uint32_t at  = me >> 16;
      uint32_t end = 1u << ((me >> 5) & 15u);
      for (uint32_t i = r >> tablebits; i < end; i += 1u << (len - tablebits)) table[at + i] = e;
    }
  }
  group_sync();
  return true;
}

// 256-entry direct table for the 19-symbol code-length alphabet: the job the
// reference gives to recp/huff's huff_init_fast_lsb (call site infl.c:938).
// Entry = sym << 8 | len, 0 = no code.  `packed` holds the 19 3-bit lengths.
// Rejects an over-subscribed set, accepts an incomplete one.
template <int T>
__device__ __noinline__ bool build_clen_table(uint16_t *tab, uint64_t packed, int lig) {
  uint32_t cnt[8];
#pragma unroll
  for (int l = 0; l < 8; l++) cnt[l] = 0;
#pragma unroll
  for (int s = 0; s < 19; s++) {
    uint32_t l = (uint32_t)(packed >> (3 * s)) & 7u;
#pragma unroll
    for (int j = 1; j < 8; j++) cnt[j] += (l == (uint32_t)j);
  }
  int      left = 1;
  uint32_t first[8];
  uint32_t code = 0, prev = 0;
  first[0] = 0;
#pragma unroll
  for (int l = 1; l < 8; l++) {
    left = (left << 1) - (int)cnt[l];
    if (left < 0) return false;
    code = (code + prev) << 1;
    prev = cnt[l];
    first[l] = code;
  }

  group_sync();
  for (int i = lig; i < 128; i += T) ((uint32_t *)tab)[i] = 0;
  group_sync();
  for (int s = lig; s < 19; s += T) {
    uint32_t len = (uint32_t)(packed >> (3 * s)) & 7u;
    if (!len) continue;
    uint32_t c = 0;
#pragma unroll
    for (int j = 1; j < 8; j++) c = (len == (uint32_t)j) ? first[j] : c;
    for (int q = 0; q < s; q++) c += (((uint32_t)(packed >> (3 * q)) & 7u) == len);
    uint32_t r = __brev(c) >> (32 - len);
    for (uint32_t i = r; i < 256; i += 1u << len) tab[i] = (uint16_t)((s << 8) | len);
  }
  group_sync();
  return true;
}

// Fixed-Huffman code lengths (apicommon.h:53-64) into L.lens, then both tables.
template <int T>
__device__ __noinline__ bool setup_fixed(StreamLds &L, int lig, int gshift) {
  group_sync();
  for (int s = lig; s < 320; s += T)
    L.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5;
  return build_table<T>(L.lit, L.lens, 288, LIT_BITS, LIT_CAP, true, L, lig, gshift) &&
         build_table<T>(L.dist, L.lens + 288, 32, DIST_BITS, DIST_CAP, false, L, lig, gshift);
}

// Dynamic block header (rules of infl.c:903-1000, including "at least 7 bits
// must be left before every code-length symbol", :947).
template <int T>
__device__ __forceinline__ int setup_dynamic(Reader &r, StreamLds &L, int lig, int gshift) {
  rd_fill(r);
  if (r.n < 14) return INFL_D_NEED_INPUT;
  uint32_t hlit  = ((uint32_t)r.bits & 31u) + 257u;
  uint32_t hdist = (((uint32_t)r.bits >> 5) & 31u) + 1u;
  uint32_t hclen = (((uint32_t)r.bits >> 10) & 15u) + 4u;
  uint32_t n     = hlit + hdist;
  rd_drop(r, 14);

  // code-length alphabet order 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15
  // (apicommon.h:52): the 16 entries after the first three, one per nibble
  const uint64_t order_tail = 0xF1E2D3C4B5A69780ull;
  uint64_t packed = 0;
  for (uint32_t i = 0; i < hclen; i++) {
    uint32_t sym = i < 3 ? 16u + i : (uint32_t)(order_tail >> (4 * (i - 3))) & 15u;
    rd_fill(r);
    if (r.n < 3) return INFL_D_NEED_INPUT;
    packed |= (uint64_t)((uint32_t)r.bits & 7u) << (3 * sym);
    rd_drop(r, 3);
  }

  uint16_t *ctab = L.rev;                          // 256 entries fit in rev[320]
  if (!build_clen_table<T>(ctab, packed, lig)) return INFL_D_INVALID;

  for (int s = lig; s < 320; s += T) L.lens[s] = 0;
  group_sync();

  uint32_t prev = 0;
  for (uint32_t i = 0; i < n;) {
    rd_fill(r);
    if (r.n < 7) return INFL_D_NEED_INPUT;
    uint32_t e = ctab[(uint32_t)r.bits & 255u];
    if (!(e & 255u)) return INFL_D_INVALID;
    uint32_t sym = e >> 8;
    rd_drop(r, e & 255u);
    if (sym < 16) {
      if (lig == 0) L.lens[i] = (uint8_t)sym;
      prev = sym;
      i++;
    } else if (sym == 16) {
      if (r.n < 2) return INFL_D_NEED_INPUT;
      if (i == 0) return INFL_D_INVALID;
      uint32_t rep = 3u + ((uint32_t)r.bits & 3u);
      rd_drop(r, 2);
      if (i + rep > n) return INFL_D_INVALID;
      if ((uint32_t)lig < rep) L.lens[i + lig] = (uint8_t)prev;   // rep <= 6 <= T
      i += rep;
    } else if (sym == 17) {
      if (r.n < 3) return INFL_D_NEED_INPUT;
      uint32_t rep = 3u + ((uint32_t)r.bits & 7u);
      rd_drop(r, 3);
      if (i + rep > n) return INFL_D_INVALID;
      i += rep;
      prev = 0;
    } else {
      if (r.n < 7) return INFL_D_NEED_INPUT;
      uint32_t rep = 11u + ((uint32_t)r.bits & 127u);
      rd_drop(r, 7);
      if (i + rep > n) return INFL_D_INVALID;
      i += rep;
      prev = 0;
    }
  }

  if (!build_table<T>(L.lit, L.lens, hlit, LIT_BITS, LIT_CAP, true, L, lig, gshift) ||
      !build_table<T>(L.dist, L.lens + hlit, hdist, DIST_BITS, DIST_CAP, false, L, lig, gshift))
    return INFL_D_INVALID;
  return INFL_D_OK;
}

// Stored block reached through the block loop (rules of infl.c:659-715).
template <int T>
__device__ __forceinline__ int stored_block(Reader &r, uint8_t *dst, uint32_t &pos, uint32_t cap, int lig) {
  rd_drop(r, r.n & 7u);
  rd_fill(r);
  if (r.n < 32) return INFL_D_NEED_INPUT;
  uint32_t len  = (uint32_t)r.bits & 0xffffu;
  uint32_t nlen = ((uint32_t)r.bits >> 16) & 0xffffu;
  rd_drop(r, 32);
  if (len != (nlen ^ 0xffffu)) return INFL_D_INVALID;
  if (len > cap - pos) return INFL_D_FULL;
  const uint8_t *s = rd_byte_ptr(r);
  if ((size_t)(r.end - s) < len) return INFL_D_NEED_INPUT;
  coop_copy<T>(dst + pos, s, len, lig);
  pos += len;
  rd_seek(r, s + len);
  return INFL_D_OK;
}

// One Huffman-coded block (rules of infl.c:717-901).
template <int T>
__device__ __forceinline__ int huffman_block(Reader &r, uint8_t *dst, uint32_t &pos_io, uint32_t cap,
                             const StreamLds &L, int lig, uint64_t &fuel) {
  uint32_t pos = pos_io;
  for (;;) {
    if (fuel-- == 0) return INFL_D_INTERNAL;
    rd_fill(r);
    uint32_t e = tab_lookup(L.lit, LIT_BITS, (uint32_t)r.bits);
    if (!e) return INFL_D_INVALID;
    uint32_t total = e & 31u;
    if (r.n < total) return INFL_D_NEED_INPUT;

    if (e & E_LITERAL) {
      rd_drop(r, total);
      if (pos >= cap) return INFL_D_FULL;
      if (lig == 0) dst[pos] = (uint8_t)(e >> 16);
      pos++;
      continue;
    }
    if (e & E_END) {
      rd_drop(r, total);
      break;
    }
    uint32_t len = (e >> 16) + (((uint32_t)r.bits & ((1u << total) - 1u)) >> ((e >> 5) & 15u));
    rd_drop(r, total);

    rd_fill(r);
    e = tab_lookup(L.dist, DIST_BITS, (uint32_t)r.bits);
    if (!e) return INFL_D_INVALID;
    total = e & 31u;
    if (r.n < total) return INFL_D_NEED_INPUT;
    uint32_t dist = (e >> 16) + (((uint32_t)r.bits & ((1u << total) - 1u)) >> ((e >> 5) & 15u));
    rd_drop(r, total);

    if (!dist || dist > pos) return INFL_D_INVALID;
    if (len > cap - pos) return INFL_D_FULL;

    // literal stores of lane 0 and earlier match stores must be ordered before
    // the loads below for the compiler too (the hardware keeps issue order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    coop_match<T>(dst, pos, dist, len, lig);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    pos += len;
  }
  pos_io = pos;
  return INFL_D_OK;
}

// Block loop (infl.c:1063-1092).
template <int T>
__device__ int block_loop(Reader &r, uint8_t *dst, uint32_t &pos, uint32_t cap, StreamLds &L,
                          int lig, int gshift, uint64_t &fuel) {
  bool     fixed_ready = false;
  uint32_t bfinal = 0;
  while (!bfinal) {
    if (fuel-- == 0) return INFL_D_INTERNAL;
    rd_fill(r);
    if (r.n < 3) return INFL_D_NEED_INPUT;
    bfinal = (uint32_t)r.bits & 1u;
    uint32_t btype = ((uint32_t)r.bits >> 1) & 3u;
    rd_drop(r, 3);
    int rc;
    if (btype == 0) {
      rc = stored_block<T>(r, dst, pos, cap, lig);
    } else if (btype == 1) {
      if (!fixed_ready) {
        if (!setup_fixed<T>(L, lig, gshift)) return INFL_D_INTERNAL;
        fixed_ready = true;
      }
      rc = huffman_block<T>(r, dst, pos, cap, L, lig, fuel);
    } else if (btype == 2) {
      fixed_ready = false;
      rc = setup_dynamic<T>(r, L, lig, gshift);
      if (rc == INFL_D_OK) rc = huffman_block<T>(r, dst, pos, cap, L, lig, fuel);
    } else {
      rc = INFL_D_INVALID;
    }
    if (rc != INFL_D_OK) return rc;
  }
  return INFL_D_OK;
}

__device__ __forceinline__ bool zlib_header_ok(const uint8_t *p) {   // infl.c:233-238
  uint32_t cmf = p[0], flg = p[1];
  return (cmf & 15u) == 8u && (cmf >> 4) <= 7u && ((cmf << 8) + flg) % 31u == 0 && !(flg & 0x20u);
}

// Whole stream: the dispatcher of infl() for one contiguous input.
// Returns the detail code.  `outlen` is what infl_output_pos() would report
// after infl(); `pos` is the output position actually reached and `cur` the
// first input byte with unread bits (both also valid when the stream stopped
// early; the incremental API reports them).
template <int T>
__device__ int inflate_stream(const uint8_t *src, uint32_t srclen, uint8_t *dst, uint32_t cap,
                              bool zlib, StreamLds &L, int lig, int gshift, uint32_t &outlen,
                              uint32_t &pos, const uint8_t *&cur) {
  outlen = 0;
  pos    = 0;
  cur    = src;
  if (srclen == 0) return INFL_D_OK;               // infl.c:1547-1550

  const uint8_t *p = src, *end = src + srclen;
  uint64_t fuel = (uint64_t)srclen * 8u + 64u;     // every loop turn consumes >= 1 input bit
  bool     try_stored = true;                      // infl.c:1519-1531
  if (!zlib) try_stored = ((p[0] >> 1) & 3u) == 0;
  else if (srclen >= 3) try_stored = ((p[2] >> 1) & 3u) == 0;

  Reader r;
  r.end = end;
  int rc;

  if (try_stored && !(zlib && srclen < 3)) {       // stored-only path, infl.c:206-267
    if (zlib) {
      if (!zlib_header_ok(p)) return INFL_D_INVALID;
      p += 2;
    }
    for (;;) {
      cur = p;
      if (p >= end) return INFL_D_NEED_INPUT;
      uint32_t b0 = p[0];
      if (((b0 >> 1) & 3u) != 0) break;
      if ((size_t)(end - p) < 5) return INFL_D_NEED_INPUT;
      uint32_t len  = p[1] | ((uint32_t)p[2] << 8);
      uint32_t nlen = p[3] | ((uint32_t)p[4] << 8);
      if (len != (nlen ^ 0xffffu)) return INFL_D_INVALID;
      if (len > cap - pos) return INFL_D_FULL;
      p += 5;
      if ((size_t)(end - p) < len) return INFL_D_NEED_INPUT;
      coop_copy<T>(dst + pos, p, len, lig);
      p += len;
      pos += len;
      if (b0 & 1u) {
        cur    = p;
        outlen = pos;
        return INFL_D_OK;
      }
    }
    // stored prefix, then a compressed block: the reference continues in its
    // older decoder with the output position already published (infl.c:256-259)
    outlen = pos;
  } else if (zlib) {                               // infl.c:1027-1039
    if (srclen < 2) return INFL_D_NEED_INPUT;
    if (!zlib_header_ok(p)) return INFL_D_INVALID;
    p += 2;
  }
  rd_seek(r, p);
  rc  = block_loop<T>(r, dst, pos, cap, L, lig, gshift, fuel);
  cur = rd_byte_ptr(r);
  if (rc == INFL_D_OK) outlen = pos;
  return rc;
}

// flags: bit 0 = zlib wrapper (INFL_ZLIB); bit 8 = publish the position reached
// in `outlen` even when the stream stopped early (incremental API).
template <int T>
__global__ __launch_bounds__(64) void infl_batch_kernel(const uint8_t *__restrict__ src,
                                                        const uint64_t *__restrict__ src_off,
                                                        uint8_t *__restrict__ dst,
                                                        const uint64_t *__restrict__ dst_off,
                                                        int flags, uint64_t n,
                                                        int32_t *__restrict__ status,
                                                        uint32_t *__restrict__ outlen,
                                                        int32_t *__restrict__ detail,
                                                        uint32_t *__restrict__ inpos) {
  constexpr int G = 64 / T;                        // streams per wavefront
  __shared__ StreamLds lds[G];

  const int      lane   = threadIdx.x;
  const int      group  = lane / T;
  const int      lig    = lane % T;
  const int      gshift = group * T;
  const uint64_t i      = (uint64_t)blockIdx.x * G + group;
  if (i >= n) return;

  const uint64_t s0 = src_off[i], s1 = src_off[i + 1];
  const uint64_t d0 = dst_off[i], d1 = dst_off[i + 1];
  uint32_t       out = 0, pos = 0;
  const uint8_t *cur = nullptr;
  int rc = inflate_stream<T>(src + s0, (uint32_t)(s1 - s0), dst + d0, (uint32_t)(d1 - d0),
                             (flags & 0xff) == 1, lds[group], lig, gshift, out, pos, cur);
  if (lig == 0) {
    status[i] = rc == INFL_D_OK ? 0 : -1;          // every failure is UNZ_ERR (infl.c:1074-1087)
    outlen[i] = ((flags & 0x100) && rc != INFL_D_OK && rc != INFL_D_INVALID) ? pos : out;
    if (detail) detail[i] = rc;
    if (inpos) inpos[i] = (uint32_t)(cur - (src + s0));
  }
}

}  // namespace defl

extern "C" int infl_launch_batch(const void *src, const uint64_t *src_off, void *dst,
                                 const uint64_t *dst_off, int flags, uint64_t n, int32_t *status,
                                 uint32_t *outlen, int32_t *detail, uint32_t *inpos, int lanes,
                                 void *stream) {
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  auto s = (const uint8_t *)src;
  auto d = (uint8_t *)dst;
#define LAUNCH(T)                                                                              \
  hipLaunchKernelGGL(defl::infl_batch_kernel<T>, dim3((unsigned)((n + (64 / T) - 1) / (64 / T))), \
                     dim3(64), 0, st, s, src_off, d, dst_off, flags, n, status, outlen, detail, inpos)
  switch (lanes) {
    case 64: LAUNCH(64); break;
    case 32: LAUNCH(32); break;
    case 16: LAUNCH(16); break;
    case 8:  LAUNCH(8);  break;
    default: return -1;
  }
#undef LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
