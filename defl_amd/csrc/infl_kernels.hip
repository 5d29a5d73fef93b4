// infl_kernels.hip -- DEFLATE / zlib block decode for gfx950 (MI355X, CDNA4).
//
// One group of T lanes of a wavefront (T = 64, 32, 16 or 8) decodes one
// independent stream with the semantics of recp/defl's one-shot infl() on a
// single contiguous input (reference src/infl/infl.c:1507-1551 dispatcher,
// :94-267 stored-only path, :1002-1104 block loop, :903-1000 dynamic header,
// :452-578 table build and its acceptance rules, :717-901 symbol loop).
// Nothing here is translated from that file: the reference is a scalar CPU loop
// over a 64-bit reservoir; this is a SIMT kernel whose lanes share the decode
// tables of their stream in LDS, replicate the bit reservoir in registers, lay
// the tables out analytically from the code-length histogram, build them
// cooperatively and copy LZ77 matches and stored blocks one byte (or 16 bytes)
// per lane.
//
// Two table formats exist:
//   RefFmt      32-bit entries in the reference's own geometry (10-bit literal
//               and 8-bit distance main tables, 2048 / 512 entries).  Decodes
//               every stream and produces the reference's exact failure result.
//   CompactFmt  16-bit entries, 9-bit / 7-bit main tables, 856 / 404 entries:
//               a quarter of the LDS per stream, so four times as many streams
//               are in flight per CU.  A stream whose tables do not fit, or
//               that fails in any way, is handed over ("deferred") to a second
//               launch that uses RefFmt, so results never depend on the format.
//
// All arithmetic is integer.  The file is compiled with packed-fp32 ops off
// like every HIP source of this library (see the Makefile).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <type_traits>

#include "infl_device.h"

namespace defl {

constexpr int REF_LIT_BITS  = 10;     // infl.c:378
constexpr int REF_DIST_BITS = 8;      // infl.c:379
constexpr int REF_LIT_CAP   = 2048;   // infl.c:382
constexpr int REF_DIST_CAP  = 512;    // infl.c:383

// A decoded table entry, format independent.
enum : uint32_t { K_INVALID = 0, K_LITERAL = 1, K_END = 2, K_BASE = 3 };
struct Sym {
  uint32_t kind;      // K_*
  uint32_t codelen;   // Huffman code length
  uint32_t xbits;     // extra bits that follow the code
  uint32_t base;      // literal byte, length base or distance base
};

// RFC 1951 length / distance bases computed instead of looked up
// (values of apicommon.h:36-49).  `i` = length symbol - 257.
__device__ __forceinline__ void length_base(uint32_t i, uint32_t &base, uint32_t &xb) {
  xb   = i < 8 || i == 28 ? 0u : (i - 4) >> 2;
  base = i < 8 ? 3 + i : i == 28 ? 258u : 3 + ((4 + (i & 3)) << xb);
}
__device__ __forceinline__ void dist_base(uint32_t d, uint32_t &base, uint32_t &xb) {
  xb   = d < 4 ? 0u : (d >> 1) - 1;
  base = d < 4 ? 1 + d : 1 + ((2 + (d & 1)) << xb);
}

// ---------------------------------------------------------------------------
// RefFmt: the reference's packed 32-bit entry (infl.c:385-397): bits 0-4
// codelen + xbits, 5-8 codelen, 9-12 xbits, 13 LITERAL, 14 END, 15 SUBTABLE,
// 16-31 base value or subtable offset.
// ---------------------------------------------------------------------------
struct RefFmt {
  typedef uint32_t entry_t;
  static constexpr int LIT_BITS = REF_LIT_BITS, DIST_BITS = REF_DIST_BITS;
  static constexpr int LIT_CAP = REF_LIT_CAP, DIST_CAP = REF_DIST_CAP;
  static constexpr uint32_t F_LITERAL = 1u << 13, F_END = 1u << 14, F_SUB = 1u << 15;

  static __device__ __forceinline__ uint32_t pack(uint32_t base, uint32_t xb, uint32_t len, uint32_t f) {
    return (base << 16) | f | (xb << 9) | (len << 5) | (len + xb);
  }
  static __device__ __forceinline__ entry_t lit_entry(uint32_t sym, uint32_t len) {
    if (sym < 256) return pack(sym, 0, len, F_LITERAL);
    if (sym == 256) return pack(0, 0, len, F_END);
    if (sym > 285) return 0;                     // 286, 287 never get an entry (infl.c:433-438)
    uint32_t base, xb;
    length_base(sym - 257, base, xb);
    return pack(base, xb, len, 0);
  }
  static __device__ __forceinline__ entry_t dist_entry(uint32_t sym, uint32_t len) {
    if (sym > 29) return 0;                      // 30, 31 never get an entry (infl.c:445-446)
    uint32_t base, xb;
    dist_base(sym, base, xb);
    return pack(base, xb, len, 0);
  }
  static __device__ __forceinline__ entry_t sub_entry(uint32_t at, uint32_t bits) {
    return (at << 16) | F_SUB | (bits << 5);
  }
  static __device__ __forceinline__ bool is_sub(uint32_t e) { return e & F_SUB; }
  static __device__ __forceinline__ uint32_t sub_at(uint32_t e) { return e >> 16; }
  static __device__ __forceinline__ uint32_t sub_bits(uint32_t e) { return (e >> 5) & 15u; }
  static __device__ __forceinline__ Sym lit_sym(uint32_t e) {
    Sym s;
    s.kind    = !e ? K_INVALID : (e & F_LITERAL) ? K_LITERAL : (e & F_END) ? K_END : K_BASE;
    s.codelen = (e >> 5) & 15u;
    s.xbits   = (e >> 9) & 15u;
    s.base    = e >> 16;
    return s;
  }
  static __device__ __forceinline__ Sym dist_sym(uint32_t e) { return lit_sym(e); }
};

// ---------------------------------------------------------------------------
// CompactFmt: 16-bit entries whose class shows in their magnitude, so that the
// symbol loops tell the classes apart with one unsigned comparison each:
//   >= 0xC000  SUBTABLE  bits 0-3 index bits, bits 4-12 offset / 2
//   >= 0x8000  LENGTH    bits 0-4 total = code length + extra bits, 5-8 code length,
//                        9-11 mantissa m, 12 "length 258"; base = 3 + (m << extra bits)
//   >= 0x4000  LITERAL   bits 0-4 code length, 5-12 the byte
//   >= 0x2000  DISTANCE  bits 0-4 total, 5-8 code length, 9-11 mantissa; base = 1 + (m << extra bits)
//   below      END       bits 0-4 code length; 0 = no code
// Bits 0-4 are what the code consumes whatever its class.  A base is stored as
// mantissa and shift, so decode needs no second table.
// ---------------------------------------------------------------------------
struct CompactFmt {
  typedef uint16_t entry_t;
  static constexpr int LIT_BITS = 9, DIST_BITS = 7;
  // Capacities that take every COMPLETE code: with a 9-bit main table 288 literal/length symbols of
  // up to 15 bits need at most 854 entries, with a 7-bit one 32 distance symbols at most 402 (the
  // search zlib's enough.c makes, which gives its 852 / 592 for 286 symbols at 9 bits and 30 at 6).
  // Smaller tables (768 / 320) sent 3 of 8 192 level-6 text streams, whose 15-bit distance codes
  // need 386 entries, to the serial passes, where one stream takes as long as a whole batch.
  static constexpr int LIT_CAP = 856, DIST_CAP = 404;
  static constexpr uint32_t C_SUB = 0xC000u, C_LENGTH = 0x8000u, C_LITERAL = 0x4000u, C_DIST = 0x2000u;
  static constexpr uint32_t F_L258 = 1u << 12;

  static __device__ __forceinline__ entry_t lit_entry(uint32_t sym, uint32_t len) {
    if (sym < 256) return (entry_t)(C_LITERAL | (sym << 5) | len);
    if (sym == 256) return (entry_t)len;
    if (sym > 285) return 0;
    uint32_t i = sym - 257;
    if (i == 28) return (entry_t)(C_LENGTH | F_L258 | (len << 5) | len);
    uint32_t x = i < 8 ? 0u : (i - 4) >> 2;
    uint32_t m = i < 8 ? i : 4 + (i & 3);
    return (entry_t)(C_LENGTH | (m << 9) | (len << 5) | (len + x));
  }
  static __device__ __forceinline__ entry_t dist_entry(uint32_t sym, uint32_t len) {
    if (sym > 29) return 0;
    uint32_t x = sym < 4 ? 0u : (sym >> 1) - 1;
    uint32_t m = sym < 4 ? sym : 2 + (sym & 1);
    return (entry_t)(C_DIST | (m << 9) | (len << 5) | (len + x));
  }
  // subtable offsets are even (main table and every subtable have even sizes): at / 2 fits 9 bits
  static __device__ __forceinline__ entry_t sub_entry(uint32_t at, uint32_t bits) {
    return (entry_t)(C_SUB | ((at >> 1) << 4) | bits);
  }
  static __device__ __forceinline__ bool is_sub(uint32_t e) { return e >= C_SUB; }
  static __device__ __forceinline__ uint32_t sub_at(uint32_t e) { return ((e >> 4) & 0x1ffu) << 1; }
  static __device__ __forceinline__ uint32_t sub_bits(uint32_t e) { return e & 15u; }
  static __device__ __forceinline__ uint32_t total(uint32_t e) { return e & 31u; }
  static __device__ __forceinline__ uint32_t codelen(uint32_t e) { return (e >> 5) & 15u; }       // LENGTH, DISTANCE
  static __device__ __forceinline__ uint32_t xbits(uint32_t e) { return total(e) - codelen(e); }  // LENGTH, DISTANCE
  static __device__ __forceinline__ uint32_t literal(uint32_t e) { return (e >> 5) & 0xffu; }
  static __device__ __forceinline__ uint32_t len_base(uint32_t e) {
    return (e & F_L258) ? 258u : 3u + (((e >> 9) & 7u) << xbits(e));
  }
  static __device__ __forceinline__ uint32_t dist_base(uint32_t e) {
    return 1u + (((e >> 9) & 7u) << xbits(e));
  }
  static __device__ __forceinline__ Sym lit_sym(uint32_t e) {
    Sym s;
    s.codelen = total(e);
    s.xbits   = 0;
    s.base    = literal(e);
    if (e >= C_LENGTH) {
      s.kind    = K_BASE;
      s.codelen = codelen(e);
      s.xbits   = xbits(e);
      s.base    = len_base(e);
    } else if (e >= C_LITERAL) {
      s.kind = K_LITERAL;
    } else {
      s.kind = e ? K_END : K_INVALID;
    }
    return s;
  }
  static __device__ __forceinline__ Sym dist_sym(uint32_t e) {
    Sym s;
    s.codelen = codelen(e);
    s.kind    = e ? K_BASE : K_INVALID;
    s.xbits   = xbits(e);
    s.base    = dist_base(e);
    return s;
  }
};

// Per-stream working set in LDS.  Two regions do double duty while a block
// is being set up, so that no separate scratch is needed:
//   * the literal table region first holds the 256-entry (16-bit) table of the
//     code-length alphabet while the header is read;
//   * the distance table region holds the block's code lengths (up to 320
//     bytes) until the literal table is built; the (at most 32) distance code
//     lengths then move to registers and the distance table takes the region.
template <class F>
struct StreamLds {
  typename F::entry_t lit[F::LIT_CAP];
  typename F::entry_t dist[F::DIST_CAP];
  uint32_t count[16];                      // symbols per code length
  uint32_t next[16];                       // next canonical code per length
  uint16_t cls_lo[16], cls_hi[16], cls_at[16];   // subtable classes per code length

  __device__ __forceinline__ uint8_t  *lens() { return (uint8_t *)dist; }
  __device__ __forceinline__ uint16_t *ctab() { return (uint16_t *)lit; }
};
static_assert(sizeof(CompactFmt::entry_t) * CompactFmt::LIT_CAP >= 512, "code-length table must fit");
static_assert(sizeof(CompactFmt::entry_t) * CompactFmt::DIST_CAP >= 320, "code lengths must fit");
static_assert(CompactFmt::LIT_CAP / 2 <= 512 && CompactFmt::DIST_CAP / 2 <= 512, "9-bit subtable offsets");

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
  const uint8_t *p;      // first input byte not yet moved into `bits`
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

// first input byte that still has unread bits
__device__ __forceinline__ const uint8_t *rd_byte_ptr(const Reader &r) { return r.p - (r.n >> 3); }
// bits consumed since `origin`
__device__ __forceinline__ uint64_t rd_bitpos(const Reader &r, const uint8_t *origin) {
  return (uint64_t)(r.p - origin) * 8u - r.n;
}

// Reader of the compact-table pass.  The group keeps a window of 4 * T input
// bytes in registers (lane j holds word j) plus the next window, prefetched one
// window ahead with one coalesced load per lane.  A refill takes its word from
// the window with a cross-lane read (ds_bpermute: no memory access, no vmcnt
// wait), so the only global loads inside the symbol loop are the match copies.
// Every word is loaded from inside the stream (at min(offset, ilen - 4)) and
// shifted into place when used, so the tail needs no byte loop and nothing
// past the stream's last byte is ever read.  Needs ilen >= 4; shorter streams
// are left to the other reader.
// Unlike the reference's reservoir this reader keeps handing out zero words
// after the last input byte and counts them in `n`; rd_overrun() tells whether
// any such bit has been consumed.  The compact pass checks it before a stored
// copy and once at the end of the stream, and gives the stream to the second
// pass if it is set: that pass rejects the stream exactly as the reference does.
template <int T>
struct FastReader {
  const uint8_t *src;
  uint32_t       ilen, ilen4;   // stream length, and length - 4
  uint32_t       ipos;          // first input byte not yet moved into `bits`
  uint64_t       bits;
  uint32_t       n;
  uint32_t       wbase;         // stream offset of the current window
  uint32_t       cur;           // this lane's word of the current window
  uint32_t       nxt, nxtsh;    // its word of the next window, and the shift still to apply
  int            lig, gshift;
};

template <int T>
__device__ __forceinline__ void rd_load_next(FastReader<T> &r) {
  uint32_t off = r.wbase + 4u * T + 4u * (uint32_t)r.lig;
  uint32_t q   = off < r.ilen4 ? off : r.ilen4;
  __builtin_memcpy(&r.nxt, r.src + q, 4);
  uint32_t d = off - q;                              // 0 inside the stream
  r.nxtsh = d >= 4 ? 32u : d * 8;
}

template <int T>
__device__ __forceinline__ void rd_promote(FastReader<T> &r) {
  r.cur = r.nxtsh >= 32 ? 0u : r.nxt >> r.nxtsh;     // zero padded past the end
  r.wbase += 4u * T;
  rd_load_next(r);
}

template <int T>
__device__ __forceinline__ void rd_seek(FastReader<T> &r, const uint8_t *p) {
  r.ipos  = (uint32_t)(p - r.src);
  r.bits  = 0;
  r.n     = 0;
  r.wbase = r.ipos - 4u * T;                         // wraps; rd_promote adds it back
  rd_load_next(r);
  rd_promote(r);
}

template <int T>
__device__ __forceinline__ void rd_fill(FastReader<T> &r) {
  if (r.n < 32) {
    uint32_t idx = (r.ipos - r.wbase) >> 2;          // word of the window, group-uniform
    if (idx >= (uint32_t)T) {
      rd_promote(r);
      idx -= T;
    }
    uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((r.gshift + idx) << 2), (int)r.cur);
    r.bits |= (uint64_t)w << r.n;                    // words past the end of the stream are zero
    r.n    += 32;
    r.ipos += 4;
  }
}

template <int T>
__device__ __forceinline__ void rd_drop(FastReader<T> &r, uint32_t k) {
  r.bits >>= k;
  r.n    -= k;
}

template <int T>
__device__ __forceinline__ const uint8_t *rd_byte_ptr(const FastReader<T> &r) {
  return r.src + (r.ipos - (r.n >> 3));
}
template <int T>
__device__ __forceinline__ const uint8_t *rd_end(const FastReader<T> &r) { return r.src + r.ilen; }
template <int T>
__device__ __forceinline__ uint64_t rd_bitpos(const FastReader<T> &r, const uint8_t *origin) {
  return (uint64_t)(r.src - origin) * 8u + (uint64_t)r.ipos * 8u - r.n;
}
// true once bits past the end of the stream have been consumed (see the reader's comment)
template <int T>
__device__ __forceinline__ bool rd_overrun(const FastReader<T> &r) {
  return (uint64_t)r.ipos * 8u - r.n > (uint64_t)r.ilen * 8u;
}
__device__ __forceinline__ bool rd_overrun(const Reader &) { return false; }
// Bits of the reservoir that come from the input (the fast reader also counts the zero bits it
// supplies past the end of the stream in n).  The reference tests its reservoir, which holds
// real bits only, in several places of the dynamic block header (infl.c:915-978), most visibly
// "at least 7 bits before every code-length symbol" whatever that symbol's own length is.
__device__ __forceinline__ uint32_t rd_real(const Reader &r) { return r.n; }
template <int T>
__device__ __forceinline__ uint32_t rd_real(const FastReader<T> &r) {
  const int64_t left = (int64_t)r.ilen * 8 - ((int64_t)r.ipos * 8 - (int64_t)r.n);
  return left <= 0 ? 0u : left < (int64_t)r.n ? (uint32_t)left : r.n;
}
__device__ __forceinline__ const uint8_t *rd_end(const Reader &r) { return r.end; }
// Input bits between the reader's position and the end of the input (what the reference's reservoir could
// still be filled with), capped at 2^30.  After rd_fill, "rd_real(r) < k" is "rd_left(r) < k" for k <= 32:
// a loop that counts what it drops needs the 64-bit arithmetic once, not at every test.
__device__ __forceinline__ int32_t rd_left(const Reader &r) {
  const uint64_t left = (uint64_t)r.n + 8u * (uint64_t)(r.end - r.p);
  return left < (1u << 30) ? (int32_t)left : (int32_t)(1u << 30);
}
template <int T>
__device__ __forceinline__ int32_t rd_left(const FastReader<T> &r) {
  const int64_t left = (int64_t)r.ilen * 8 - ((int64_t)r.ipos * 8 - (int64_t)r.n);
  return left <= 0 ? 0 : left < (int64_t)(1 << 30) ? (int32_t)left : (int32_t)(1 << 30);
}

template <class F>
__device__ __forceinline__ uint32_t tab_lookup(const typename F::entry_t *tab, int mainbits,
                                               uint32_t bits) {
  uint32_t e = tab[bits & ((1u << mainbits) - 1u)];
  if (F::is_sub(e)) e = tab[F::sub_at(e) + ((bits >> mainbits) & ((1u << F::sub_bits(e)) - 1u))];
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
// Table build for one alphabet.
//
// Acceptance is the reference's (infl.c:452-578): reject an over-subscribed
// length set, reject when the main table plus all subtables of the REFERENCE
// geometry (ref_bits / ref_cap) exceed its capacity, accept incomplete sets;
// symbols without an entry still take part in code assignment and subtable
// sizing.  Returns BUILD_OK, BUILD_REJECT (the reference rejects this set) or
// BUILD_NOFIT (valid, but larger than this format's own capacity).
//
// Layout.  Canonical codes of one length are consecutive, so the main-table
// prefixes that need a subtable, and the widest code under each of them, follow
// from the histogram alone: for length l > mainbits the codes cover the prefix
// range [first[l] >> (l - mainbits), (first[l] + count[l] - 1) >> (l - mainbits)];
// ranges of growing l are ordered and overlap in at most their end points, where
// the longer length wins.  Walking l downwards gives every length its "class":
// the prefixes whose widest code has that length, with subtables of 2^(l -
// mainbits) entries laid out consecutively.  No per-prefix scratch is needed.
// ---------------------------------------------------------------------------
enum : int { BUILD_OK = 0, BUILD_REJECT = 1, BUILD_NOFIT = 2 };

#ifdef DEFL_PAR_PROFILE
__device__ unsigned long long g_par_prof[32];   // diagnostic builds: cycle and event counters of the parallel pass
#endif
#ifdef DEFL_PAR_PROFILE_SETUP                    // split of the block set-up time (takes the slots of the scan histogram)
#define SETUP_MARK(k) { if (T == 64) { const uint64_t t_ = __builtin_readcyclecounter(); if (lig == 0) atomicAdd(&g_par_prof[16 + (k)], (unsigned long long)(t_ - setup_t)); setup_t = t_; } }
#define SETUP_DECL    uint64_t setup_t = __builtin_readcyclecounter()
// (the wave path keeps its five intervals in registers and adds them up at its end: an atomic in the middle would be
//  waited for at the next call boundary and land in the next interval)
#define WSETUP_MARK(k) { const uint64_t t_ = __builtin_readcyclecounter(); setup_acc[k] = t_ - setup_t; setup_t = t_; }
#define WSETUP_DECL    uint64_t setup_acc[5] = {0, 0, 0, 0, 0}
#define WSETUP_FLUSH   { if (lane == 0) for (int k_ = 0; k_ < 5; k_++) atomicAdd(&g_par_prof[16 + k_], (unsigned long long)setup_acc[k_]); }
#else
#define SETUP_MARK(k)
#define SETUP_DECL
#define WSETUP_MARK(k)
#define WSETUP_DECL
#define WSETUP_FLUSH
#endif


// lanes below this one that are set in m
__device__ __forceinline__ uint32_t lanes_below(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// build_table for a whole wavefront (T = 64), same tables, same verdicts.  Everything the lanes have to agree on
// comes from ballots, so it lives in scalar registers and is worked out by the scalar unit: the histogram
// (no LDS atomics: 64 lanes on 15 counters serialise), the layout, the next code of every length.  Codes much
// shorter than the main table's index (dozens of copies of one entry) are written by all lanes together, one
// symbol after the other; the rest by their own lanes (at most four copies each).
// Lengths in registers: lane l holds the length of symbol l + 64 j in lj (0 where there is no such symbol).
template <class F, bool LITLEN>
__device__ __noinline__ int build_table_wave(typename F::entry_t *table, uint32_t l0, uint32_t l1, uint32_t l2, uint32_t l3,
                                             uint32_t l4, int lane) {
  constexpr int      mainbits  = LITLEN ? F::LIT_BITS : F::DIST_BITS;
  constexpr uint32_t cap       = LITLEN ? F::LIT_CAP : F::DIST_CAP;
  constexpr int      ref_bits  = LITLEN ? REF_LIT_BITS : REF_DIST_BITS;
  constexpr uint32_t ref_cap   = LITLEN ? REF_LIT_CAP : REF_DIST_CAP;
  constexpr uint32_t main_size = 1u << mainbits;
  constexpr int      NB        = LITLEN ? 5 : 1;     // batches of 64 symbols (at most 288 / 32)

  const uint32_t len5[5] = {l0, l1, l2, l3, l4};
  uint32_t       len[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) len[j] = len5[j];
  uint32_t count[16], first[16];
#pragma unroll
  for (int l = 1; l <= 15; l++) {
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) c += (uint32_t)__popcll(__ballot(len[j] == (uint32_t)l));
    count[l] = c;
  }
  uint32_t used = main_size;
  uint32_t cls_lo[16], cls_hi[16], cls_at[16];
  {
    int      left = 1;
    uint32_t code = 0, prev = 0;
#pragma unroll
    for (int l = 1; l <= 15; l++) {
      left = (left << 1) - (int)count[l];
      if (left < 0) return BUILD_REJECT;
      code = (code + prev) << 1;                     // count[0] left out: canonical codes
      prev = count[l];
      first[l] = code;
    }
    uint32_t ref_used = 1u << ref_bits, ref_claim = 0x7fffffffu, claim = 0x7fffffffu;
#pragma unroll
    for (int l = 15; l >= 1; l--) {
      const uint32_t c = count[l];
      if (l > ref_bits && c) {
        uint32_t lo = first[l] >> (l - ref_bits), hi = (first[l] + c - 1) >> (l - ref_bits);
        if (hi >= ref_claim) hi = ref_claim - 1;
        if (hi + 1 > lo) ref_used += (hi + 1 - lo) << (l - ref_bits);
        ref_claim = lo < ref_claim ? lo : ref_claim;
      }
      cls_lo[l] = 1, cls_hi[l] = 0, cls_at[l] = 0;   // empty class
      if (l > mainbits && c) {
        const uint32_t clo = first[l] >> (l - mainbits);
        uint32_t       chi = (first[l] + c - 1) >> (l - mainbits);
        if (chi >= claim) chi = claim - 1;
        if (chi + 1 > clo) {
          cls_lo[l] = clo;
          cls_hi[l] = chi;
          cls_at[l] = used;
          used += (chi + 1 - clo) << (l - mainbits);
        }
        claim = clo < claim ? clo : claim;
      }
    }
    if (ref_used > ref_cap) return BUILD_REJECT;     // infl.c:524-525
    if (used > cap) return BUILD_NOFIT;
  }

  for (uint32_t i = lane; i < used; i += 64) table[i] = 0;
  group_sync();
  // main-table entries that point at subtables
#pragma unroll
  for (int l = mainbits + 1; l <= 15; l++) {
    if (cls_hi[l] >= cls_lo[l])
      for (uint32_t p = cls_lo[l] + (uint32_t)lane; p <= cls_hi[l]; p += 64)
        table[__brev(p) >> (32 - mainbits)] =
            F::sub_entry(cls_at[l] + ((p - cls_lo[l]) << (l - mainbits)), (uint32_t)(l - mainbits));
  }
  group_sync();

  // canonical codes in symbol order (first[] now counts on), entries written at once
#pragma unroll
  for (int j = 0; j < NB; j++) {
    uint32_t code = 0;
#pragma unroll
    for (int l = 1; l <= 15; l++) {
      const uint64_t m = __ballot(len[j] == (uint32_t)l);
      if (len[j] == (uint32_t)l) code = first[l] + lanes_below(m);
      first[l] += (uint32_t)__popcll(m);
    }
    const uint32_t s = (uint32_t)lane + 64u * j, ln = len[j];
    const typename F::entry_t e = !ln ? 0 : LITLEN ? F::lit_entry(s, ln) : F::dist_entry(s, ln);
    const uint32_t r = ln ? __brev(code) >> (32 - ln) : 0u;
    const bool     wide = e && ln + 2u < (uint32_t)mainbits;          // more than four copies
    if (e && !wide) {
      if (ln <= (uint32_t)mainbits) {
        for (uint32_t i = r; i < main_size; i += 1u << ln) table[i] = e;
      } else {
        const uint32_t me  = table[r & (main_size - 1u)];
        const uint32_t at  = F::sub_at(me);
        const uint32_t end = 1u << F::sub_bits(me);
        for (uint32_t i = r >> mainbits; i < end; i += 1u << (ln - mainbits)) table[at + i] = e;
      }
    }
    for (uint64_t todo = __ballot(wide); todo; todo &= todo - 1) {
      const uint32_t b  = (uint32_t)__builtin_ctzll(todo);
      const uint32_t eb = (uint32_t)__builtin_amdgcn_readlane((int)e, (int)b), rb = (uint32_t)__builtin_amdgcn_readlane((int)r, (int)b),
                     lb = (uint32_t)__builtin_amdgcn_readlane((int)ln, (int)b);
      for (uint32_t i = rb + ((uint32_t)lane << lb); i < main_size; i += 64u << lb) table[i] = (typename F::entry_t)eb;
    }
  }
  group_sync();
  return BUILD_OK;
}

//
// Code lengths come from LDS (`lens`) or, when `lens` is null, from `lens_reg`:
// byte j of that register is the length of symbol lig + j * T (at most 32
// symbols, the distance alphabet).
template <int T, class F>
__device__ __noinline__ int build_table(typename F::entry_t *table, const uint8_t *lens,
                                        uint32_t lens_reg, uint32_t nsyms, bool litlen,
                                        StreamLds<F> &L, int lig, int gshift) {
  if constexpr (T == 64) {
    group_sync();                                    // the lengths were written by other lanes
    uint32_t len[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const uint32_t s = (uint32_t)lig + 64u * j;
      len[j] = s >= nsyms ? 0u : lens ? lens[s] : j < 4 ? (lens_reg >> (8 * j)) & 0xffu : 0u;
    }
    return litlen ? build_table_wave<F, true>(table, len[0], len[1], len[2], len[3], len[4], lig)
                  : build_table_wave<F, false>(table, len[0], 0, 0, 0, 0, lig);
  }
  const int      mainbits  = litlen ? F::LIT_BITS : F::DIST_BITS;
  const uint32_t cap       = litlen ? F::LIT_CAP : F::DIST_CAP;
  const int      ref_bits  = litlen ? REF_LIT_BITS : REF_DIST_BITS;
  const uint32_t ref_cap   = litlen ? REF_LIT_CAP : REF_DIST_CAP;
  const uint32_t main_size = 1u << mainbits;
  const uint64_t gmask     = (T == 64) ? ~0ull : (((1ull << (T & 63)) - 1ull) << gshift);

  group_sync();
  for (int i = lig; i < 16; i += T) L.count[i] = 0;
  group_sync();
  for (uint32_t s = lig, j = 0; s < nsyms; s += T, j++)
    atomicAdd(&L.count[lens ? lens[s] : (lens_reg >> (8 * j)) & 0xffu], 1u);
  group_sync();

  // every lane derives the same layout from the histogram; lane 0 publishes it
  uint32_t used = main_size;
  {
    int      left = 1;
    uint32_t code = 0, prev = 0, first[16];
#pragma unroll
    for (int l = 1; l <= 15; l++) {
      uint32_t c = L.count[l];
      left = (left << 1) - (int)c;
      if (left < 0) return BUILD_REJECT;
      code = (code + prev) << 1;                   // count[0] left out: canonical codes
      prev = c;
      first[l] = code;
      if (lig == 0) L.next[l] = code;
    }
    uint32_t ref_used = 1u << ref_bits, ref_claim = 0x7fffffffu, claim = 0x7fffffffu;
#pragma unroll
    for (int l = 15; l >= 1; l--) {
      uint32_t c = L.count[l];
      if (l > ref_bits && c) {
        uint32_t lo = first[l] >> (l - ref_bits), hi = (first[l] + c - 1) >> (l - ref_bits);
        if (hi >= ref_claim) hi = ref_claim - 1;
        if (hi + 1 > lo) ref_used += (hi + 1 - lo) << (l - ref_bits);
        ref_claim = lo < ref_claim ? lo : ref_claim;
      }
      if (l > mainbits) {
        uint32_t lo = 1, hi = 0, at = 0;           // empty class
        if (c) {
          uint32_t clo = first[l] >> (l - mainbits);
          uint32_t chi = (first[l] + c - 1) >> (l - mainbits);
          if (chi >= claim) chi = claim - 1;
          if (chi + 1 > clo) {
            lo = clo;
            hi = chi;
            at = used;
            used += (hi + 1 - lo) << (l - mainbits);
          }
          claim = clo < claim ? clo : claim;
        }
        if (lig == 0) {
          L.cls_lo[l] = (uint16_t)lo;
          L.cls_hi[l] = (uint16_t)hi;
          L.cls_at[l] = (uint16_t)(at < 0xffffu ? at : 0xffffu);
        }
      }
    }
    if (ref_used > ref_cap) return BUILD_REJECT;   // infl.c:524-525
    if (used > cap) return BUILD_NOFIT;
  }

  for (uint32_t i = lig; i < used; i += T) table[i] = 0;
  group_sync();

  // main-table entries that point at subtables
  for (int l = mainbits + 1; l <= 15; l++) {
    uint32_t lo = L.cls_lo[l], hi = L.cls_hi[l], at = L.cls_at[l];
    if (hi >= lo)
      for (uint32_t p = lo + lig; p <= hi; p += T)
        table[__brev(p) >> (32 - mainbits)] =
            F::sub_entry(at + ((p - lo) << (l - mainbits)), (uint32_t)(l - mainbits));
  }
  group_sync();

  // canonical codes in symbol order, T symbols per round, entries written at once
  for (uint32_t base = 0, j = 0; base < nsyms; base += T, j++) {
    uint32_t s    = base + lig;
    uint32_t len  = s >= nsyms ? 0 : lens ? lens[s] : (lens_reg >> (8 * j)) & 0xffu;
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
      typename F::entry_t e = litlen ? F::lit_entry(s, len) : F::dist_entry(s, len);
      uint32_t r = __brev(code) >> (32 - len);
      if (e) {
        if (len <= (uint32_t)mainbits) {
          for (uint32_t i = r; i < main_size; i += 1u << len) table[i] = e;
        } else {
          uint32_t me  = table[r & (main_size - 1u)];
          uint32_t at  = F::sub_at(me);
          uint32_t end = 1u << F::sub_bits(me);
          for (uint32_t i = r >> mainbits; i < end; i += 1u << (len - mainbits)) table[at + i] = e;
        }
      }
    }
    group_sync();
  }
  return BUILD_OK;
}

// 256-entry direct table for the 19-symbol code-length alphabet: the job the
// reference gives to recp/huff's huff_init_fast_lsb (call site infl.c:938).
// Entry = sym << 8 | len, 0 = no code.  `packed` holds the 19 3-bit lengths.
// Rejects an over-subscribed set, accepts an incomplete one.
template <int T>
__device__ __noinline__ bool build_clen_table(uint16_t *tab, uint64_t packed, int lig) {
  if constexpr (T == 64) {                           // lane s holds symbol s; ballots instead of counting loops
    const uint32_t len = lig < 19 ? (uint32_t)(packed >> (3 * lig)) & 7u : 0u;
    int      left = 1;
    uint32_t code = 0, prev = 0, mine = 0;
#pragma unroll
    for (int l = 1; l < 8; l++) {
      const uint64_t m = __ballot(len == (uint32_t)l);
      const uint32_t c = (uint32_t)__popcll(m);
      left = (left << 1) - (int)c;
      if (left < 0) return false;
      code = (code + prev) << 1;
      prev = c;
      if (len == (uint32_t)l) mine = code + lanes_below(m);
    }
    group_sync();
    ((uint32_t *)tab)[lig] = 0;
    ((uint32_t *)tab)[lig + 64] = 0;
    group_sync();
    const uint32_t e = ((uint32_t)lig << 8) | len, r = len ? __brev(mine) >> (32 - len) : 0u;
    for (uint64_t todo = __ballot(len != 0); todo; todo &= todo - 1) {
      const uint32_t b  = (uint32_t)__builtin_ctzll(todo);
      const uint32_t eb = (uint32_t)__builtin_amdgcn_readlane((int)e, (int)b), rb = (uint32_t)__builtin_amdgcn_readlane((int)r, (int)b),
                     lb = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)b);
      for (uint32_t i = rb + ((uint32_t)lig << lb); i < 256; i += 64u << lb) tab[i] = (uint16_t)eb;
    }
    group_sync();
    return true;
  }
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

// The (at most 32) distance code lengths lens[first .. first + n) as one register
// per lane: byte j = length of distance symbol lig + j * T.
template <int T>
__device__ __forceinline__ uint32_t dist_lens_to_reg(const uint8_t *lens, uint32_t first, uint32_t n,
                                                     int lig) {
  uint32_t v = 0;
  for (uint32_t s = lig, j = 0; s < n && j < 4; s += T, j++) v |= (uint32_t)lens[first + s] << (8 * j);
  return v;
}

// Fixed-Huffman code lengths (apicommon.h:53-64), then both tables.
template <int T, class F>
__device__ __noinline__ int setup_fixed(StreamLds<F> &L, int lig, int gshift) {
  uint8_t *lens = L.lens();
  group_sync();
  for (int s = lig; s < 288; s += T) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
  int rc = build_table<T, F>(L.lit, lens, 0, 288, true, L, lig, gshift);
  if (rc == BUILD_OK) rc = build_table<T, F>(L.dist, nullptr, 0x05050505u, 32, false, L, lig, gshift);
  return rc;
}

// Dynamic block header (rules of infl.c:903-1000, including "at least 7 bits
// must be left before every code-length symbol", :947).  Returns INFL_D_* or
// INFL_D_NOFIT.
constexpr int INFL_D_NOFIT = 100;   // valid so far, but the tables exceed the compact format

// The dynamic block header for a whole wavefront on the register-window reader.  Everything in a header is the
// same for all lanes, so it is kept in scalar registers and worked out by the scalar unit (the vector pipe is
// what the decoder runs short of): the bit reservoir is refilled with v_readlane from the window, the 256-entry
// code-length table sits in two vector registers per lane (four 16-bit entries) and is read with v_readlane, the
// lengths are collected in LDS and go from there into five vector registers (lane l, register j = symbol
// l + 64 j) for the table builders.  Same rules and the same order of tests as setup_dynamic below.
struct ScalarBits {
  uint64_t bits;
  uint32_t n, idx;       // idx: next word of the reader's window
  int32_t  left;         // input bits not yet consumed (rd_left)
};
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ void sb_open(ScalarBits &s, const FastReader<64> &r) {
  s.bits = ((uint64_t)uniform((uint32_t)(r.bits >> 32)) << 32) | uniform((uint32_t)r.bits);
  s.n    = uniform(r.n);
  s.idx  = uniform((r.ipos - r.wbase) >> 2);
  s.left = (int32_t)uniform((uint32_t)rd_left(r));
}
__device__ __forceinline__ void sb_fill(ScalarBits &s, FastReader<64> &r) {
  if (s.n < 32) {
    if (s.idx >= 64u) {
      rd_promote(r);
      s.idx -= 64u;
    }
    const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)r.cur, (int)s.idx);
    s.bits |= (uint64_t)w << s.n;
    s.n += 32;
    s.idx++;
  }
}
__device__ __forceinline__ void sb_drop(ScalarBits &s, uint32_t k) {
  s.bits >>= k;
  s.n -= k;
  s.left -= (int32_t)k;
}
__device__ __forceinline__ void sb_close(const ScalarBits &s, FastReader<64> &r) {
  r.bits = s.bits;
  r.n    = s.n;
  r.ipos = r.wbase + 4u * s.idx;
}

// (The reader goes in and comes back by value: a reader whose address is handed to an out-of-line function lives in
// memory for the whole kernel, and the symbol-by-symbol loops that read it at every step ran at half their speed.)
struct ReaderAfterHeader {
  uint64_t bits;
  uint32_t n, ipos, wbase, cur, nxt, nxtsh;
  int      rc;
};
template <class F>
__device__ __forceinline__ int setup_dynamic_wave_body(FastReader<64> &r, StreamLds<F> &L, int lane);

template <class F>
__device__ __noinline__ ReaderAfterHeader setup_dynamic_wave_call(FastReader<64> r, StreamLds<F> *L, int lane) {
  ReaderAfterHeader o;
  o.rc    = setup_dynamic_wave_body<F>(r, *L, lane);
  o.bits  = r.bits;
  o.n     = r.n;
  o.ipos  = r.ipos;
  o.wbase = r.wbase;
  o.cur   = r.cur;
  o.nxt   = r.nxt;
  o.nxtsh = r.nxtsh;
  return o;
}
template <class F>
__device__ __forceinline__ int setup_dynamic_wave(FastReader<64> &r, StreamLds<F> &L, int lane) {
  const ReaderAfterHeader o = setup_dynamic_wave_call<F>(r, &L, lane);
  r.bits  = o.bits;
  r.n     = o.n;
  r.ipos  = o.ipos;
  r.wbase = o.wbase;
  r.cur   = o.cur;
  r.nxt   = o.nxt;
  r.nxtsh = o.nxtsh;
  return o.rc;
}

template <class F>
__device__ __forceinline__ int setup_dynamic_wave_body(FastReader<64> &r, StreamLds<F> &L, int lane) {
  SETUP_DECL;
  WSETUP_DECL;
  ScalarBits b;
  sb_open(b, r);
  sb_fill(b, r);
  if (b.left < 14) return INFL_D_NEED_INPUT;
  const uint32_t hlit  = ((uint32_t)b.bits & 31u) + 257u;
  const uint32_t hdist = (((uint32_t)b.bits >> 5) & 31u) + 1u;
  const uint32_t hclen = (((uint32_t)b.bits >> 10) & 15u) + 4u;
  const uint32_t n     = hlit + hdist;
  sb_drop(b, 14);
  const uint64_t order_tail = 0xF1E2D3C4B5A69780ull;   // see setup_dynamic
  uint64_t packed = 0;
  for (uint32_t i = 0; i < hclen; i++) {
    const uint32_t sym = i < 3 ? 16u + i : (uint32_t)(order_tail >> (4 * (i - 3))) & 15u;
    sb_fill(b, r);
    if (b.left < 3) return INFL_D_NEED_INPUT;
    packed |= (uint64_t)((uint32_t)b.bits & 7u) << (3 * sym);
    sb_drop(b, 3);
  }
  uint16_t *ctab = L.ctab();
  WSETUP_MARK(0);
  if (!build_clen_table<64>(ctab, packed, lane)) return INFL_D_INVALID;
  WSETUP_MARK(1);
  // entries lane, lane + 64, lane + 128, lane + 192
  const uint32_t t01 = (uint32_t)ctab[lane] | ((uint32_t)ctab[lane + 64] << 16);
  const uint32_t t23 = (uint32_t)ctab[lane + 128] | ((uint32_t)ctab[lane + 192] << 16);

  // The lengths are collected in LDS (320 bytes where the distance table will be built later, zeroed first:
  // a run of zeros is then only a step forward) and go to registers when the header is through.
  uint8_t *lens = L.lens();
  group_sync();
  ((uint32_t *)lens)[lane] = 0;
  if (lane < 16) ((uint32_t *)lens)[lane + 64] = 0;
  group_sync();
  uint32_t prev = 0;
  for (uint32_t i = 0; i < n;) {
    sb_fill(b, r);
    if (b.left < 7) return INFL_D_NEED_INPUT;
    const uint32_t x = (uint32_t)b.bits & 255u;
    const uint32_t w = (x & 128u) ? (uint32_t)__builtin_amdgcn_readlane((int)t23, (int)(x & 63u))
                                  : (uint32_t)__builtin_amdgcn_readlane((int)t01, (int)(x & 63u));
    const uint32_t e = (x & 64u) ? w >> 16 : w & 0xffffu;
    if (!(e & 255u)) return INFL_D_INVALID;
    const uint32_t sym = e >> 8;
    sb_drop(b, e & 255u);
    if (sym < 16) {
      lens[i] = (uint8_t)sym;                        // (every lane stores the same byte to the same place)
      prev = sym;
      i++;
    } else {
      // 16: the length before, 3..6 times (2 more bits); 17: zero 3..10 times (3 bits); 18: zero 11..138 times (7 bits)
      const uint32_t xb = sym == 16 ? 2u : sym == 17 ? 3u : 7u;
      if (b.left < (int32_t)xb) return INFL_D_NEED_INPUT;
      if (sym == 16 && i == 0) return INFL_D_INVALID;
      const uint32_t rep = (sym == 18 ? 11u : 3u) + ((uint32_t)b.bits & ((1u << xb) - 1u));
      sb_drop(b, xb);
      if (i + rep > n) return INFL_D_INVALID;
      if (sym == 16) {
        if (prev && (uint32_t)lane < rep) lens[i + (uint32_t)lane] = (uint8_t)prev;
      } else {
        prev = 0;
      }
      i += rep;
    }
  }
  group_sync();
  uint32_t len[5];
#pragma unroll
  for (int j = 0; j < 5; j++) len[j] = lens[lane + 64 * j];
  group_sync();
  sb_close(b, r);
  WSETUP_MARK(2);

  int rc = build_table_wave<F, true>(L.lit, len[0], len[1], len[2], len[3], (uint32_t)lane + 256u < hlit ? len[4] : 0u, lane);
  WSETUP_MARK(3);
  if (rc == BUILD_OK) {
    // the distance lengths are symbols hlit .. hlit + hdist - 1, all in the fifth register: lane d takes symbol hlit + d
    const uint32_t dl = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((hlit - 256u + (uint32_t)lane) & 63u) << 2), (int)len[4]);
    rc = build_table_wave<F, false>(L.dist, (uint32_t)lane < hdist ? dl : 0u, 0, 0, 0, 0, lane);
  }
  WSETUP_MARK(4);
  WSETUP_FLUSH;
  return rc == BUILD_OK ? INFL_D_OK : rc == BUILD_REJECT ? INFL_D_INVALID : INFL_D_NOFIT;
}

template <class R> struct IsWindowReader64 { static constexpr bool value = false; };
template <> struct IsWindowReader64<FastReader<64>> { static constexpr bool value = true; };

template <int T, class F, class R>
__device__ __forceinline__ int setup_dynamic(R &r, StreamLds<F> &L, int lig, int gshift) {
  if constexpr (T == 64 && IsWindowReader64<R>::value) return setup_dynamic_wave<F>(r, L, lig);
  SETUP_DECL;
  rd_fill(r);
  int32_t left = rd_left(r);                         // input bits not yet consumed: what the reference's tests are about
  if (left < 14) return INFL_D_NEED_INPUT;
  uint32_t hlit  = ((uint32_t)r.bits & 31u) + 257u;
  uint32_t hdist = (((uint32_t)r.bits >> 5) & 31u) + 1u;
  uint32_t hclen = (((uint32_t)r.bits >> 10) & 15u) + 4u;
  uint32_t n     = hlit + hdist;
  rd_drop(r, 14);
  left -= 14;

  // code-length alphabet order 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15
  // (apicommon.h:52): the 16 entries after the first three, one per nibble
  const uint64_t order_tail = 0xF1E2D3C4B5A69780ull;
  uint64_t packed = 0;
  for (uint32_t i = 0; i < hclen; i++) {
    uint32_t sym = i < 3 ? 16u + i : (uint32_t)(order_tail >> (4 * (i - 3))) & 15u;
    rd_fill(r);
    if (left < 3) return INFL_D_NEED_INPUT;
    packed |= (uint64_t)((uint32_t)r.bits & 7u) << (3 * sym);
    rd_drop(r, 3);
    left -= 3;
  }

  uint16_t *ctab = L.ctab();
  uint8_t  *lens = L.lens();
  SETUP_MARK(0);
  if (!build_clen_table<T>(ctab, packed, lig)) return INFL_D_INVALID;

  for (int s = lig; s < 80; s += T) ((uint32_t *)lens)[s] = 0;       // 320 lengths
  group_sync();
  SETUP_MARK(1);

  uint32_t prev = 0;
  for (uint32_t i = 0; i < n;) {
    rd_fill(r);
    if (left < 7) return INFL_D_NEED_INPUT;
    uint32_t e = ctab[(uint32_t)r.bits & 255u];
    if (!(e & 255u)) return INFL_D_INVALID;
    uint32_t sym = e >> 8;
    rd_drop(r, e & 255u);
    left -= (int32_t)(e & 255u);
    if (sym < 16) {
      if (lig == 0) lens[i] = (uint8_t)sym;
      prev = sym;
      i++;
    } else {
      // 16: the length before, 3..6 times (2 more bits); 17: zero 3..10 times (3 bits); 18: zero 11..138 times (7 bits)
      const uint32_t xb = sym == 16 ? 2u : sym == 17 ? 3u : 7u;
      if (left < (int32_t)xb) return INFL_D_NEED_INPUT;
      if (sym == 16 && i == 0) return INFL_D_INVALID;
      const uint32_t rep = (sym == 18 ? 11u : 3u) + ((uint32_t)r.bits & ((1u << xb) - 1u));
      rd_drop(r, xb);
      left -= (int32_t)xb;
      if (i + rep > n) return INFL_D_INVALID;
      if (sym == 16) {
        if ((uint32_t)lig < rep) lens[i + lig] = (uint8_t)prev;   // rep <= 6 <= T
      } else {
        prev = 0;
      }
      i += rep;
    }
  }

  SETUP_MARK(2);
  int rc = build_table<T, F>(L.lit, lens, 0, hlit, true, L, lig, gshift);
  SETUP_MARK(3);
  if (rc == BUILD_OK) {
    uint32_t dl = dist_lens_to_reg<T>(lens, hlit, hdist, lig);
    rc = build_table<T, F>(L.dist, nullptr, dl, hdist, false, L, lig, gshift);
  }
  SETUP_MARK(4);
  return rc == BUILD_OK ? INFL_D_OK : rc == BUILD_REJECT ? INFL_D_INVALID : INFL_D_NOFIT;
}

// Stored block reached through the block loop (rules of infl.c:659-715).
template <int T, class R>
__device__ __forceinline__ int stored_block(R &r, uint8_t *dst, uint32_t &pos, uint32_t cap,
                                            int lig) {
  rd_drop(r, r.n & 7u);
  rd_fill(r);
  if (r.n < 32) return INFL_D_NEED_INPUT;
  uint32_t len  = (uint32_t)r.bits & 0xffffu;
  uint32_t nlen = ((uint32_t)r.bits >> 16) & 0xffffu;
  rd_drop(r, 32);
  if (len != (nlen ^ 0xffffu)) return INFL_D_INVALID;
  if (len > cap - pos) return INFL_D_FULL;
  if (rd_overrun(r)) return INFL_D_NEED_INPUT;
  const uint8_t *s = rd_byte_ptr(r);
  if ((size_t)(rd_end(r) - s) < len) return INFL_D_NEED_INPUT;
  if (len == 0) return INFL_D_OK;                  // a flush point: the reader stands where it has to
  coop_copy<T>(dst + pos, s, len, lig);
  pos += len;
  rd_seek(r, s + len);
  return INFL_D_OK;
}

// One Huffman-coded block (rules of infl.c:717-901).
// With `symbit` the bit offset (from `origin`) of the symbol the loop stopped at is stored there:
// the first symbol that was not carried out, behind which a later call can go on (infl_stream).
template <int T, class F, class R>
__device__ __forceinline__ int huffman_block(R &r, uint8_t *dst, uint32_t &pos_io, uint32_t cap,
                                             const StreamLds<F> &L, int lig, uint64_t &fuel,
                                             uint64_t *symbit = nullptr, const uint8_t *origin = nullptr) {
  (void)fuel;   // every turn consumes >= 1 bit of a finite input; blocks are counted by the caller
  uint32_t pos = pos_io;
  int      rc  = INFL_D_OK;
  for (;;) {
    rd_fill(r);
    if (symbit) *symbit = rd_bitpos(r, origin);
    Sym s = F::lit_sym(tab_lookup<F>(L.lit, F::LIT_BITS, (uint32_t)r.bits));
    if (s.kind == K_INVALID) { rc = INFL_D_INVALID; break; }
    uint32_t total = s.codelen + s.xbits;
    if (r.n < total) { rc = INFL_D_NEED_INPUT; break; }

    if (s.kind == K_LITERAL) {
      rd_drop(r, total);
      if (pos >= cap) { rc = INFL_D_FULL; break; }
      if (lig == 0) dst[pos] = (uint8_t)s.base;
      pos++;
      continue;
    }
    if (s.kind == K_END) {
      rd_drop(r, total);
      break;
    }
    uint32_t len = s.base + (((uint32_t)r.bits & ((1u << total) - 1u)) >> s.codelen);
    rd_drop(r, total);

    rd_fill(r);
    Sym d = F::dist_sym(tab_lookup<F>(L.dist, F::DIST_BITS, (uint32_t)r.bits));
    if (d.kind == K_INVALID) { rc = INFL_D_INVALID; break; }
    total = d.codelen + d.xbits;
    if (r.n < total) { rc = INFL_D_NEED_INPUT; break; }
    uint32_t dist = d.base + (((uint32_t)r.bits & ((1u << total) - 1u)) >> d.codelen);
    rd_drop(r, total);

    if (!dist || dist > pos) { rc = INFL_D_INVALID; break; }
    if (len > cap - pos) { rc = INFL_D_FULL; break; }

    // literal stores of lane 0 and earlier match stores must be ordered before
    // the loads below for the compiler too (the hardware keeps issue order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    coop_match<T>(dst, pos, dist, len, lig);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    pos += len;
  }
  pos_io = pos;
  return rc;
}

// Symbol loop of the compact pass.  Same acceptance as huffman_block(); a stream
// that stops for any reason is decoded again by the second pass, so the reason
// is not reported.
//
// Input accounting is relaxed inside this loop: the reader hands out zero bits
// past the end of the stream and counts them as if they were there, and the
// caller checks once, when the stream is finished, that no bit past the end was
// consumed (a stream that needs such bits is rejected by the reference, so it
// goes to the second pass).  Output bounds are checked per symbol.
//
// BOUNDED: the loop also ends, at a symbol boundary, once the input position has reached
// `stop_bits` (the parallel block decoder has stretches of very long matches decoded here).
// Returns COMPACT_END (the end-of-block code has been consumed), COMPACT_STOP (BOUNDED only) or
// COMPACT_FAIL.
enum : int { COMPACT_FAIL = 0, COMPACT_END = 1, COMPACT_STOP = 2 };
template <int T, bool BOUNDED = false>
__device__ __forceinline__ int huffman_block_compact(FastReader<T> &r, uint8_t *dst, uint32_t &pos_io,
                                                     uint32_t cap, const uint16_t *lit,
                                                     const uint16_t *dist, int lig, uint32_t stop_bits = 0) {
  typedef CompactFmt F;
  constexpr uint32_t NONE = 0xffffffffu;
  uint32_t pos = pos_io;
  int      ok  = COMPACT_FAIL;
  // The first T bytes of a non-overlapping match are loaded when the match is
  // decoded and stored when the next match starts (or the block ends), so the
  // load latency overlaps with the decode of the symbols in between.  Issue order
  // keeps it exact: the pending store is issued before any later load.
  uint32_t pend_off = NONE, pend_val = 0;
  // Runs (distance 1) repeat the byte before them: while that byte is known (a literal, or the
  // run before) they are stored without the load, whose way through memory behind the stores
  // in front of it is what a match costs here.
  uint32_t last = 0;
  bool     last_known = false;
  for (;;) {
    if (BOUNDED && r.ipos * 8u - r.n >= stop_bits) {
      ok = COMPACT_STOP;
      break;
    }
    // a truncated stream whose zero padding happens to decode (a literal, a valid match) must not
    // run on until the output is full: stop once the reservoir lies wholly behind the input
    if (r.ipos > r.ilen + 8u) break;
    rd_fill(r);
    uint32_t b = (uint32_t)r.bits;
    uint32_t e = lit[b & ((1u << F::LIT_BITS) - 1u)];
    if (F::is_sub(e)) e = lit[F::sub_at(e) + ((b >> F::LIT_BITS) & ((1u << F::sub_bits(e)) - 1u))];
    if (e < F::C_LENGTH) {
      if (e >= F::C_LITERAL) {
        if (pos >= cap) break;
        rd_drop(r, F::total(e));
        if (lig == 0) dst[pos] = (uint8_t)F::literal(e);
        last       = F::literal(e);
        last_known = true;
        pos++;
        continue;
      }
      if (e) {                                       // end of block (0: no code)
        rd_drop(r, F::total(e));
        ok = COMPACT_END;
      }
      break;
    }
    uint32_t cl  = F::codelen(e);
    uint32_t xb  = F::xbits(e);
    uint32_t len = F::len_base(e) + ((b >> cl) & ((1u << xb) - 1u));
    rd_drop(r, F::total(e));

    rd_fill(r);
    b = (uint32_t)r.bits;
    e = dist[b & ((1u << F::DIST_BITS) - 1u)];
    if (F::is_sub(e)) e = dist[F::sub_at(e) + ((b >> F::DIST_BITS) & ((1u << F::sub_bits(e)) - 1u))];
    cl = F::codelen(e);
    xb = F::xbits(e);
    uint32_t d = F::dist_base(e) + ((b >> cl) & ((1u << xb) - 1u));
    rd_drop(r, F::total(e));
    if (!e || d > pos || len > cap - pos) break;   // d >= 1 for every real entry

    if (pend_off != NONE) dst[pend_off] = (uint8_t)pend_val;
    pend_off = NONE;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (d == 1) {                                  // a run of the byte before it
      if (!last_known) last = dst[pos - 1];
      last_known = true;
      for (uint32_t k = lig; k < len; k += T) dst[pos + k] = (uint8_t)last;
    } else if (d >= len) {
      const uint8_t *sp = dst + (pos - d);
      if ((uint32_t)lig < len) {
        pend_val = sp[lig];
        pend_off = pos + lig;
      }
      for (uint32_t k = lig + T; k < len; k += T) dst[pos + k] = sp[k];
      last_known = false;
    } else {
      coop_match<T>(dst, pos, d, len, lig);
      last_known = false;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    pos += len;
  }
  if (pend_off != NONE) dst[pend_off] = (uint8_t)pend_val;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  pos_io = pos;
  return ok;
}

// ---------------------------------------------------------------------------
// Parallel symbol decode of one Huffman block by a whole wavefront (T = 64).
//
// The serial loops above advance one symbol per stream per pass over the
// instruction stream, so a stream never goes faster than one dependency chain
// allows.  Here the 64 lanes decode 64 consecutive sub-sequences of PAR_S bits
// of the SAME block at once.  Per round:
//   1. stage     the round's input goes to LDS, one column per sub-sequence;
//   2. scan      every lane decodes from its (guessed) start to the end of its
//                sub-sequence and notes where it left it (positions only);
//   3. fix up    every lane scans once more from where its predecessor left,
//                now counting bytes and taking three checkpoints; a lane whose
//                predecessor then leaves at yet another position rescans;
//                repeated until nothing changes.  DEFLATE chains re-synchronise
//                after ~128 bits on text, so a round takes 2.4 such scans.  With
//                codes that re-synchronise badly (near fixed-length codes) a round
//                is cut short after PAR_MAX_FIXUP scans, in front of the first
//                lane that still starts wrong;
//   4. place     a prefix sum of the byte counts gives every chain its output
//                offset; the chains are then handled in groups of up to 16 whose
//                bytes fit the PAR_R-entry link buffer;
//   5. emit      four lanes share a chain (one stretch between checkpoints each)
//                and write one 16-bit "head" per symbol at the symbol's first
//                byte: a literal, or a match as flag | distance - 1;
//   6. expand    every byte finds the head at or before it and becomes a literal,
//                the index of an earlier byte of the group, or "n bytes before
//                the group" (already final in global memory);
//   7. resolve   pointer jumping until every link ends at a literal or before
//                the group, then the bytes are written out (coalesced), those
//                from before the group fetched on the way.
// The next round starts where the last lane of the true chain stopped.  Where a
// sub-sequence holds more bytes than the link buffer takes (very long matches)
// a stretch of the block is decoded symbol by symbol instead (serial_stretch).
// Malformed data on the true chain or output capacity reached returns false
// and the stream is decoded again by the serial passes (they own the exact
// failure result), so this function only has to be right when it succeeds.
// ---------------------------------------------------------------------------
#ifndef DEFL_PAR_S
#define DEFL_PAR_S 256
#endif
#ifndef DEFL_PAR_R
#define DEFL_PAR_R 2048
#endif
constexpr int      PAR_S        = DEFL_PAR_S;        // sub-sequence length in bits
constexpr int      PAR_R        = DEFL_PAR_R;        // output bytes per group of chains (LDS links)
constexpr int      PAR_IN       = 2 * PAR_S + 4;     // staged input words: 64 * PAR_S bits + 4 for the last column
constexpr uint32_t PAR_LANE_CAP = 1024;              // a lane over this many bytes at a checkpoint ends the round
#ifndef DEFL_PAR_MAX_FIXUP
#define DEFL_PAR_MAX_FIXUP 16
#endif
constexpr int      PAR_MAX_FIXUP = DEFL_PAR_MAX_FIXUP;   // counting scans per round before the round is cut short
#ifndef DEFL_PAR_STRETCH
#define DEFL_PAR_STRETCH 2048
#endif
#ifndef DEFL_PAR_STRETCH_LANES
#define DEFL_PAR_STRETCH_LANES 8                     // a byte cap within this many lanes of a round: serial stretch
#endif
constexpr uint32_t PAR_STRETCH  = DEFL_PAR_STRETCH;              // input bits of a serial stretch (doubled while they follow each other)
#ifndef DEFL_PAR_MIN_INPUT
#define DEFL_PAR_MIN_INPUT 256
#endif
constexpr uint32_t PAR_MIN_INPUT = DEFL_PAR_MIN_INPUT;   // input bytes below which a stream is left to the serial pass
#ifndef DEFL_PAR_SHORT_BATCH
#define DEFL_PAR_SHORT_BATCH 6144
#endif
constexpr uint64_t PAR_SHORT_BATCH = DEFL_PAR_SHORT_BATCH;   // ... in batches of at least this many streams
constexpr int      PAR_U        = 4;                 // independent LDS/global accesses kept in flight
constexpr uint32_t LK_LIT = 0x4000u, LK_EXT = 0x8000u;

// Staged input words, one column per sub-sequence: row k of column l holds word l * PAR_G + k
// of the round, for k up to PAR_G + 3, i.e. a column repeats the first four words of the next
// one.  A lane walking sub-sequence l (it may run up to 48 bits over its end) finds the two
// words around any of its bit positions at a fixed distance from each other, and lanes at
// the same depth read neighbouring banks.
constexpr int PAR_G    = PAR_S / 32;
constexpr int PAR_ROWS = PAR_G + 4;
static_assert((PAR_G & (PAR_G - 1)) == 0 && PAR_G >= 4, "sub-sequence length is a power of two >= 128");

// Issue priority of the wavefront by phase (experiments; 0 and 0: no s_setprio at all)
#ifndef DEFL_PAR_PRIO_STEPS
#define DEFL_PAR_PRIO_STEPS 0
#endif
#ifndef DEFL_PAR_PRIO_BYTES
#define DEFL_PAR_PRIO_BYTES 0
#endif
#if DEFL_PAR_PRIO_STEPS != DEFL_PAR_PRIO_BYTES
#define PAR_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define PAR_PRIO(p)
#endif
#ifndef DEFL_PAR_FETCH_BATCHES
#define DEFL_PAR_FETCH_BATCHES 4                     // of 256 output bytes: 16 byte fetches per lane in flight in the write-out
#endif
#ifndef DEFL_PAR_PAD
#define DEFL_PAR_PAD 0                               // experiments: extra LDS per wave (lowers occupancy)
#endif
struct alignas(16) ParLds {
  uint32_t in[PAR_ROWS * 64];
  uint16_t link[PAR_R + DEFL_PAR_PAD / 2];
};

// Optional cycle accounting of the phases (diagnostic builds only: make prof).
#ifdef DEFL_PAR_PROFILE
#define PROF_DECL      uint64_t prof_t = __builtin_readcyclecounter(), prof_acc[10] = {}, prof_cnt[4] = {}, prof_hist[12] = {}
#define PROF_MARK(k)   { uint64_t t_ = __builtin_readcyclecounter(); prof_acc[k] += t_ - prof_t; prof_t = t_; }
#define PROF_COUNT(k, v) prof_cnt[k] += (v)
#ifdef DEFL_PAR_PROFILE_SETUP
#define PROF_HIST(v)
#else
#define PROF_HIST(v)   { _Pragma("unroll") for (int k_ = 0; k_ < 12; k_++) prof_hist[k_] += ((v) == k_ || (k_ == 11 && (v) > 11)); }
#endif
#define PROF_BAIL(k)   { if (lane == 0) atomicAdd(&g_par_prof[k], 1ull); }
#define PROF_FLUSH     if (lane == 0) { for (int k_ = 0; k_ < 10; k_++) atomicAdd(&g_par_prof[k_], prof_acc[k_]); \
                                        for (int k_ = 0; k_ < 4; k_++) atomicAdd(&g_par_prof[10 + k_], prof_cnt[k_]); \
                                        for (int k_ = 0; k_ < 12; k_++) atomicAdd(&g_par_prof[16 + k_], prof_hist[k_]); }
#else
#define PROF_DECL
#define PROF_MARK(k)
#define PROF_COUNT(k, v)
#define PROF_HIST(v)
#define PROF_BAIL(k)
#define PROF_FLUSH
#endif

struct ParCtx {
  const uint32_t *in;            // staged words (ParLds::in)
  uint32_t        column;        // the lane's column there (row stride 64), as a byte offset: 4 * column
  uint32_t        colbit;        // bit position of the column's first word
  uint32_t        total_bits;    // stream length in bits
  const uint16_t *lit, *dist;
};

__device__ __forceinline__ uint64_t wave_uniform(uint64_t v) {          // v is the same in every lane
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// inclusive maximum over the lanes at or below each lane (data-parallel-primitive row shifts and
// row broadcasts: six vector instructions, no LDS)
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v) {
#define DEFL_DPP_MAX(ctrl, rows)                                                                   \
  {                                                                                                 \
    const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, false);    \
    v = v > t_ ? v : t_;                                                                            \
  }
  DEFL_DPP_MAX(0x111, 0xf)                           // row_shr:1, 2, 4, 8
  DEFL_DPP_MAX(0x112, 0xf)
  DEFL_DPP_MAX(0x114, 0xf)
  DEFL_DPP_MAX(0x118, 0xf)
  DEFL_DPP_MAX(0x142, 0xa)                           // row_bcast:15 into rows 1 and 3
  DEFL_DPP_MAX(0x143, 0xc)                           // row_bcast:31 into rows 2 and 3
#undef DEFL_DPP_MAX
  return v;
}

__device__ __forceinline__ uint32_t wave_read(uint32_t v, uint32_t srclane) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(srclane << 2), (int)v);
}

// The value as it is, with its range hidden from the compiler: an OR of bit fields that it can prove disjoint
// becomes an addition, which then misses the and-or instruction.
__device__ __forceinline__ uint32_t opaque(uint32_t v) {
#if defined(__clang__)
  asm volatile("" : "+v"(v));
#endif
  return v;
}

// Two 16-bit values to a register, worked on with the packed integer instructions (v_pk_sub_u16, v_pk_ashrrev_i16);
// a portable form serves compilers without vector extensions (the host-side emulation of the test suite).
#if defined(__clang__)
typedef unsigned short pk_u16x2 __attribute__((ext_vector_type(2)));
typedef short          pk_i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, (pk_u16x2)(__builtin_bit_cast(pk_u16x2, a) - __builtin_bit_cast(pk_u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sign16(uint32_t a) {           // 0xffff where a half has its top bit set, else 0
  return __builtin_bit_cast(uint32_t, (pk_i16x2)(__builtin_bit_cast(pk_i16x2, a) >> (pk_i16x2)(15)));
}
#else
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b) {
  return ((a - b) & 0xffffu) | (((a >> 16) - (b >> 16)) << 16);
}
__device__ __forceinline__ uint32_t pk_sign16(uint32_t a) {
  return ((a & 0x8000u) ? 0xffffu : 0u) | ((a & 0x80000000u) ? 0xffff0000u : 0u);
}
#endif

// 32 bits of the stream from bit `rel` of the column on (see ParLds)
__device__ __forceinline__ uint32_t par_peek(const ParCtx &c, uint32_t rel) {
  static_assert(PAR_ROWS <= 16, "the row index is masked to four bits");
  const uint32_t  at = ((rel << 3) & 0xf00u) | c.column;   // byte offset of row : column (shift, and-or)
  const uint32_t *w  = (const uint32_t *)((const char *)c.in + at);
  return __builtin_amdgcn_alignbit(w[64], w[0], rel);
}

// Per-lane decoder state.  One step decodes ONE code: a literal/length code or, after a length
// code, the distance code.  The step is straight-line code (selects, no divergent branches but
// the rare second-level table): all 64 lanes run it together whatever kind of code each of them
// meets.  Which table the next code is looked up in travels as the table's address and index
// mask (msk = PL_DMASK: a distance code is due), chosen when the code before it is classified,
// so a step begins with the look-up itself.  A lane goes on while p < lim: lim is the end of its
// stretch, everything while it is inside a match, nothing once it has stopped.
constexpr uint32_t PL_LMASK = (1u << CompactFmt::LIT_BITS) - 1u, PL_DMASK = (1u << CompactFmt::DIST_BITS) - 1u;
struct ParLane {
  uint32_t        p, o;          // bit position (relative to the lane's column), bytes produced
  uint32_t        msk;           // index mask of the table the next code is looked up in
  const uint16_t *tab;           // that table
  uint32_t        lim, limit, last;   // see above; the entry decoded last (0 = there was no such code)
  uint32_t        st;            // 3 = stopped by the byte cap
};

// p and limit are positions in the stream; the lane keeps them relative to its column
__device__ __forceinline__ void pl_init(ParLane &b, const ParCtx &c, uint32_t p, uint32_t o, uint32_t limit,
                                        bool enable) {
  b.p     = p - c.colbit;
  b.msk   = PL_LMASK;
  b.tab   = c.lit;
  b.o     = o;
  b.st    = 0;
  b.last  = 1;
  b.limit = limit - c.colbit;
  b.lim   = enable ? b.limit : 0u;
}
__device__ __forceinline__ uint32_t pl_pos(const ParLane &b, const ParCtx &c) { return b.p + c.colbit; }
__device__ __forceinline__ bool pl_in_match(const ParLane &b) { return b.msk != PL_LMASK; }

// Decodes the code at the lane's position.  MODE 0: positions only (the first scan looks for
// where a chain ends); MODE 1: also counts the bytes produced; MODE 2: also returns `value`:
// the literal byte, or distance - 1 after a distance code.  What kind of code it was shows in
// the lane's state afterwards: pl_in_match() after a length code, lim = 0 after anything that
// ends the walk, lim = limit after a literal or a distance.  Whether a code runs past the end of
// the stream is checked by the caller once per scan (the staged input is zero padded).
template <int MODE>
__device__ __forceinline__ void pl_step(ParLane &r, const ParCtx &c, uint32_t &value) {
  typedef CompactFmt F;
  const uint32_t b   = par_peek(c, r.p);
  const bool     isd = r.msk != PL_LMASK;            // (MODE 2 only)
  uint32_t       e   = r.tab[b & r.msk];
#ifdef DEFL_PAR_PROFILE_STEPS                       // (atomics per step: distorts the cycle counts)
  if (threadIdx.x == (unsigned)__builtin_ctzll(__ballot(true))) {
    atomicAdd(&g_par_prof[30], 1ull);
    if (__ballot(e >= F::C_SUB)) atomicAdd(&g_par_prof[31], 1ull);
  }
#endif
  if (e >= F::C_SUB) {                               // second-level table (skipped when no lane needs it)
    const uint32_t mbits = r.msk == PL_LMASK ? (uint32_t)F::LIT_BITS : (uint32_t)F::DIST_BITS;
    e = r.tab[F::sub_at(e) + ((b >> mbits) & ((1u << F::sub_bits(e)) - 1u))];
  }
  // Entry classes by magnitude (see CompactFmt): each condition is one comparison that feeds
  // selects directly; conditions are never combined.
  const bool goon    = e >= F::C_DIST;               // a literal, a length or a distance
  const bool tomatch = e >= F::C_LENGTH;             // a length code: the distance code follows
  r.p += e & 31u;
  if (MODE >= 1) {
    const uint32_t cl    = (e >> 5) & 15u;
    const uint32_t extra = __builtin_amdgcn_ubfe(b, cl, (e & 31u) - cl);   // (a literal: not used)
    const uint32_t mant  = (((e >> 9) & 7u) << ((e & 31u) - cl)) + extra;
    const uint32_t len   = 3u + mant + (uint32_t)__umul24(255u, opaque((e >> 12) & 1u));   // ("length 258": mantissa and extra bits are 0;
                                                                           //  a multiply-add, not a compare and a select)
    r.o += tomatch ? len : (e >> 14) & 1u;                   // a distance entry adds nothing
    if (MODE == 2) value = isd ? mant : (e >> 5) & 0xffu;    // unused after a length code
  }
  r.last = e;
  r.tab  = tomatch ? c.dist : c.lit;
  r.msk  = tomatch ? PL_DMASK : PL_LMASK;
  r.lim  = tomatch ? ~0u : r.limit;
  r.lim  = goon ? r.lim : 0u;
}

// Result of scanning one sub-sequence.  ck[k-1] (k = 1..3) packs the chain's state at the
// symbol boundary it had reached after k quarters of the expected number of steps, so that
// four lanes can later re-walk one chain: (bit position - round start) << 12 | bytes so far.
struct ParScan {
  uint32_t exit, nout, stop;     // stop: 0 = left the sub-sequence, 1 = end of block, 2 = malformed, 3 = byte cap
  uint32_t ck[3];
};
static_assert(PAR_R < (1u << 12) && 64 * PAR_S + 64 < (1 << 20), "checkpoint packing");

// Scans the sub-sequences of the lanes with `enable`.  All lanes must call: the loops are kept
// wave-uniform (`quarter` steps, then a checkpoint for every lane, four times over; the last
// stretch runs until every lane is through).  Returns the number of steps taken.
template <bool FULL>
__device__ __forceinline__ uint32_t par_scan(const ParCtx &c, bool enable, uint32_t round0,
                                             uint32_t start, uint32_t limit, uint32_t quarter,
                                             ParScan &s) {
  ParLane  r;
  uint32_t it = 0, ck[3] = {0u, 0u, 0u};
  int      nq = 0;
  pl_init(r, c, start, 0, limit, enable);
  if constexpr (!FULL) {                             // positions only: no checkpoints, one plain loop (tested at its end:
    do {                                             // the compiler then keeps no copies of the lane's state around it)
      if (r.p < r.lim) {
        uint32_t value;
        pl_step<0>(r, c, value);
      }
      it++;
    } while (__ballot(r.p < r.lim));
  }
  if constexpr (FULL) {
    // one loop; every `quarter` steps (three times) all lanes note a checkpoint
    uint32_t left = quarter;
    for (;;) {
      const bool more = r.p < r.lim;
      if (!__ballot(more)) break;
      if (more) {
        uint32_t value;
        pl_step<1>(r, c, value);
      }
      it++;
      if (nq < 3 && --left == 0) {
        if (__ballot(pl_in_match(r))) {              // lanes inside a match finish it first
          if (pl_in_match(r)) {
            uint32_t value;
            pl_step<1>(r, c, value);
          }
        }
        const uint32_t state = ((pl_pos(r, c) - round0) << 12) | r.o;
        ck[0] = nq == 0 ? state : ck[0];
        ck[1] = nq == 1 ? state : ck[1];
        ck[2] = nq == 2 ? state : ck[2];
        nq++;
        left = quarter;
        if (!pl_in_match(r) && r.o >= PAR_LANE_CAP && r.p < r.lim) {   // byte cap: the round ends with this lane
          r.st  = 3;
          r.lim = 0;
        }
      }
    }
  }
  // why the lane stopped: byte cap (set above), no such code, end of block (the two ways a
  // step ends a walk), or a code that needs bits past the end of the stream
  if (!r.st && r.lim == 0) r.st = r.last == 0 ? 2u : 1u;
  if (pl_pos(r, c) > c.total_bits) r.st = 2;
  if (enable) {
    const uint32_t fin = ((pl_pos(r, c) - round0) << 12) | r.o;   // stretches never begun are empty
    s.ck[0] = nq > 0 ? ck[0] : fin;
    s.ck[1] = nq > 1 ? ck[1] : fin;
    s.ck[2] = nq > 2 ? ck[2] : fin;
    s.exit  = pl_pos(r, c);
    s.nout  = r.o;
    s.stop  = r.st;
  }
  return it;
}

// Returns INFL_D_OK; INFL_D_INVALID when the block certainly fails in the reference too (no such
// code or a code past the end of the input on the true chain, a distance before the output
// start, more output than capacity: the reason itself is not worked out here); INFL_D_NOFIT
// when the stream has to be decoded by the serial passes.
//
// NOTE (the pass that looks at failing streams a second time): when the block fails, lane 0 leaves in note[1] /
// note[2] a symbol boundary of the true chain in front of the failure and the output position there, so that the
// pass that works out the exact result goes on from it instead of from the start.  The first pass is compiled
// without: its loops sit at the register limit and any addition moves the compiler's spills into them.
template <bool NOTE>
__device__ __forceinline__ int huffman_block_parallel(FastReader<64> &r, uint8_t *dst, uint32_t &pos_io,
                                                    uint32_t cap, const uint16_t *lit,
                                                    const uint16_t *dist, ParLds &P, int lane,
                                                    bool expect_short, uint32_t *note) {
  uint32_t       pos        = pos_io;
  uint32_t       B          = r.ipos * 8u - r.n;     // bit position of the block's first symbol
  const uint32_t total_bits = r.ilen * 8u;
  uint32_t       guard      = total_bits + 8u;       // every round consumes at least one bit
  bool           done       = false;
  uint32_t       quarter    = 10;                    // steps per quarter of a chain, adapted per round
  uint32_t       boost      = 0;                     // serial stretches in a row (each twice as long as the one before)
  PROF_DECL;

  // Very long matches (runs, repeated records) give a sub-sequence more bytes than the link
  // buffer takes.  Such a stretch of the block is decoded one symbol after the other by the whole
  // wavefront (64 lanes per match copy: what suits that data), from bit position `from` up to a
  // symbol boundary some PAR_STRETCH bits on; the rounds go on behind it.  Sets B and `done`.
  auto serial_stretch = [&](uint32_t from) -> bool {        // false: the stream fails
    PROF_BAIL(28);
    if (NOTE && lane == 0) {
      note[1] = from;
      note[2] = pos;
    }
    group_sync();
    rd_seek(r, r.src + (from >> 3));
    rd_fill(r);
    rd_drop(r, from & 7u);
    const uint32_t span = PAR_STRETCH << boost;
    const int      res  = huffman_block_compact<64, true>(r, dst, pos, cap, lit, dist, lane,
                                                          span < total_bits - from ? from + span : total_bits);
    if (res == COMPACT_FAIL || rd_overrun(r)) return false;
    boost = boost < 6u ? boost + 1u : boost;
    B     = r.ipos * 8u - r.n;
    done  = res == COMPACT_END;
    group_sync();
    return true;
  };

  // A stream of many short blocks (flush points every few bytes) would pay a whole round per
  // block: after a short block the next one is tried symbol by symbol first, with the reader
  // as it stands (at B); a block that ends inside the stretch leaves it where it has to be.
  if (expect_short) {
    PROF_BAIL(28);
    const int res = huffman_block_compact<64, true>(r, dst, pos, cap, lit, dist, lane,
                                                    PAR_STRETCH < total_bits - B ? B + PAR_STRETCH : total_bits);
    if (res == COMPACT_FAIL || rd_overrun(r)) return INFL_D_INVALID;
    if (res == COMPACT_END) {
      pos_io = pos;
      return INFL_D_OK;
    }
    boost = 1;
    B     = r.ipos * 8u - r.n;
    group_sync();
  }

  while (!done) {
    if (guard-- == 0) return INFL_D_NOFIT;
    PROF_COUNT(0, 1);

    // stage the input of this round (coalesced, clamped inside the stream, zero padded)
    const uint32_t inbyte = (B >> 3) & ~3u;
    group_sync();
#ifdef DEFL_PAR_PROFILE
    __builtin_amdgcn_s_waitcnt(0);                   // everything outstanding (diagnostic split of the staging time)
    PROF_MARK(9);
#endif
    {
      constexpr int NW = (PAR_IN + 63) / 64;
      uint32_t      raw[NW], sh[NW];
#pragma unroll
      for (int u = 0; u < NW; u++) {                 // all loads first: one memory round trip
        uint32_t off = inbyte + 4u * (uint32_t)(u * 64 + lane);
        uint32_t q   = off < r.ilen4 ? off : r.ilen4;
        __builtin_memcpy(&raw[u], r.src + q, 4);
        sh[u] = off - q;
      }
#pragma unroll
      for (int u = 0; u < NW; u++) {
        uint32_t w   = (uint32_t)(u * 64 + lane);
        uint32_t val = sh[u] >= 4 ? 0u : raw[u] >> (8u * sh[u]);
        uint32_t k = w & (uint32_t)(PAR_G - 1), col = w / (uint32_t)PAR_G;
        if (col < 64u) P.in[k * 64u + col] = val;
        if (k < 4u && col >= 1u && col <= 64u) P.in[(k + (uint32_t)PAR_G) * 64u + col - 1u] = val;
      }
    }
    group_sync();
    PROF_MARK(0);
    PAR_PRIO(DEFL_PAR_PRIO_STEPS);
    ParCtx c;
    c.in         = P.in;
    c.column     = opaque(((uint32_t)lane & 63u) << 2);   // (its place in the address stays an OR: and-or)
    c.colbit     = inbyte * 8u + (uint32_t)lane * PAR_S;
    c.total_bits = total_bits;
    c.lit        = lit;
    c.dist       = dist;

    // 1. scan, 2. fix up
    const uint32_t limit = B + (uint32_t)(lane + 1) * PAR_S;
    uint32_t       start = B + (uint32_t)lane * PAR_S;
    ParScan        sc;
    {
      // lane 0 starts on the true chain; the other lanes only learn where a chain that starts
      // at their guess ends (no byte counts, no checkpoints)
      const uint32_t steps = par_scan<false>(c, true, B, start, limit, quarter, sc);
      const uint32_t q4    = (steps + 3u) / 4u;      // the full scans aim their checkpoints at these
      quarter = q4 < 4u ? 4u : q4;
    }
    PROF_MARK(1);
    bool     counted = false;
    uint32_t trunc   = 64u;                          // lanes in front of the first one that still starts wrong
    int      iter;
    for (iter = 0; iter < 65; iter++) {
      uint32_t pe       = wave_read(sc.exit, lane ? lane - 1 : 0);
      uint64_t stopmask = __ballot(sc.stop != 0);
      bool     dead     = (stopmask & ((1ull << lane) - 1ull)) != 0;   // behind a lane that stopped
      // every lane does one full scan at least, then rescans whenever its predecessor leaves at
      // another position than it started from (inside its own column: a predecessor that
      // stopped early has nothing to tell).  Lanes behind a stop keep scanning, the stop may be
      // an artefact of a chain that started wrong; they just do not keep the loop alive.
      const bool moved  = lane > 0 && pe != start && pe >= limit - PAR_S;
      bool     changed  = moved || !counted;
      const uint64_t chg = __ballot(changed && !dead);
      if (!chg) break;
      // a round that needs many scans ends in front of the first lane that still starts wrong
      // (the lanes before it hold the true chain; lane 0 always does, and every scan adds one
      // lane at least), the rest waits for the next round.  Codes that re-synchronise badly
      // (near fixed-length codes) thus cost short rounds, never the stream: handing such a stream
      // to the serial passes was measured slower on every workload tried, and one stream in a
      // serial pass takes about as long as a whole batch in this one.
      if (iter > PAR_MAX_FIXUP) {
        trunc = (uint32_t)__builtin_ctzll(chg);
        break;
      }
      PROF_COUNT(1, 1);
      start   = moved ? pe : start;
      counted = counted || changed;
      par_scan<true>(c, changed, B, start, limit, quarter, sc);
    }

    PROF_MARK(2);
    PROF_HIST(iter);
    // 3. place: lanes up to the first one that stopped inside its sub-sequence hold the true
    //    chain (the lanes behind it never learnt where to start: they wait for the next round)
    const uint64_t stopmask   = __ballot(sc.stop != 0);
    const uint32_t first_stop = stopmask ? (uint32_t)__builtin_ctzll(stopmask) : 64u;
    const bool     cut        = trunc <= first_stop; // the round was cut short in front of any stop
    const uint32_t why        = !cut && first_stop < 64 ? wave_read(sc.stop, first_stop) : 0u;
    if (why == 2) {                                  // malformed on the true chain
      PROF_BAIL(29);
      if (NOTE && lane == 0) {
        note[1] = B;
        note[2] = pos;
      }
      return INFL_D_INVALID;
    }
    if (why == 3 && first_stop < DEFL_PAR_STRETCH_LANES) {                // very long matches: the serial copy loops do better
      if (!serial_stretch(B)) return INFL_D_INVALID;
      continue;
    }
    const uint32_t nact = cut ? trunc : first_stop < 64 ? first_stop + 1u : 64u;
    uint32_t       cum  = (uint32_t)lane < nact ? sc.nout : 0;
    for (uint32_t dlt = 1; dlt < 64; dlt <<= 1) {
      uint32_t t = wave_read(cum, (uint32_t)lane >= dlt ? lane - dlt : 0);
      if ((uint32_t)lane >= dlt) cum += t;
    }
    if (wave_read(cum, nact - 1u) > cap - pos) {     // more output than capacity
      if (NOTE && lane == 0) {
        note[1] = B;
        note[2] = pos;
      }
      return INFL_D_INVALID;
    }
    PROF_MARK(3);

    // The round's output is produced in groups of up to 16 consecutive chains whose bytes fit
    // the link buffer; four lanes share a chain (one quarter each) while the links are written.
    // Where the bytes of the next 32 chains fit (data that compresses badly: few bytes to a chain), the
    // group takes all of them with two lanes to a chain (one half each): as many trips through the emit
    // loop for twice the chains, half as many groups.
    uint32_t a = 0;
    bool     stretched = false;
    while (a < nact) {
      const uint32_t base_cum = a ? wave_read(cum, a - 1u) : 0u;
      const bool     fits     = (uint32_t)lane >= a && (uint32_t)lane < nact && (uint32_t)lane < a + 32u &&
                            cum - base_cum <= (uint32_t)PAR_R;
      const uint32_t n32 = (uint32_t)__popcll(__ballot(fits)), rest = nact - a;
#ifdef DEFL_PAR_NO_PAIRS                             // (experiments: always four lanes to a chain)
      const bool two = false;
#else
#ifndef DEFL_PAR_PAIR_MIN
#define DEFL_PAR_PAIR_MIN 32                         // chains that have to fit for a group of pairs (fewer: only at a round's end)
#endif
      const bool two = n32 > 16u && (n32 == (rest < 32u ? rest : 32u) || n32 >= (uint32_t)DEFL_PAR_PAIR_MIN);
#endif
      const uint32_t nch = two ? n32 : n32 < 16u ? n32 : 16u;
      if (nch == 0) {                                // a chain of more bytes than the link buffer takes
        if (!serial_stretch(wave_read(start, a))) return INFL_D_INVALID;
        stretched = true;
        break;
      }
      const uint32_t total_out = wave_read(cum, a + nch - 1u) - base_cum;
      PROF_COUNT(3, nch);

      // 4. emit: clear the group's links, then walk the chains once more and write one "head"
      //    per symbol at the symbol's first byte: a literal value, or a match as
      //    LK_EXT | (distance - 1).  Bytes inside a match stay 0.
      group_sync();
      const uint32_t padded = (total_out + 255u) & ~255u;    // the per-byte passes work on whole 256-entry batches
      for (uint32_t i = (uint32_t)lane * 8u; i < padded; i += 512) {
        uint4 z = {0u, 0u, 0u, 0u};
        __builtin_memcpy(&P.link[i], &z, 16);
      }
      bool bad = false;
      {
        // the lane's stretch of its chain: quarters q .. qe - 1 (one quarter, or two where two lanes share a chain)
        const uint32_t ch  = a + (two ? (uint32_t)lane >> 1 : (uint32_t)lane >> 2);
        const uint32_t q   = two ? ((uint32_t)lane & 1u) << 1 : (uint32_t)lane & 3u, qe = two ? q + 2u : q + 1u;
        const uint32_t chl = ch < 64u ? ch : 63u;
        const uint32_t cs  = wave_read(start, chl), ce = wave_read(sc.exit, chl);
        const uint32_t o0  = wave_read(cum - sc.nout, chl) - base_cum;
        const uint32_t k1 = wave_read(sc.ck[0], chl), k2 = wave_read(sc.ck[1], chl), k3 = wave_read(sc.ck[2], chl);
        const uint32_t kb = q == 1 ? k1 : q == 2 ? k2 : k3;      // state at the stretch's begin
        const uint32_t ke = qe == 1 ? k1 : qe == 2 ? k2 : k3;    // and at its end
        const uint32_t pb = q ? B + (kb >> 12) : cs;
        const uint32_t pe = qe < 4 ? B + (ke >> 12) : ce;
        ParLane  rl;
        ParCtx   cq  = c;                            // the chain's column instead of the lane's own
        uint32_t oh  = 0;
        bool     act = ch < a + nch && pb < pe;
        cq.column = opaque(chl << 2);
        cq.colbit = inbyte * 8u + chl * PAR_S;
        group_sync();
        // (the lane counts in entries of the link buffer: its chain's offset in the group is in rl.o from the start)
        pl_init(rl, cq, act ? pb : B + chl * PAR_S, o0 + (q ? kb & 0xfffu : 0u), act ? pe : B + chl * PAR_S, act);
        // A distance that reaches before the start of the output can only occur within the first 32 KiB of output:
        // behind them the walk runs without the test (two copies of the loop, chosen once per group).
        auto walk = [&](auto near_start) {
          do {                                       // (tested at the end: no copies of the lane's state around the loop)
            if (rl.p < rl.lim) {
              uint32_t       value;
              const uint32_t at  = rl.o;
              const bool     isd = pl_in_match(rl);
              pl_step<2>(rl, cq, value);
              // one store serves both kinds of head: a literal at its own byte, a match (once its
              // distance is known) at the byte the length code stood for, which is where the step
              // before this one began; a length code stores nothing
              const uint32_t where = isd ? oh : at;
              const uint32_t head  = (isd ? LK_EXT : LK_LIT) | value;
              if (rl.lim == rl.limit) P.link[where] = (uint16_t)head;
              oh = at;
              if constexpr (decltype(near_start)::value) {
                const uint32_t dist = isd ? value + 1u : 0u;
                if (dist > pos + where) {            // distance reaches before the output start
                  bad    = true;
                  rl.lim = 0;
                }
              }
            }
          } while (__ballot(rl.p < rl.lim));
        };
        if (pos < 32768u) walk(std::true_type{});
        else walk(std::false_type{});
      }
      if (__ballot(bad)) {                           // (nothing of this group has been written yet)
        if constexpr (NOTE) {
          const uint32_t from = wave_read(start, a);
          if (lane == 0) {
            note[1] = from;
            note[2] = pos;
          }
        }
        return INFL_D_INVALID;
      }
      group_sync();
      PROF_MARK(4);
      PAR_PRIO(DEFL_PAR_PRIO_BYTES);

      // 5a. expand: every byte finds the head at or before it and becomes a literal, an index of an earlier
      //     byte of the group, or LK_EXT | how far before it.  A lane takes four consecutive entries of a
      //     256-entry batch: the last head among its four goes into a maximum scan over the lanes (key =
      //     lane : head, so the maximum over the lanes below is the nearest head before the lane's entries,
      //     with its value; the last head of earlier batches is seeded into lane 0), the four entries then take
      //     their heads one from the other, and the link arithmetic runs on two entries to a register.
      {
        // A byte inside a match with head LK_EXT | (dd - 1) at entry i becomes i - dd where that is an entry of
        // the group, else LK_EXT | (dd - i - 1).  In 16 bits: t = i - dd = (i + 0x7fff) - head, and a negative t
        // with its low 15 bits flipped is LK_EXT | (-t - 1).  Entries behind the end of the group (the batch's
        // padding) become copies of the last symbol's links: valid, never written out.
        constexpr uint32_t K15 = 0x7fff7fffu;
        uint32_t carry = 0;
        uint32_t at01  = ((4u * (uint32_t)lane + 1u + 0x7fffu) << 16) | (4u * (uint32_t)lane + 0x7fffu);   // i + 0x7fff, entries 0 and 1
        for (uint32_t base = 0; base < total_out; base += 256) {
          uint2 q;
          __builtin_memcpy(&q, &P.link[base + 4u * (uint32_t)lane], 8);
          const uint32_t h0 = q.x & 0xffffu, h1 = q.x >> 16, h2 = q.y & 0xffffu, h3 = q.y >> 16;
          uint32_t last = h2 ? h2 : h1 ? h1 : h0;
          last          = h3 ? h3 : last;
          const uint32_t seed = lane == 0 ? carry : 0u;
          const uint32_t s    = wave_scan_max(last ? ((uint32_t)lane << 16) | last : seed);
          carry = (uint32_t)__builtin_amdgcn_readlane((int)s, 63) & 0xffffu;
          // the nearest head before the lane's entries: the scan's result one lane down
          const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp((int)seed, (int)s, 0x138 /* wave_shr:1 */, 0xf, 0xf, false) & 0xffffu;
          const uint32_t c0 = h0 ? h0 : before, c1 = h1 ? h1 : c0, c2 = h2 ? h2 : c1, c3 = h3 ? h3 : c2;
          const uint32_t c01 = (c1 << 16) | c0, c23 = (c3 << 16) | c2;
          const uint32_t t01 = pk_sub16(at01, c01), t23 = pk_sub16(at01 + 0x00020002u, c23);
          const uint32_t o01 = t01 ^ (pk_sign16(t01) & K15), o23 = t23 ^ (pk_sign16(t23) & K15);
          const uint32_t m01 = pk_sign16(c01), m23 = pk_sign16(c23);            // a match head (LK_EXT is the top bit)
          q.x = (o01 & m01) | (c01 & ~m01);
          q.y = (o23 & m23) | (c23 & ~m23);
          __builtin_memcpy(&P.link[base + 4u * (uint32_t)lane], &q, 8);
          at01 += 0x01000100u;                       // the next batch: 256 entries on
        }
      }
      group_sync();
      PROF_MARK(5);

      // 5b. pointer jumping: every link ends at a literal or at a byte from before the group
      //     (both are position independent, so they can be copied along the chains).  Straight-
      //     line bodies (every access is made, harmless ones where nothing is to do) so that
      //     PAR_U independent accesses are in flight at once.
      uint32_t dirty = ~0u;                          // batches that may still hold unresolved links
      for (int it = 0; it < 16 && dirty; it++) {
        PROF_COUNT(2, 1);
        uint32_t again = 0;
        for (uint32_t bt = 0; bt * 256u < padded; bt++) {
          if (!((dirty >> bt) & 1u)) continue;       // wave-uniform
          const uint32_t base = bt * 256u + (uint32_t)lane;
          uint32_t v[PAR_U], t[PAR_U];
          bool     more = false;
#pragma unroll
          for (int u = 0; u < PAR_U; u++) v[u] = P.link[base + 64u * u];
#pragma unroll                                   // a resolved entry reads itself: no select afterwards
          for (int u = 0; u < PAR_U; u++) t[u] = P.link[v[u] >= LK_LIT ? base + 64u * u : v[u]];
#pragma unroll
          for (int u = 0; u < PAR_U; u++) {
            P.link[base + 64u * u] = (uint16_t)t[u];
            more |= t[u] < LK_LIT;
          }
          if (__ballot(more)) again |= 1u << bt;
        }
        group_sync();
        dirty = again;
      }
      PROF_MARK(6);

      // 5c. write out, one byte per lane and access (neighbouring lanes, neighbouring bytes:
      //     the stores and most of the loads of bytes from before the group coalesce).  All loads
      //     first, the bytes parked four to a register, then all stores: loads and stores share
      //     one in-order counter, so a load behind a store would wait for the store to complete.
      //     The fetches of DEFL_PAR_FETCH_BATCHES batches are issued together, in straight-line code
      //     (a branch per batch makes one memory round trip per batch; registers limit the number).
      {
        constexpr int NB = PAR_R / 256;
        constexpr int HB = DEFL_PAR_FETCH_BATCHES;   // batches whose fetches are in flight together
        uint32_t      pk[NB];
#pragma unroll
        for (int h = 0; h < NB; h += HB) {
#pragma unroll
          for (int j = 0; j < HB; j++) pk[h + j] = 0;
          if ((uint32_t)h * 256u < total_out) {      // (wave-uniform)
            uint64_t q[HB];
            uint32_t g[HB][4];
#pragma unroll
            for (int j = 0; j < HB; j++) {           // four consecutive bytes per lane and batch
              __builtin_memcpy(&q[j], &P.link[(h + j) * 256 + 4 * lane], 8);
              // a batch behind the group's end holds what an earlier group left there: it fetches like literals
              if ((uint32_t)(h + j) * 256u >= total_out) q[j] = 0;
#pragma unroll
              for (int k = 0; k < 4; k++) {
                // a resolved match byte is LK_EXT | (how far before the group - 1); a literal fetches the stream's
                // first byte (the subtraction wraps for it and the minimum brings it back: no select)
                const uint32_t v    = (uint32_t)(q[j] >> (16 * k)) & 0xffffu;
#if defined(DEFL_DIAG_WRITEOUT) && DEFL_DIAG_WRITEOUT == 1     // (diagnostic, wrong output: every fetch hits the line before the group)
                const uint32_t back = pos ? 1u : 0u;
#else
                const uint32_t far  = v - (LK_EXT - 1u);
                const uint32_t back = far < pos ? far : pos;
#endif
#if defined(DEFL_DIAG_WRITEOUT) && (DEFL_DIAG_WRITEOUT == 2 || DEFL_DIAG_WRITEOUT == 4)     // (diagnostic, wrong output: no fetch at all)
                g[j][k] = back;
#else
                g[j][k] = dst[pos - back];
#endif
              }
            }
#pragma unroll
            for (int j = 0; j < HB; j++) {           // four bytes at a time: literals and fetched bytes, merged under a byte mask
              const uint32_t qx = (uint32_t)q[j], qy = (uint32_t)(q[j] >> 32);
              const uint32_t lit4  = __builtin_amdgcn_perm(qy, qx, 0x06040200u);
              const uint32_t ext4  = __builtin_amdgcn_perm(pk_sign16(qy), pk_sign16(qx), 0x06040200u);
              const uint32_t got4  = g[j][0] | (g[j][1] << 8) | (g[j][2] << 16) | (g[j][3] << 24);
              pk[h + j] = (got4 & ext4) | (lit4 & ~ext4);
            }
          }
        }
#if defined(DEFL_DIAG_WRITEOUT) && DEFL_DIAG_WRITEOUT >= 3     // (diagnostic, wrong output: one store per lane and group instead of eight;
        {                                                      //  4: no fetch either and a store only where the bytes happen to cancel)
          uint32_t x = 0;
#pragma unroll
          for (int bt = 0; bt < NB; bt++) x ^= pk[bt];
          if (4u * (uint32_t)lane + 4u <= total_out && (DEFL_DIAG_WRITEOUT == 3 || x == 0x12345678u)) __builtin_memcpy(dst + pos + 4 * lane, &x, 4);
        }
#pragma unroll
        for (int bt = NB; bt < NB; bt++) {
#else
#pragma unroll
        for (int bt = 0; bt < NB; bt++) {
#endif
          const uint32_t i = (uint32_t)(bt * 256 + 4 * lane);
          if (i + 4u <= total_out) {
            __builtin_memcpy(dst + pos + i, &pk[bt], 4);
          } else {
#pragma unroll
            for (int k = 0; k < 3; k++)
              if (i + (uint32_t)k < total_out) dst[pos + i + k] = (uint8_t)(pk[bt] >> (8 * k));
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      PROF_MARK(7);
      PAR_PRIO(DEFL_PAR_PRIO_STEPS);
      pos += total_out;
      a += nch;
    }

    if (stretched) continue;                         // (B and done are set)
    boost = 0;
    B     = wave_read(sc.exit, nact - 1u);
    done  = why == 1;                                // the end-of-block code has been consumed
  }
  PROF_FLUSH;
  pos_io = pos;
  rd_seek(r, r.src + (B >> 3));                      // put the serial reader behind the block
  rd_fill(r);
  rd_drop(r, B & 7u);
  return INFL_D_OK;
}

// Block loop (infl.c:1063-1092).  SYM selects the symbol loop: 0 = huffman_block (any table
// format, exact failure reasons), 1 = huffman_block_compact, 2 = huffman_block_parallel.
// SYM 3 = 2 for the pass that looks at failing streams a second time: lane 0 notes where a later pass can take the
// stream up again: note[0] = bit position of the header of the block in hand, note[1] = bit position of a symbol of
// that block on the true chain (0 = the block's first), note[2] = the output position there, note[3] = 1.
template <int T, class F, class R, int SYM>
__device__ __forceinline__ int block_loop(R &r, uint8_t *dst, uint32_t &pos, uint32_t cap,
                                          StreamLds<F> &L, ParLds *par, int lig, int gshift,
                                          uint64_t &fuel, uint32_t *note) {
  bool     fixed_ready = false;
  bool     short_block = false;                    // the Huffman block before this one took under 1024 input bits
  uint32_t bfinal = 0;
  while (!bfinal) {
    if (fuel-- == 0) return INFL_D_INTERNAL;
    rd_fill(r);
    if constexpr (SYM == 3) {
      if (lig == 0) {
        note[0] = r.ipos * 8u - r.n;
        note[1] = 0;
        note[2] = pos;
        note[3] = 1;
        note[4] = pos;                               // (the block's own start, for callers without an inpos array)
      }
    }
    if (r.n < 3) return INFL_D_NEED_INPUT;
    bfinal = (uint32_t)r.bits & 1u;
    uint32_t btype = ((uint32_t)r.bits >> 1) & 3u;
    rd_drop(r, 3);
    int rc;
    if (btype == 0) {
      rc = stored_block<T>(r, dst, pos, cap, lig);
    } else if (btype == 3) {
      rc = INFL_D_INVALID;
    } else {
      if (btype == 1) {
        rc = INFL_D_OK;
        if (!fixed_ready) {
          int b = setup_fixed<T, F>(L, lig, gshift);
          rc = b == BUILD_OK ? INFL_D_OK : b == BUILD_NOFIT ? INFL_D_NOFIT : INFL_D_INTERNAL;
          fixed_ready = rc == INFL_D_OK;
        }
      } else {
        fixed_ready = false;
#ifdef DEFL_PAR_PROFILE
        const uint64_t setup_t0 = __builtin_readcyclecounter();
#endif
        rc = setup_dynamic<T, F>(r, L, lig, gshift);
#ifdef DEFL_PAR_PROFILE
        if (SYM >= 2 && lig == 0) atomicAdd(&g_par_prof[14], (unsigned long long)(__builtin_readcyclecounter() - setup_t0));
#endif
      }
#ifdef DEFL_PAR_PROFILE
      if (SYM >= 2 && rc == INFL_D_NOFIT && lig == 0) atomicAdd(&g_par_prof[15], 1ull);   // tables too large for the compact format
#endif
      if (rc == INFL_D_OK) {
        if constexpr (SYM >= 2) {
          const uint32_t at = r.ipos * 8u - r.n;
          rc = huffman_block_parallel<SYM == 3>(r, dst, pos, cap, L.lit, L.dist, *par, lig, short_block, note);
          short_block = r.ipos * 8u - r.n - at < 1024u;
        } else if constexpr (SYM == 1) {
          rc = huffman_block_compact<T>(r, dst, pos, cap, L.lit, L.dist, lig) == COMPACT_END ? INFL_D_OK
                                                                                              : INFL_D_NOFIT;
        } else {
          rc = huffman_block<T, F>(r, dst, pos, cap, L, lig, fuel);
        }
      }
    }
    if (rc != INFL_D_OK) return rc;
  }
  return INFL_D_OK;
}

__device__ __forceinline__ bool zlib_header_ok(const uint8_t *p) {   // infl.c:233-238
  uint32_t cmf = p[0], flg = p[1];
  return (cmf & 15u) == 8u && (cmf >> 4) <= 7u && ((cmf << 8) + flg) % 31u == 0 && !(flg & 0x20u);
}

__device__ __forceinline__ void rd_init(Reader &r, const uint8_t *src, uint32_t srclen, int, int) {
  r.end = src + srclen;
}
template <int T>
__device__ __forceinline__ void rd_init(FastReader<T> &r, const uint8_t *src, uint32_t srclen, int lig,
                                        int gshift) {
  r.src    = src;
  r.ilen   = srclen;
  r.ilen4  = srclen - 4;
  r.lig    = lig;
  r.gshift = gshift;
}

// Whole stream: the dispatcher of infl() for one contiguous input.
// Returns the detail code.  `outlen` is what infl_output_pos() would report
// after infl(); `pos` is the output position actually reached and `cur` the
// first input byte with unread bits (both also valid when the stream stopped
// early; the incremental API reports them).
template <int T, class F, class R, int SYM>
__device__ __forceinline__ int inflate_stream(const uint8_t *src, uint32_t srclen, uint8_t *dst,
                                              uint32_t cap, bool zlib, StreamLds<F> &L,
                                              ParLds *par, int lig, int gshift,
                                              uint32_t &outlen, uint32_t &pos, const uint8_t *&cur,
                                              uint32_t *note = nullptr) {
  outlen = 0;
  pos    = 0;
  cur    = src;
  if (SYM == 3 && lig == 0) note[3] = 0;
  if (srclen == 0) return INFL_D_OK;               // infl.c:1547-1550
  // The fast reader needs a few bytes.  In a batch that fills the device the parallel decoder leaves short streams to the
  // serial pass behind it, several to a wavefront: one of its rounds costs the same for three busy lanes as for 64, and
  // so does the set-up of a block.
  if (SYM != 0 && srclen < ((SYM == 2 || SYM == 3) ? PAR_MIN_INPUT : 8u)) return INFL_D_NOFIT;   // (SYM 4: a small batch)
  if (SYM >= 2 && srclen >= (1u << 28)) return INFL_D_NOFIT;   // 32-bit bit positions

  const uint8_t *p = src, *end = src + srclen;
  uint64_t fuel = (uint64_t)srclen * 8u + 64u;     // every block consumes >= 1 input bit
  bool     try_stored = true;                      // infl.c:1519-1531
  if (!zlib) try_stored = ((p[0] >> 1) & 3u) == 0;
  else if (srclen >= 3) try_stored = ((p[2] >> 1) & 3u) == 0;

  R r;
  rd_init(r, src, srclen, lig, gshift);
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
  rc  = block_loop<T, F, R, SYM>(r, dst, pos, cap, L, par, lig, gshift, fuel, note);
  if (SYM == 3 && outlen && lig == 0) note[3] = 0;   // (a stored prefix has been published: that stream starts over)
  if (rc == INFL_D_OK && rd_overrun(r)) rc = INFL_D_NEED_INPUT;
  cur = rd_byte_ptr(r);
  if (rc == INFL_D_OK) outlen = pos;
  return rc;
}


// The last pass's way into a stream that the parallel pass has found to fail (INFL_STATUS_DEFERRED_RESUME): the
// output up to `pos` is in place, the block whose header stands at bit position `hdr` holds the failure, and `bit` is
// a symbol boundary of that block in front of it (0 = the block's first symbol).  Builds the block's tables again and
// decodes on from there, one symbol after the other, to the exact result a decode from the start would give.
template <int T>
__device__ __forceinline__ int inflate_from(const uint8_t *src, uint32_t srclen, uint8_t *dst, uint32_t cap,
                                            uint32_t hdr, uint32_t bit, uint32_t &pos, StreamLds<RefFmt> &L, int lig,
                                            int gshift, const uint8_t *&cur) {
  typedef RefFmt F;
  Reader r;
  r.end = src + srclen;
  uint64_t fuel = (uint64_t)srclen * 8u + 64u;
  int      rc;
  rd_seek(r, src + (hdr >> 3));
  rd_fill(r);
  rd_drop(r, hdr & 7u);
  for (bool first = true;; first = false) {
    if (fuel-- == 0) {
      rc = INFL_D_INTERNAL;
      break;
    }
    rd_fill(r);
    if (r.n < 3) {
      rc = INFL_D_NEED_INPUT;
      break;
    }
    const uint32_t bfinal = (uint32_t)r.bits & 1u, btype = ((uint32_t)r.bits >> 1) & 3u;
    rd_drop(r, 3);
    if (btype == 0) {
      rc = stored_block<T>(r, dst, pos, cap, lig);
    } else if (btype == 3) {
      rc = INFL_D_INVALID;
    } else {
      if (btype == 1) {
        const int b = setup_fixed<T, F>(L, lig, gshift);
        rc = b == BUILD_OK ? INFL_D_OK : INFL_D_INTERNAL;
      } else {
        rc = setup_dynamic<T, F>(r, L, lig, gshift);
        if (rc == INFL_D_NOFIT) rc = INFL_D_INTERNAL;
      }
      if (rc == INFL_D_OK) {
        if (first && bit) {
          rd_seek(r, src + (bit >> 3));
          rd_fill(r);
          rd_drop(r, bit & 7u);
        }
        rc = huffman_block<T, F>(r, dst, pos, cap, L, lig, fuel);
      }
    }
    if (rc != INFL_D_OK || bfinal) break;
  }
  cur = rd_byte_ptr(r);
  return rc;
}

// Copies the bytes a stream produced to the same offset of a second arena (host memory mapped
// into the device's address space: the copy then is the stream's way to the host, made by the
// lanes that decoded it, 16 bytes per lane and store once the addresses are aligned).
template <int T>
__device__ __forceinline__ void mirror_out(uint8_t *to, const uint8_t *from, uint64_t arena_off, uint32_t n,
                                           int lig) {
  uint32_t head = (16u - (uint32_t)(arena_off & 15u)) & 15u;
  if (head > n) head = n;
  if ((uint32_t)lig < head) to[lig] = from[lig];
  if (T < 16 && (uint32_t)lig + T < head) to[lig + T] = from[lig + T];
  const uint32_t body = (n - head) & ~15u;
  for (uint32_t k = head + 16u * (uint32_t)lig; k < head + body; k += 16u * T) {
    uint4 v;
    __builtin_memcpy(&v, from + k, 16);
    __builtin_memcpy(to + k, &v, 16);
  }
  for (uint32_t k = head + body + (uint32_t)lig; k < n; k += T) to[k] = from[k];
}

// flags: bit 0 = zlib wrapper (INFL_ZLIB); INFL_F_PROGRESS = publish the position
// reached in `outlen` even when the stream stopped early (incremental API).
//
// MODE 0: decode every stream and write its result.
// MODE 1: first pass with the compact tables.  A stream that completes writes
//         its result; any other stream is marked INFL_STATUS_DEFERRED.
// MODE 2: last pass: only streams marked INFL_STATUS_DEFERRED, INFL_STATUS_DEFERRED_FAILING or
//         INFL_STATUS_DEFERRED_RESUME are decoded (the last kind from where the parallel pass says).
// MODE 3: like MODE 1, but a whole wavefront decodes each Huffman block in parallel (T = 64).
// MODE 4: like MODE 1, but only for streams marked INFL_STATUS_DEFERRED (by MODE 3; a stream it
//         found to fail carries INFL_STATUS_DEFERRED_FAILING and waits for MODE 2).
// MODE 5: MODE 3 once more for the streams marked INFL_STATUS_DEFERRED_FAILING, this time noting where the
//         failure lies: the mark becomes INFL_STATUS_DEFERRED_RESUME and MODE 2 goes on from there instead of
//         decoding the stream from its start, one symbol after the other (launched only when the caller asked
//         for detail codes or positions: a failing 64 KiB stream costs 2 ms instead of 15).
template <int T, class F, int MODE>
__global__ __launch_bounds__(64, (MODE == 3 || MODE == 5 || MODE == 6) ? 4 : 1) void infl_batch_kernel(const uint8_t *__restrict__ src,
                                                        const uint64_t *__restrict__ src_off,
                                                        uint8_t *__restrict__ dst,
                                                        const uint64_t *__restrict__ dst_off,
                                                        int flags, uint64_t n,
                                                        int32_t *__restrict__ status,
                                                        uint32_t *__restrict__ outlen,
                                                        int32_t *__restrict__ detail,
                                                        uint32_t *__restrict__ inpos,
                                                        uint8_t *__restrict__ mirror,
                                                        const uint32_t *__restrict__ skip,
                                                        const uint32_t *__restrict__ order) {
  constexpr int  G    = 64 / T;                    // streams per wavefront
  // MODE 6 = MODE 3 for a batch too small to fill the device: wavefronts are to be had, and a short stream is through
  // the parallel decoder sooner than through the serial pass (85 us against 90 .. 170 us for 128 .. 384 bytes), so it
  // hands over inputs of under 8 bytes only (SYM 4).
  constexpr int  SYM  = (MODE == 1 || MODE == 4) ? 1 : MODE == 3 ? 2 : MODE == 5 ? 3 : MODE == 6 ? 4 : 0;
  static_assert((MODE != 3 && MODE != 5 && MODE != 6) || T == 64, "the parallel block decoder uses the whole wavefront");
  __shared__ StreamLds<F> lds[G];
  ParLds   *parp = nullptr;
  uint32_t *note = nullptr;
  if constexpr (MODE == 3 || MODE == 5 || MODE == 6) {
    __shared__ ParLds par;
    parp = &par;
  }
  if constexpr (MODE == 5) {
    __shared__ uint32_t note_buf[5];
    note = note_buf;
  }

  const int      lane   = threadIdx.x;
  const int      group  = lane / T;
  const int      lig    = lane % T;
  const int      gshift = group * T;
  const uint64_t slot   = (uint64_t)blockIdx.x * G + group;

  if (slot >= n) return;
  // `order` (may be null): the streams in the order the workgroups are to take them (by size class, see
  // infl_order_* below): workgroups that start together then decode streams of similar length
  const uint64_t i = order ? (uint64_t)order[slot] : slot;
  bool     resume = false;
  uint32_t resume_hdr = 0, resume_bit = 0, resume_pos = 0;
  if (MODE == 2 || MODE == 4 || MODE == 5) {
    const int32_t mark = status[i];
    const bool    mine = MODE == 5 ? mark == INFL_STATUS_DEFERRED_FAILING
                                   : mark == INFL_STATUS_DEFERRED ||
                                         (MODE == 2 && (mark == INFL_STATUS_DEFERRED_FAILING || mark == INFL_STATUS_DEFERRED_RESUME));
    if (MODE == 2 && mark == INFL_STATUS_DEFERRED_RESUME) {          // (set only when `detail` exists)
      resume     = true;
      resume_pos = outlen[i];
      resume_hdr = (uint32_t)detail[i];
      resume_bit = inpos ? inpos[i] : 0u;            // (without an inpos array: the block's first symbol)
    }
    group_sync();                                    // every lane has looked before one of them writes the result
    if (MODE == 2 && mark == INFL_STATUS_DEFERRED_DONE) {            // finished by the windowed serial pass
      if (lig == 0) status[i] = 0;
      return;
    }
    if (!mine) return;
  }

  uint64_t s0 = src_off[i], s1 = src_off[i + 1];
  uint64_t d0 = dst_off[i], d1 = dst_off[i + 1];
  if (skip) {                                        // a container header in front of the stream (gzip): not ours to decode
    const uint64_t k = skip[i];
    s0 = k < s1 - s0 ? s0 + k : s1;
  }
  if constexpr (T == 64) {                           // one stream per wavefront: keep its bounds in scalar
    s0 = wave_uniform(s0);                           // registers, so that addresses are base + 32-bit offset
    s1 = wave_uniform(s1);
    d0 = wave_uniform(d0);
    d1 = wave_uniform(d1);
  }
  uint32_t       out = 0, pos = 0;
  const uint8_t *cur = nullptr;
  int rc;
  if constexpr (SYM != 0) {
    rc = inflate_stream<T, F, FastReader<T>, SYM>(src + s0, (uint32_t)(s1 - s0), dst + d0,
                                                  (uint32_t)(d1 - d0), (flags & 0xff) == 1,
                                                  lds[group], parp, lig,
                                                  gshift, out, pos, cur, note);
  } else if (MODE == 2 && resume) {
    if constexpr (MODE == 2) {
      pos = resume_pos;
      rc  = inflate_from<T>(src + s0, (uint32_t)(s1 - s0), dst + d0, (uint32_t)(d1 - d0), resume_hdr, resume_bit, pos,
                            lds[group], lig, gshift, cur);
      if (rc == INFL_D_OK) out = pos;
    }
  } else {
    rc = inflate_stream<T, F, Reader, 0>(src + s0, (uint32_t)(s1 - s0), dst + d0,
                                         (uint32_t)(d1 - d0), (flags & 0xff) == 1, lds[group],
                                         nullptr, lig, gshift, out, pos, cur);
  }
  if constexpr (MODE == 5) {
    // Three numbers tell the last pass where to take the stream up again: outlen[i], detail[i] and inpos[i].  A
    // caller without an inpos array gives room for two: the last pass then goes on from the first symbol of the
    // block in hand (nothing is ever parked in the caller's output area).  Anything else leaves the mark as it is.
    if (lig == 0 && rc != INFL_D_OK && rc != INFL_D_NOFIT && rc != INFL_D_INTERNAL && note[3] && detail != nullptr) {
      outlen[i] = inpos ? note[2] : note[4];
      detail[i] = (int32_t)note[0];
      if (inpos) inpos[i] = note[1];
      status[i] = INFL_STATUS_DEFERRED_RESUME;
    }
    return;
  }
  if ((MODE == 1 || MODE == 3 || MODE == 4 || MODE == 6) && rc != INFL_D_OK) {
    // The parallel pass reports a stream that certainly fails by itself when the caller wants
    // the reference's result only (UNZ_ERR and the output position infl() leaves, which does not
    // depend on why or where the stream failed); the reason, the input position and the
    // position reached are worked out by the serial passes.
    const bool exact = detail != nullptr || inpos != nullptr || (flags & INFL_F_PROGRESS);
    const bool undecided = (MODE != 3 && MODE != 6) || rc == INFL_D_NOFIT || rc == INFL_D_INTERNAL;
    if (undecided || exact) {
      if (lig == 0) status[i] = undecided ? INFL_STATUS_DEFERRED : INFL_STATUS_DEFERRED_FAILING;
      return;
    }
  }
  const uint32_t produced = ((flags & INFL_F_PROGRESS) && rc != INFL_D_OK && rc != INFL_D_INVALID) ? pos : out;
  if (lig == 0) {
    status[i] = rc == INFL_D_OK ? 0 : -1;          // every failure is UNZ_ERR (infl.c:1074-1087)
    outlen[i] = produced;
    if (detail) detail[i] = rc;
    if (inpos) inpos[i] = (uint32_t)(cur - (src + s0));
  }
  if (mirror) {
    group_sync();                                    // the lanes read what other lanes of the group wrote
    mirror_out<T>(mirror + d0, dst + d0, d0, produced, lig);
  }
}

// The serial pass behind the parallel one (MODE 4 of infl_batch_kernel) for groups of 8 or 16 lanes.  The streams it is
// for lie anywhere in the batch, often a few among many.  The wavefronts of a window of kDeferWindow streams share the
// window's marked streams out by rank: the first wavefront's groups take the first 64 / T marked ones, the next
// wavefront's the following ones, and a wavefront with nothing left ends at once; so no group idles beside a busy one
// wherever neighbours are not all marked, and a window of marked streams only is handled as by the plain kernel.
// A stream this pass cannot finish keeps its mark for the last pass; one it finishes gets INFL_STATUS_DEFERRED_DONE, which
// the last pass (it looks at every stream) turns into 0.
constexpr int kDeferWindow = 64;
template <int T>
__global__ __launch_bounds__(64, 1) void infl_deferred_window_kernel(const uint8_t *__restrict__ src,
                                                                      const uint64_t *__restrict__ src_off,
                                                                      uint8_t *__restrict__ dst,
                                                                      const uint64_t *__restrict__ dst_off, int flags,
                                                                      uint64_t n, int32_t *__restrict__ status,
                                                                      uint32_t *__restrict__ outlen,
                                                                      int32_t *__restrict__ detail,
                                                                      uint32_t *__restrict__ inpos,
                                                                      uint8_t *__restrict__ mirror,
                                                                      const uint32_t *__restrict__ skip) {
  typedef CompactFmt F;
  constexpr int G = 64 / T;
  static_assert(T <= kDeferWindow && kDeferWindow % T == 0 && kDeferWindow <= 64, "one mask word per window");
  __shared__ StreamLds<F> lds[G];
  const int lane = threadIdx.x, group = lane / T, lig = lane % T, gshift = group * T;

  // every group works the marks out for itself (the same ones)
  constexpr int  WPW    = kDeferWindow / G;          // wavefronts per window
  const uint64_t base   = (uint64_t)(blockIdx.x / WPW) * kDeferWindow;
  uint64_t       marked = 0;
#pragma unroll
  for (int w = 0; w < kDeferWindow; w += T) {
    const uint64_t k    = base + (uint64_t)(w + lig);
    const int32_t  mark = k < n ? status[k] : 0;
    const bool     mine = mark == INFL_STATUS_DEFERRED || mark == INFL_STATUS_DEFERRED_DONE;
    marked |= ((__ballot(mine) >> gshift) & ((1ull << T) - 1ull)) << w;
  }
  group_sync();                                      // every mark has been read before a result is written
  // (the window's wavefronts take their turns in an order that changes from window to window: workgroups go to the
  // chip's eight parts by their index, and the busy ones should not all have the same index modulo 8)
  const int rank = (int)((blockIdx.x + blockIdx.x / WPW) % WPW) * G + group;
  if ((int)__popcll(marked) <= rank) return;         // (a whole wavefront, or its last groups: nothing left for them)
  for (int k = 0; k < rank; k++) marked &= marked - 1ull;
  {
    const uint64_t i = base + (uint64_t)__builtin_ctzll(marked);

    uint64_t       s0 = src_off[i], s1 = src_off[i + 1];
    const uint64_t d0 = dst_off[i], d1 = dst_off[i + 1];
    if (skip) {                                      // a container header in front of the stream (gzip)
      const uint64_t k = skip[i];
      s0 = k < s1 - s0 ? s0 + k : s1;
    }
    uint32_t       out = 0, pos = 0;
    const uint8_t *cur = nullptr;
    const int rc = inflate_stream<T, F, FastReader<T>, 1>(src + s0, (uint32_t)(s1 - s0), dst + d0, (uint32_t)(d1 - d0),
                                                          (flags & 0xff) == 1, lds[group], nullptr, lig, gshift, out, pos,
                                                          cur);
    if (rc == INFL_D_OK) {
      if (lig == 0) {
        // (not 0 yet: the window's other wavefronts, which may start much later, must still see that this stream was
        // one of the marked ones, or they would count their ranks differently)
        status[i] = INFL_STATUS_DEFERRED_DONE;
        outlen[i] = out;
        if (detail) detail[i] = rc;
        if (inpos) inpos[i] = (uint32_t)(cur - (src + s0));
      }
      if (mirror) {
        group_sync();                                // the lanes read what other lanes of the group wrote
        mirror_out<T>(mirror + d0, dst + d0, d0, out, lig);
      }
    }
  }
}

#ifdef DEFL_SMALL_BATCH_UNIT
}  // namespace defl

// This unit holds the parallel pass for batches too small to fill the device (MODE 6) and nothing else.  It is
// compiled apart from the other kernels because a second instantiation of the parallel kernel beside the first changes
// how the compiler treats the functions both call, and the first one's loops, which sit at the register limit, came
// out 0.5 % slower on the headline batch.
extern "C" int infl_launch_parallel_small(const void *src, const uint64_t *src_off, void *dst, const uint64_t *dst_off,
                                          int flags, uint64_t n, int32_t *status, uint32_t *outlen, int32_t *detail,
                                          uint32_t *inpos, void *mirror, void *stream, const uint32_t *skip,
                                          const uint32_t *order) {
  hipLaunchKernelGGL((defl::infl_batch_kernel<64, defl::CompactFmt, 6>), dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream,
                     (const uint8_t *)src, src_off, (uint8_t *)dst, dst_off, flags, n, status, outlen, detail, inpos,
                     (uint8_t *)mirror, skip, order);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
#else

// Marks every stream for the serial passes (variant 5 of infl_launch_batch: those passes alone).
__global__ __launch_bounds__(256) void infl_mark_kernel(int32_t *__restrict__ status, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i < n) status[i] = INFL_STATUS_DEFERRED;
}

// ---------------------------------------------------------------------------
// Incremental decode of ONE stream (infl_stream(), reference src/infl/stream.c:468-794): the
// input seen so far is decoded from where the last call stopped, symbol by symbol, by one
// wavefront with tables in the reference's geometry.  The state kept between calls is small:
// the bit position of the symbol to go on with, the output position, and the bit position of
// the header of the block that symbol belongs to (the tables are built again from it; the
// reference keeps its tables in the stream object instead, stream.c:348-352, 669-675).
// A stop is always at a block header or at a symbol boundary: a symbol that lacks input bits
// or output space is left for the next call as a whole.
// ---------------------------------------------------------------------------
//
// A call's traffic with the host rides on the launch itself when the host gives pinned buffers:
// the new input bytes are fetched from `fresh` (host memory the device can read) into the input
// arena at src[fresh_at ..) before the decode, the new output bytes (at most hout_cap of them) and
// the state are stored to `hout` / `hstate` after it, so a call is one launch and one wait.
template <int T>
__global__ __launch_bounds__(64, 1) void infl_resume_kernel(uint8_t *src, uint32_t srclen,
                                                             uint8_t *__restrict__ dst, uint32_t cap, int zlib,
                                                             InflResume *__restrict__ state,
                                                             const uint8_t *__restrict__ fresh, uint32_t fresh_at,
                                                             uint8_t *__restrict__ hout, uint32_t hout_cap,
                                                             InflResume *__restrict__ hstate,
                                                             uint4 *__restrict__ tables) {
  typedef RefFmt F;
  static_assert(T == 64, "one wavefront decodes the stream");
  static_assert(offsetof(StreamLds<F>, dist) == sizeof(F::entry_t) * F::LIT_CAP &&
                    sizeof(F::entry_t) * (F::LIT_CAP + F::DIST_CAP) == INFL_RESUME_TABLE_BYTES,
                "the two tables lie one behind the other at the start of the working set");
  __shared__ __attribute__((aligned(16))) StreamLds<F> L;
  constexpr uint32_t TABLE_VEC = INFL_RESUME_TABLE_BYTES / 16u;
  const int lig = threadIdx.x;
  uint64_t  bit = state->bit, block_bit = state->block_bit;
  uint32_t  pos = state->pos, flags = state->flags;
  // Byte-wise feeding means a call per symbol or less: building the block's tables again from its header at every
  // call (tens of microseconds) was most of a call.  The tables of the block a call stops in stay in the stream's
  // table buffer; the call that resumes in that block loads them (10 KiB, coalesced).
  uint64_t  tables_bit = tables ? state->tables_bit : 0;
  bool      tables_fresh = false;                    // built by this call for the block in hand
  const uint32_t pos_before = pos;
  if (fresh && fresh_at < srclen)
    for (uint32_t k = (uint32_t)lig; k < srclen - fresh_at; k += T) src[fresh_at + k] = fresh[k];
  group_sync();                                      // every lane has read the state before lane 0 writes it
  int       rc = INFL_D_OK;
  uint64_t  fuel = (uint64_t)srclen * 8u + 64u;
  Reader    r;
  r.end = src + srclen;
  auto seek = [&](uint64_t at) {
    rd_seek(r, src + (at >> 3));
    rd_fill(r);
    rd_drop(r, (uint32_t)(at & 7u));
  };
  if (!(flags & INFL_RS_HEADER)) {                   // zlib wrapper (infl.c:1027-1039), empty for raw streams
    if (zlib) {
      if (srclen < 2) rc = INFL_D_NEED_INPUT;
      else if (!zlib_header_ok(src)) rc = INFL_D_INVALID;
      else bit = block_bit = 16;
    }
    if (rc == INFL_D_OK) flags |= INFL_RS_HEADER;
  }
  while (rc == INFL_D_OK && !(flags & INFL_RS_DONE)) {
    if (fuel-- == 0) {
      rc = INFL_D_INTERNAL;
      break;
    }
    // the block header (again, when the call resumes inside the block: it gives the tables)
    seek(block_bit);
    if (rd_real(r) < 3) {
      rc = INFL_D_NEED_INPUT;
      break;
    }
    const uint32_t bfinal = (uint32_t)r.bits & 1u, btype = ((uint32_t)r.bits >> 1) & 3u;
    rd_drop(r, 3);
    if (btype == 3) {
      rc = INFL_D_INVALID;
      break;
    }
    if (btype == 0) {                                // stored: carried out once it is there as a whole
      rc = stored_block<T>(r, dst, pos, cap, lig);
      if (rc != INFL_D_OK) break;
      bit = block_bit = rd_bitpos(r, src);
      flags &= ~INFL_RS_INBLOCK;
      if (bfinal) flags |= INFL_RS_DONE;
      continue;
    }
    int b = INFL_D_OK;
    if ((flags & INFL_RS_INBLOCK) && tables_bit == block_bit + 1u) {   // the tables the last call left
      uint4 *lv = (uint4 *)&L;
      for (uint32_t k = (uint32_t)lig; k < TABLE_VEC; k += T) lv[k] = tables[k];
      group_sync();
    } else {
      if (btype == 1) {
        const int f = setup_fixed<T, F>(L, lig, 0);
        b = f == BUILD_OK ? INFL_D_OK : INFL_D_INTERNAL;
      } else {
        b = setup_dynamic<T, F>(r, L, lig, 0);
      }
      tables_fresh = b == INFL_D_OK;
    }
    if (b != INFL_D_OK) {
      rc = b == INFL_D_NOFIT ? INFL_D_INTERNAL : b;
      break;
    }
    if (flags & INFL_RS_INBLOCK) seek(bit);          // go on at the symbol the last call stopped at
    flags |= INFL_RS_INBLOCK;
    uint64_t symbit = rd_bitpos(r, src);
    rc = huffman_block<T, F>(r, dst, pos, cap, L, lig, fuel, &symbit, src);
    if (rc != INFL_D_OK) {
      bit = symbit;                                  // (block_bit stays)
      break;
    }
    bit = block_bit = rd_bitpos(r, src);
    flags &= ~INFL_RS_INBLOCK;
    tables_fresh = false;
    if (bfinal) flags |= INFL_RS_DONE;
  }
  if (rc != INFL_D_OK && !(flags & INFL_RS_INBLOCK)) bit = block_bit;
  group_sync();
  if (tables && (flags & INFL_RS_INBLOCK) && tables_fresh) {   // the block goes on in the next call: keep its tables
    const uint4 *lv = (const uint4 *)&L;
    for (uint32_t k = (uint32_t)lig; k < TABLE_VEC; k += T) tables[k] = lv[k];
    tables_bit = block_bit + 1u;
  }
  if (hout) {
    const uint32_t fresh_out = pos - pos_before < hout_cap ? pos - pos_before : hout_cap;
    for (uint32_t k = (uint32_t)lig; k < fresh_out; k += T) hout[k] = dst[pos_before + k];
  }
  if (lig == 0) {
    InflResume ns;
    ns.bit       = bit;
    ns.block_bit = block_bit;
    ns.pos       = pos;
    ns.flags     = flags;
    ns.detail    = rc;
    ns.reserved  = pos_before;                       // where this call's output begins
    ns.tables_bit = tables_bit;
    *state = ns;
    if (hstate) *hstate = ns;
  }
}

// ---------------------------------------------------------------------------
// Opt-in integrity and framing.  The reference never looks at what follows the final block
// (src/infl/zlib.h:45-96 reads the two header bytes only; no Adler-32 anywhere), so none of this
// runs behind its ten functions: it belongs to entry points of its own (infl_batch_dev_checked).
//   zlib (RFC 1950)  the Adler-32 of the output against the 4 bytes behind the final block;
//   gzip (RFC 1952)  the member header is skipped by a pass of its own before the decode, then
//                    CRC-32 and ISIZE (output length modulo 2^32) against the 8-byte trailer.
// One wavefront per stream reads the stream's output once more, 16 bytes per lane and access.
// Both sums are linear in the data, so every lane sums its own 16-byte chunks (one every 1024
// bytes) and the 64 partial results are combined at the end.
// ---------------------------------------------------------------------------
constexpr uint32_t CRC_POLY = 0xEDB88320u;           // reflected CRC-32 polynomial: bit 31 is the x^0 term

// product of two polynomials modulo the CRC polynomial (reflected representation)
__device__ constexpr uint32_t crc_mul(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int i = 0; i < 32; i++) {
    if (a & (0x80000000u >> i)) p ^= b;
    b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
  }
  return p;
}

struct CrcTables {
  uint32_t t[4][256];      // slicing-by-4 tables of the byte-wise update
  uint32_t adv[4][256];    // a register moved over 1008 zero bytes (from one 16-byte chunk of a lane to its next)
  uint32_t x2n[32];        // x^(2^k) modulo the polynomial
};
constexpr CrcTables make_crc_tables() {
  CrcTables T{};
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
    T.t[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; i++)
    for (int j = 1; j < 4; j++) T.t[j][i] = (T.t[j - 1][i] >> 8) ^ T.t[0][T.t[j - 1][i] & 0xffu];
  T.x2n[0] = 0x40000000u;                              // x^1
  for (int k = 1; k < 32; k++) T.x2n[k] = crc_mul(T.x2n[k - 1], T.x2n[k - 1]);
  uint32_t xk = 0x80000000u;                           // x^(8 * 1008)
  for (uint32_t n = 1008u, k = 3; n; n >>= 1, k++)
    if (n & 1u) xk = crc_mul(T.x2n[k & 31u], xk);
  for (int j = 0; j < 4; j++)
    for (uint32_t b = 0; b < 256; b++) T.adv[j][b] = crc_mul(xk, b << (8 * j));
  return T;
}
__constant__ const CrcTables c_crc = make_crc_tables();

// x^(8 * nbytes) modulo the polynomial
__device__ __forceinline__ uint32_t crc_xpow_bytes(uint64_t nbytes, const uint32_t *x2n) {
  uint32_t p = 0x80000000u;
  uint32_t k = 3;
  for (uint64_t n = nbytes; n; n >>= 1, k++)
    if (n & 1u) p = crc_mul(x2n[k & 31u], p);
  return p;
}

__device__ __forceinline__ uint32_t wave_xor(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) v ^= (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63u) ^ (uint32_t)d) << 2), (int)v);
  return v;
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int      from = (int)(((threadIdx.x & 63u) ^ (uint32_t)d) << 2);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)(uint32_t)(v >> 32));
    v += ((uint64_t)hi << 32) | lo;
  }
  return v;
}

// The lane's up to 16 bytes of the 1024-byte block that begins at p[base]: bytes [base + 16 * lane, ...) below n,
// zero filled.  Returns how many of them are real.
__device__ __forceinline__ uint32_t check_chunk(const uint8_t *p, uint64_t base, uint64_t n, int lane, uint32_t w[4]) {
  const uint64_t at = base + 16u * (uint32_t)lane;
  w[0] = w[1] = w[2] = w[3] = 0;
  if (at + 16u <= n) {
    uint4 q;
    __builtin_memcpy(&q, p + at, 16);
    w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
    return 16u;
  }
  if (at >= n) return 0u;
  const uint32_t k = (uint32_t)(n - at);
  for (uint32_t j = 0; j < k; j++) w[j >> 2] |= (uint32_t)p[at + j] << (8u * (j & 3u));
  return k;
}

// Adler-32 (RFC 1950) of p[0 .. n) by one wavefront.  a = 1 + sum of the bytes, b = n + sum of
// (n - i) * byte i, both modulo 65521: every lane adds up its chunks with the weights reduced
// modulo 65521 (64-bit sums, reduced once at the end).
__device__ __forceinline__ uint32_t wave_adler32(const uint8_t *p, uint64_t n, int lane) {
  constexpr uint32_t MOD = 65521u;
  uint64_t A = 0, B = 0;
  for (uint64_t base = 0; base < n; base += 1024u) {
    uint32_t w[4];
    if (check_chunk(p, base, n, lane, w) == 0) continue;
    const uint64_t i0 = base + 16u * (uint32_t)lane;
    const uint32_t w0 = (uint32_t)((n - i0) % MOD);
    uint32_t s1 = 0, s2 = 0;                           // sum of the bytes, sum of j * byte j
#pragma unroll
    for (int q = 0; q < 4; q++) {
      s1 = __builtin_amdgcn_udot4(w[q], 0x01010101u, s1, false);
      s2 = __builtin_amdgcn_udot4(w[q], 0x03020100u + 0x04040404u * (uint32_t)q, s2, false);
    }
    A += s1;
    B += (uint64_t)w0 * s1 + (MOD - s2 % MOD);         // sum of (w0 - j) * byte j, kept non-negative
  }
  A = wave_sum64(A);
  B = wave_sum64(B);
  const uint32_t a = (uint32_t)((1u + A) % MOD), b = (uint32_t)((n % MOD + B) % MOD);
  return (b << 16) | a;
}

// CRC-32 (RFC 1952) of p[0 .. n) by one wavefront; T = the tables in LDS.
__device__ __forceinline__ uint32_t wave_crc32(const uint8_t *p, uint64_t n, int lane, const CrcTables &T) {
  auto step4 = [&](uint32_t c, uint32_t word) {
    c ^= word;
    return T.t[3][c & 0xffu] ^ T.t[2][(c >> 8) & 0xffu] ^ T.t[1][(c >> 16) & 0xffu] ^ T.t[0][c >> 24];
  };
  const uint64_t full = n & ~(uint64_t)1023u;          // whole 1024-byte blocks: 16 bytes per lane each
  uint32_t acc = 0;
  for (uint64_t base = 0; base < full; base += 1024u) {
    uint4 q;
    __builtin_memcpy(&q, p + base + 16u * (uint32_t)lane, 16);
    acc = T.adv[0][acc & 0xffu] ^ T.adv[1][(acc >> 8) & 0xffu] ^ T.adv[2][(acc >> 16) & 0xffu] ^ T.adv[3][acc >> 24];
    acc = step4(step4(step4(step4(acc, q.x), q.y), q.z), q.w);
  }
  // behind a lane's last chunk of the whole blocks lie the chunks of the lanes above it, then the tail
  const uint32_t tail = (uint32_t)(n - full);
  uint32_t total = crc_mul(crc_xpow_bytes(16u * (63u - (uint32_t)lane) + tail, T.x2n), acc);
  // the tail (under 1024 bytes): again up to 16 bytes per lane, byte by byte
  {
    uint32_t w[4];
    const uint32_t k = check_chunk(p, full, n, lane, w);
    uint32_t c = 0;
    for (uint32_t j = 0; j < k; j++) c = T.t[0][(c ^ (w[j >> 2] >> (8u * (j & 3u)))) & 0xffu] ^ (c >> 8);
    if (k) total ^= crc_mul(crc_xpow_bytes(tail - (16u * (uint32_t)lane + k), T.x2n), c);
  }
  total = wave_xor(total);
  // the register starts at all ones (moved over the n bytes) and is inverted at the end
  return total ^ crc_mul(crc_xpow_bytes(n, T.x2n), 0xffffffffu) ^ 0xffffffffu;
}

// gzip member header (RFC 1952 2.3): magic, CM = 8, flags, then the optional fields.  One lane per
// stream.  skip[i] = bytes in front of the DEFLATE data, or INFL_SKIP_BAD / INFL_SKIP_SHORT.
__global__ __launch_bounds__(256) void infl_gzip_header_kernel(const uint8_t *__restrict__ src,
                                                              const uint64_t *__restrict__ src_off, uint64_t n,
                                                              uint32_t *__restrict__ skip) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint8_t *p = src + src_off[i];
  const uint64_t len = src_off[i + 1] - src_off[i];
  uint32_t res = INFL_SKIP_SHORT;
  do {
    if (len < 10) {
      if (len >= 1 && p[0] != 0x1f) res = INFL_SKIP_BAD;
      if (len >= 2 && p[1] != 0x8b) res = INFL_SKIP_BAD;
      if (len >= 3 && p[2] != 8) res = INFL_SKIP_BAD;
      break;
    }
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0u)) {   // reserved flag bits must be zero
      res = INFL_SKIP_BAD;
      break;
    }
    const uint32_t flg = p[3];
    uint64_t at = 10;
    if (flg & 4u) {                                  // FEXTRA: two length bytes, then that many
      if (at + 2 > len) break;
      at += 2u + (p[at] | ((uint32_t)p[at + 1] << 8));
      if (at > len) break;
    }
    bool shortfall = false;
    for (uint32_t f = 8u; f <= 16u && !shortfall; f <<= 1)   // FNAME, FCOMMENT: zero-terminated
      if (flg & f) {
        while (at < len && p[at]) at++;
        if (at >= len) shortfall = true;
        at++;
      }
    if (shortfall) break;
    if (flg & 2u) at += 2;                           // FHCRC (not verified)
    if (at > len || at > 0xfffffff0u) break;
    res = (uint32_t)at;
  } while (false);
  skip[i] = res;
}

// Checks every stream that decoded: what = INFL_CHECK_ADLER32 or INFL_CHECK_GZIP.  inpos[i] = input bytes the
// decode consumed (from the DEFLATE data's first byte on: the trailer begins there); skip as above (gzip only).
template <int WHAT>
__global__ __launch_bounds__(256) void infl_verify_kernel(const uint8_t *__restrict__ src, const uint64_t *__restrict__ src_off,
                                                         const uint8_t *__restrict__ dst, const uint64_t *__restrict__ dst_off,
                                                         uint64_t n, int32_t *__restrict__ status, uint32_t *__restrict__ outlen,
                                                         int32_t *__restrict__ detail, const uint32_t *__restrict__ inpos,
                                                         const uint32_t *__restrict__ skip) {
  __shared__ CrcTables T;
  if (WHAT == INFL_CHECK_GZIP) {
    const uint32_t *from = (const uint32_t *)&c_crc;
    for (uint32_t k = threadIdx.x; k < sizeof(CrcTables) / 4u; k += 256u) ((uint32_t *)&T)[k] = from[k];
    __syncthreads();
  }
  const int      lane = threadIdx.x & 63;
  const uint64_t i    = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
  if (i >= n) return;
  const uint32_t hdr = WHAT == INFL_CHECK_GZIP ? skip[i] : 0u;
  int32_t st = status[i], dt = st == 0 ? INFL_D_OK : detail ? detail[i] : INFL_D_INVALID;
  uint32_t ol = outlen[i];
  if (WHAT == INFL_CHECK_GZIP && hdr >= INFL_SKIP_SHORT) {                 // no usable header: nothing was decoded
    st = -1;
    dt = hdr == INFL_SKIP_BAD ? INFL_D_INVALID : INFL_D_NEED_INPUT;
    ol = 0;
  } else if (st == 0) {
    const uint64_t s0 = src_off[i], len = src_off[i + 1] - s0, at = (uint64_t)hdr + inpos[i];
    const uint32_t need = WHAT == INFL_CHECK_GZIP ? 8u : 4u;
    const uint8_t *out = dst + dst_off[i];
    if (at + need > len) {                           // the trailer is not all there
      st = -1;
      dt = INFL_D_NEED_INPUT;
    } else {
      const uint8_t *t = src + s0 + at;
      bool good;
      if (WHAT == INFL_CHECK_GZIP) {
        const uint32_t want = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        const uint32_t isize = t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        good = wave_crc32(out, ol, lane, T) == want && isize == ol;
      } else {
        const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
        good = wave_adler32(out, ol, lane) == want;
      }
      if (!good) {
        st = -1;
        dt = INFL_D_CHECKSUM;
      }
    }
  }
  if (lane == 0) {
    status[i] = st;
    outlen[i] = ol;
    if (detail) detail[i] = dt;
  }
}

// ---------------------------------------------------------------------------------------------
// PNG un-filter: the step that follows inflate in a PNG decoder (the reference names it as its
// caller's job, README.md:150-152; the rule itself is the PNG specification's, section 9).  Every
// scanline arrives as one filter-type byte and rowbytes filtered bytes; a byte is rebuilt from the
// byte bpp to its left (a), the byte above (b) and the byte above a (c).
//
// One wavefront per image, one lane per scanline, 64 scanlines in flight as a diagonal: at step t
// lane l rebuilds byte x = t - l of its row, so the byte above it left lane l-1 one step earlier
// (one DPP shift brings it over) and a and c are values the lane itself had bpp steps earlier.
// Memory moves in windows of PNG_WIN steps through an LDS tile, in pieces of 16 bytes: PNG_WIN / 16
// lanes cover one scanline's window, so one load instruction brings the windows of several
// scanlines, coalesced (a lane reading its own row from memory instead touches 64 cache lines per
// instruction, and the lines are evicted between touches: 3 x the traffic, measured).  The lanes then
// run down their tile rows 16 bytes at a time, write the rebuilt bytes over the filtered ones, and the
// tile goes back to memory the way it came.  While a window is being rebuilt the next one's bytes are
// already on their way from memory into registers.  Rows beyond the first 64 follow in bands; lane 0
// of a later band gets the last row of the band before from the output (tile row 64).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t PNG_WIN    = 256;                  // steps (= bytes of every scanline) per window
constexpr uint32_t PNG_PIECES = PNG_WIN / 16;         // lanes that cover one scanline's window
constexpr uint32_t PNG_RPI    = 64 / PNG_PIECES;      // scanlines per load / store instruction
constexpr uint32_t PNG_NI     = 64 / PNG_RPI;         // such instructions per window
constexpr uint32_t PNG_PITCH  = PNG_WIN + 16;         // tile row pitch in bytes: 16-byte accesses of 16 lanes in a row or in a
                                                      // column spread over all banks (pitch / 4 = 4 mod 8)
static_assert(PNG_WIN == 256 || PNG_WIN == 128, "pitch and bank layout are worked out for these");

struct __attribute__((packed)) PngPiece { uint32_t w[4]; };          // 16 bytes at any address
struct __attribute__((aligned(16))) PngPiece16 { uint32_t w[4]; };   // in the tile

// 16 bytes at p, any alignment; what lies outside [lo, hi) reads as 0.
__device__ __forceinline__ PngPiece16 png_load16_guarded(const uint8_t *p, const uint8_t *lo, const uint8_t *hi) {
  PngPiece16 c = {{0, 0, 0, 0}};
#pragma unroll
  for (int j = 0; j < 16; j++)
    if (p + j >= lo && p + j < hi) c.w[j >> 2] |= (uint32_t)p[j] << (8 * (j & 3));
  return c;
}

__device__ __forceinline__ uint32_t png_piece_byte(const PngPiece &p, uint32_t j) {   // byte j of 16, no indexed registers
  const uint32_t w = j < 8 ? (j < 4 ? p.w[0] : p.w[1]) : (j < 12 ? p.w[2] : p.w[3]);
  return (w >> (8u * (j & 3u))) & 0xffu;
}

// v of lane l-1; lane 0 keeps `first`
__device__ __forceinline__ uint32_t from_lane_below(uint32_t first, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// The last bytes of a lane's own row (a comes from there) and of the row above it (c): BPP known at compile time =
// the values of the last BPP steps by name, no shifting; BPP = 0 = any bpp of 1..8 at run time, two 8-byte shift
// registers.
template <int BPP>
struct PngPast {
  uint32_t own[BPP], above[BPP];
  __device__ __forceinline__ void clear(uint32_t) {
#pragma unroll
    for (int j = 0; j < BPP; j++) own[j] = above[j] = 0;
  }
};
template <>
struct PngPast<0> {
  uint64_t own, above;
  uint32_t sh;
  __device__ __forceinline__ void clear(uint32_t bpp) {
    own = above = 0;
    sh  = 8u * (bpp - 1u);
  }
};

// One image (geometry checked by the caller: rows >= 1, rb >= 1, need_in < 2^32 - 1024); true if a scanline names a filter
// type above 4.  The windows of all bands form one sequence.
template <int BPP>
__device__ __forceinline__ bool png_unfilter_image(uint8_t *tile, const uint8_t *ib, uint32_t need_in, uint8_t *ob,
                                                   uint32_t rows, uint32_t rb, uint32_t bpp, uint32_t lane) {
  const uint32_t pitch = rb + 1u;
  const uint32_t steps = rb + 63u;                   // until lane 63 has finished its row
  const bool     fast  = rb >= 16;
  const uint32_t sub = lane / PNG_PIECES, piece = lane % PNG_PIECES;   // this lane's part in a load / store instruction
  bool bad = false;
  // Tile row q, byte j = byte x = t0 - q + j of scanline r0 + q (x wraps to a large number left of the row; the
  // address stays inside the image all the same: scanline r0 + q starts at least q + 1 bytes into it).  No branch
  // per load, so that all of them are in flight together: a piece that would reach past the image is read from the
  // image's last 16 bytes instead.  With rows of 16 bytes or more the only bytes of such a piece that anyone uses
  // are the last 15 of the last row: those are put right from `tail` once the tile is written.
  PngPiece nxt[PNG_NI], nxt_up = {{0, 0, 0, 0}}, tail = {{0, 0, 0, 0}}, up_tail = {{0, 0, 0, 0}};
  uint32_t nxt_ft;
  if (fast) tail = *(const PngPiece *)(ib + (need_in - 16u));
#define PNG_FETCH(R0, T0)                                                                            \
  if (fast) {                                                                                        \
    const uint32_t nr_ = rows - (R0) < 64u ? rows - (R0) : 64u, xl_ = (T0) + 16u * piece + 1u;        \
    _Pragma("unroll") for (uint32_t u_ = 0; u_ < PNG_NI; u_++) {                                     \
      const uint32_t q_ = u_ * PNG_RPI + sub, qq_ = q_ < nr_ ? q_ : nr_ - 1u;                        \
      const uint32_t at_ = ((R0) + qq_) * pitch - qq_ + xl_;                                         \
      nxt[u_] = *(const PngPiece *)(ib + (at_ < need_in - 16u ? at_ : need_in - 16u));               \
    }                                                                                                \
  }
  PNG_FETCH(0u, 0u)
  nxt_ft = lane < rows ? ib[lane * pitch] : 0u;
  uint32_t r0 = 0, t0 = 0, ft = 0, m_sub = 0, m_up = 0, m_avg = 0, m_paeth = 0, prev = 0;
  PngPast<BPP> past;
  past.clear(bpp);
  for (;;) {
    const uint32_t nrow = rows - r0 < 64u ? rows - r0 : 64u;
    const uint8_t *up = r0 ? ob + (uint64_t)(r0 - 1u) * rb : nullptr;   // the row above lane 0's
    if (t0 == 0) {                                   // a new band
      ft = nxt_ft;
      if (ft > 4) {
        bad = true;
        ft  = 0;
      }
      // the predictor is picked with masks, not branches: the lanes of a wavefront rarely agree on the filter type
      m_sub = ft == 1 ? 0xffu : 0u, m_up = ft == 2 ? 0xffu : 0u, m_avg = ft == 3 ? 0xffu : 0u, m_paeth = ft == 4 ? 0xffu : 0u;
      past.clear(bpp);
      prev = 0;
      // the band before has just been written: its last row is this band's row above
      if (up && fast && lane < 16) {
        const uint32_t at = 16u * lane;
        if (lane < PNG_PIECES) nxt_up = *(const PngPiece *)(up + (at < rb - 16u ? at : rb - 16u));
        up_tail = *(const PngPiece *)(up + (rb - 16u));
      }
    }
    if (fast) {
#pragma unroll
      for (uint32_t u = 0; u < PNG_NI; u++) {
        PngPiece16 v = {{nxt[u].w[0], nxt[u].w[1], nxt[u].w[2], nxt[u].w[3]}};
        *(PngPiece16 *)(tile + (u * PNG_RPI + sub) * PNG_PITCH + 16u * piece) = v;
      }
      if (lane < PNG_PIECES) {
        PngPiece16 v = {{nxt_up.w[0], nxt_up.w[1], nxt_up.w[2], nxt_up.w[3]}};
        *(PngPiece16 *)(tile + 64u * PNG_PITCH + 16u * lane) = v;
      }
    } else {                                         // rows of fewer than 16 bytes: no hurry
      for (uint32_t q = sub; q < nrow; q += PNG_RPI)
        *(PngPiece16 *)(tile + q * PNG_PITCH + 16u * piece) =
            png_load16_guarded(ib + (r0 + q) * pitch + 1u + (int64_t)(int32_t)(t0 + 16u * piece - q), ib, ib + need_in);
      if (lane < PNG_PIECES) {
        PngPiece16 v = {{0, 0, 0, 0}};
        if (up) v = png_load16_guarded(up + t0 + 16u * lane, up, up + rb);
        *(PngPiece16 *)(tile + 64u * PNG_PITCH + 16u * lane) = v;
      }
    }
    group_sync();
    if (fast && lane < 15) {                         // the last 15 bytes of the image and of the row above
      const uint32_t x = rb - 1u - lane;
      if (r0 + nrow == rows) {
        const uint32_t off = x - (t0 - (nrow - 1u));
        if (off < PNG_WIN) tile[(nrow - 1u) * PNG_PITCH + off] = (uint8_t)png_piece_byte(tail, 15u - lane);
      }
      if (up && x - t0 < PNG_WIN)
        tile[64u * PNG_PITCH + (x - t0)] = (uint8_t)png_piece_byte(up_tail, 15u - lane);
    }
    group_sync();
    // start on the next window's bytes
    const bool     more_here = t0 + PNG_WIN < steps, more = more_here || r0 + 64u < rows;
    const uint32_t r0n = more_here ? r0 : r0 + 64u, t0n = more_here ? t0 + PNG_WIN : 0u;
    if (more) {
      PNG_FETCH(r0n, t0n)
      if (more_here) {
        if (up && fast && lane < PNG_PIECES) {
          const uint32_t at = t0n + 16u * lane;
          nxt_up = *(const PngPiece *)(up + (at < rb - 16u ? at : rb - 16u));
        }
      } else {
        nxt_ft = r0n + lane < rows ? ib[(r0n + lane) * pitch] : 0u;
      }
    }
    if (t0 == 0) {
      // What lies left of a row (steps before the lane's first) must rebuild to 0, so that a and c of the first
      // pixel are 0: zero it here (0 filtered + predictor 0 = 0) instead of testing the position at every step.
      uint32_t *mine = (uint32_t *)(tile + lane * PNG_PITCH);
#pragma unroll
      for (uint32_t w = 0; w < 16; w++) {
        const uint32_t nz = lane > 4u * w ? lane - 4u * w : 0u;         // bytes of this dword left of the row
        if (nz) mine[w] = nz >= 4u ? 0u : mine[w] & (~0u << (8u * nz));
      }
    }
    group_sync();
    const uint32_t nchunk = steps - t0 < PNG_WIN ? (steps - t0 + 15u) / 16u : PNG_WIN / 16u;
    for (uint32_t k = 0; k < nchunk; k++) {
      const PngPiece16 cur = *(const PngPiece16 *)(tile + lane * PNG_PITCH + 16u * k),
                       upc = *(const PngPiece16 *)(tile + 64u * PNG_PITCH + 16u * k);
      PngPiece16 o = {{0, 0, 0, 0}};
      uint32_t byte[16], bu[16];
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const uint32_t v = (cur.w[j >> 2] >> (8 * (j & 3))) & 0xffu;
        bu[j] = from_lane_below((upc.w[j >> 2] >> (8 * (j & 3))) & 0xffu, prev);
        uint32_t a, c;
        if constexpr (BPP != 0) {
          a = j >= BPP ? byte[j >= BPP ? j - BPP : 0] : past.own[j < BPP ? j : 0];
          c = j >= BPP ? bu[j >= BPP ? j - BPP : 0] : past.above[j < BPP ? j : 0];
        } else {
          a = (uint32_t)(past.own >> past.sh) & 0xffu;
          c = (uint32_t)(past.above >> past.sh) & 0xffu;
        }
        const uint32_t b = bu[j], ab = a + b;
        const uint32_t pa = __builtin_amdgcn_sad_u8(b, c, 0), pb = __builtin_amdgcn_sad_u8(a, c, 0),
                       pc = __builtin_amdgcn_sad_u16(ab, c << 1, 0);            // |a + b - 2c|, both sides below 2^16
        const uint32_t paeth = pa <= (pb < pc ? pb : pc) ? a : pb <= pc ? b : c;
        const uint32_t pred = (a & m_sub) | (b & m_up) | ((ab >> 1) & m_avg) | (paeth & m_paeth);
        byte[j] = (v + pred) & 0xffu;
        prev = byte[j];
        if constexpr (BPP == 0) {
          past.own   = (past.own << 8) | byte[j];
          past.above = (past.above << 8) | bu[j];
        }
        o.w[j >> 2] |= byte[j] << (8 * (j & 3));
      }
      if constexpr (BPP != 0) {
#pragma unroll
        for (int j = 0; j < BPP; j++) {
          past.own[j]   = byte[16 - BPP + j];
          past.above[j] = bu[16 - BPP + j];
        }
      }
      *(PngPiece16 *)(tile + lane * PNG_PITCH + 16u * k) = o;
    }
    group_sync();
    const bool inner = t0 + 1u >= nrow && t0 + PNG_WIN <= rb;     // every row's window lies inside the row
#pragma unroll
    for (uint32_t u = 0; u < PNG_NI; u++) {
      const uint32_t q = u * PNG_RPI + sub, x = t0 + 16u * piece - q;
      if (q < nrow) {
        const PngPiece16 v = *(const PngPiece16 *)(tile + q * PNG_PITCH + 16u * piece);
        uint8_t *d = ob + (uint64_t)(r0 + q) * rb + (int64_t)(int32_t)x;
        if (inner || (x < rb && rb - x >= 16u)) {
          PngPiece o = {{v.w[0], v.w[1], v.w[2], v.w[3]}};
          *(PngPiece *)d = o;
        }
      }
    }
    if (!inner && lane < nrow) {
      // The bytes of pieces that straddle an end of their row: at most the first 15 and the last 15 bytes of a row;
      // here every lane looks after its own row.
      const uint32_t ws = t0 - lane;                                 // x of the tile row's first byte
      for (uint32_t e = 0; e < 30; e++) {
        const uint32_t x = e < 15 ? e : rb - 1u - (e - 15u), off = x - ws;
        if (x < rb && off < PNG_WIN) {
          const uint32_t ps = ws + (off & ~15u);                     // x of the piece's first byte
          if (!(ps < rb && rb - ps >= 16u)) ob[(uint64_t)(r0 + lane) * rb + x] = tile[lane * PNG_PITCH + off];
        }
      }
    }
    group_sync();
    if (!more) break;
    // The next band's lane 0 reads this band's last row back from memory.  Writer and reader are this wavefront:
    // its stores have to be complete, no cache has to be written back or invalidated.
    if (!more_here) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    r0 = r0n;
    t0 = t0n;
  }
#undef PNG_FETCH
  return bad;
}

__device__ __forceinline__ void png_unfilter_one(uint8_t *tile, uint64_t i, uint32_t lane, const uint8_t *in,
                                                 const uint64_t *in_off, const uint32_t *in_len, const int32_t *in_status,
                                                 uint8_t *out, const uint64_t *out_off, const InflPngGeom *geom,
                                                 int32_t *status, uint64_t in_bytes, uint64_t out_bytes) {
  const InflPngGeom g = geom[i];
  const uint64_t i0 = in_off[i], i1 = in_off[i + 1], o0 = out_off[i], o1 = out_off[i + 1];
  uint64_t have = i1 >= i0 ? i1 - i0 : 0;
  if (in_len && in_len[i] < have) have = in_len[i];
  const uint32_t rows = g.rows, rb = g.rowbytes;
  const uint64_t need_in = (uint64_t)rows * ((uint64_t)rb + 1u), need_out = (uint64_t)rows * rb;
  const bool fits = g.bpp >= 1 && g.bpp <= 8 && need_in < 0xfffffc00ull && need_in <= have && o1 >= o0 &&
                    need_out <= o1 - o0 && !(in_status && in_status[i] != 0) &&
                    i1 <= in_bytes && o1 <= out_bytes;   // (the arenas' sizes where the caller gives them, else ~0)
  if (!fits) {
    if (lane == 0) status[i] = -1;
    return;
  }
  const uint8_t *ib = in + i0;
  uint8_t       *ob = out + o0;
  bool bad = false;
  if (rows == 0) {                                   // nothing to rebuild
  } else if (rb == 0) {                              // filter bytes only
    for (uint32_t r = lane; r < rows; r += 64) bad |= ib[r] > 4;
  } else {
    switch (g.bpp) {                                 // the pixel sizes a PNG can have, each with its own code
      case 1: bad = png_unfilter_image<1>(tile, ib, (uint32_t)need_in, ob, rows, rb, 1, lane); break;
      case 2: bad = png_unfilter_image<2>(tile, ib, (uint32_t)need_in, ob, rows, rb, 2, lane); break;
      case 3: bad = png_unfilter_image<3>(tile, ib, (uint32_t)need_in, ob, rows, rb, 3, lane); break;
      case 4: bad = png_unfilter_image<4>(tile, ib, (uint32_t)need_in, ob, rows, rb, 4, lane); break;
      case 6: bad = png_unfilter_image<6>(tile, ib, (uint32_t)need_in, ob, rows, rb, 6, lane); break;
      case 8: bad = png_unfilter_image<8>(tile, ib, (uint32_t)need_in, ob, rows, rb, 8, lane); break;
      default: bad = png_unfilter_image<0>(tile, ib, (uint32_t)need_in, ob, rows, rb, g.bpp, lane); break;
    }
  }
  const bool any_bad = __ballot(bad) != 0;
  if (lane == 0) status[i] = any_bad ? -1 : 0;
}

// One wavefront per workgroup; a workgroup takes images blockIdx.x, blockIdx.x + gridDim.x, ... (the launcher sizes the
// grid to what the device holds at once: thousands of one-wavefront workgroups that live a few tens of microseconds
// each are started more slowly than they finish).
__global__ __launch_bounds__(64) void infl_png_unfilter_kernel(const uint8_t *in, const uint64_t *in_off,
                                                                const uint32_t *in_len, const int32_t *in_status,
                                                                uint8_t *out, const uint64_t *out_off,
                                                                const InflPngGeom *geom, uint64_t n, int32_t *status,
                                                                uint64_t in_bytes, uint64_t out_bytes) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[65 * PNG_PITCH];
  for (uint64_t i = blockIdx.x; i < n; i += gridDim.x) {
    png_unfilter_one(tile, i, threadIdx.x, in, in_off, in_len, in_status, out, out_off, geom, status, in_bytes, out_bytes);
    group_sync();
  }
}

// ---------------------------------------------------------------------------------------------
// Gather of input pieces that already lie in device memory (the device-side counterpart of a run of
// infl_include() calls, reference src/infl/mem.c:114-216: every call appends one byte range to the stream's
// logical input).  Piece p is blob[piece_off[p] .. + piece_len[p]) and goes to out[piece_dst[p] ..): the
// caller lays the pieces of a stream one behind the other and the streams into one arena.  One wavefront
// per piece; a lane moves 16 bytes per access at whatever alignment source and target have (neighbouring
// lanes, neighbouring bytes), the last bytes of a piece go one at a time.  blob_bytes / out_bytes are the
// sizes of the blob and of the arena where the caller gives them (~0 = not known): a piece that does not lie
// inside both is not touched and gets piece_status -1 (piece_status may be null).
// ---------------------------------------------------------------------------------------------
struct __attribute__((packed)) Bytes16 { uint32_t w[4]; };          // 16 bytes at any address
__global__ __launch_bounds__(64) void infl_gather_kernel(const uint8_t *__restrict__ blob,
                                                         const uint64_t *__restrict__ piece_off,
                                                         const uint32_t *__restrict__ piece_len,
                                                         const uint64_t *__restrict__ piece_dst, uint64_t n_pieces,
                                                         uint8_t *__restrict__ out, uint64_t blob_bytes,
                                                         uint64_t out_bytes, int32_t *__restrict__ piece_status) {
  const uint32_t lane = threadIdx.x;
  for (uint64_t p = blockIdx.x; p < n_pieces; p += gridDim.x) {
    const uint64_t so = piece_off[p], to = piece_dst[p];
    const uint32_t len = piece_len[p], body = len & ~15u;
    const bool     inside = so <= blob_bytes && len <= blob_bytes - so && to <= out_bytes && len <= out_bytes - to;
    if (piece_status && lane == 0) piece_status[p] = inside ? 0 : -1;
    if (!inside) continue;
    const uint8_t *s   = blob + so;
    uint8_t       *d   = out + to;
    for (uint32_t k = 16u * lane; k < body; k += 1024u) *(Bytes16 *)(d + k) = *(const Bytes16 *)(s + k);
    if (lane < len - body) d[body + lane] = s[body + lane];
  }
}

// ---------------------------------------------------------------------------------------------
// Order of a batch by size class.  A batch whose neighbouring streams differ widely in size decodes a quarter
// slower than the same streams with similar sizes side by side (1 KiB .. 256 KiB log-uniform, 262 144 streams:
// 59.4 ms in no order, 43.1 ms counting-sorted into one class per doubling of a stream's compressed + decompressed
// bytes, 42.0 ms with eight classes per doubling, 41.7 ms fully sorted).  A counting sort on the device gives the
// parallel pass that order: classes by falling size (four per doubling), any order inside a class.  Results do not
// depend on the order streams are decoded in.  work: ORDER_CLASSES counters, then n indices.
// ---------------------------------------------------------------------------------------------
constexpr int ORDER_CLASSES = 256;
__device__ __forceinline__ uint32_t order_class(uint64_t cost) {
  if (cost < 4) return ORDER_CLASSES - 1u - (uint32_t)cost;
  const uint32_t lg = 63u - (uint32_t)__builtin_clzll(cost);             // 2 .. 63
  const uint32_t c  = 4u * lg + (uint32_t)((cost >> (lg - 2)) & 3u);     // four classes per doubling: 8 .. 255
  return ORDER_CLASSES - 1u - c;                                         // largest first
}
__global__ __launch_bounds__(256) void infl_order_clear_kernel(uint32_t *__restrict__ counters) {
  counters[threadIdx.x] = 0;
}
__global__ __launch_bounds__(256) void infl_order_count_kernel(const uint64_t *__restrict__ src_off,
                                                               const uint64_t *__restrict__ dst_off, uint64_t n,
                                                               uint32_t *__restrict__ counters) {
  __shared__ uint32_t local[ORDER_CLASSES];
  local[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i < n) atomicAdd(&local[order_class((src_off[i + 1] - src_off[i]) + (dst_off[i + 1] - dst_off[i]))], 1u);
  __syncthreads();
  if (local[threadIdx.x]) atomicAdd(&counters[threadIdx.x], local[threadIdx.x]);
}
__global__ __launch_bounds__(256) void infl_order_scan_kernel(uint32_t *__restrict__ counters) {   // counts -> first slots
  __shared__ uint32_t v[ORDER_CLASSES];
  v[threadIdx.x] = counters[threadIdx.x];
  __syncthreads();
  uint32_t before = 0;
  for (unsigned k = 0; k < threadIdx.x; k++) before += v[k];
  counters[threadIdx.x] = before;
}
__global__ __launch_bounds__(256) void infl_order_place_kernel(const uint64_t *__restrict__ src_off,
                                                               const uint64_t *__restrict__ dst_off, uint64_t n,
                                                               uint32_t *__restrict__ counters,
                                                               uint32_t *__restrict__ order) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i < n) {
    const uint32_t c = order_class((src_off[i + 1] - src_off[i]) + (dst_off[i + 1] - dst_off[i]));
    order[atomicAdd(&counters[c], 1u)] = (uint32_t)i;
  }
}

}  // namespace defl

#ifndef DEFL_LAUNCH_GROUP
#define DEFL_LAUNCH_GROUP(T)     // (the test suite's host-side emulation is told the lanes per stream of a launch here)
#endif

#ifndef DEFL_SINGLE_UNIT
extern "C" int infl_launch_parallel_small(const void *src, const uint64_t *src_off, void *dst, const uint64_t *dst_off,
                                          int flags, uint64_t n, int32_t *status, uint32_t *outlen, int32_t *detail,
                                          uint32_t *inpos, void *mirror, void *stream, const uint32_t *skip,
                                          const uint32_t *order);
#endif

// variant 0: reference-geometry tables only (one launch).
// variant 1: compact tables first, reference-geometry tables for what they defer (two launches).
// variant 2: wavefront-parallel block decode first (one wavefront per stream); what it defers
//            (anything malformed or out of space, tables too large for the compact format,
//            tiny streams) goes through the two serial passes of variant 1 (three launches).
// variant 3: diagnostic: the parallel pass alone; deferred streams keep INFL_STATUS_DEFERRED.
// variant 4: diagnostic: the parallel pass and the serial pass behind it; what they leave keeps its mark, and a stream
//            the windowed serial pass finished keeps INFL_STATUS_DEFERRED_DONE (only the last pass turns it into 0).
// variant 5: the two serial passes of variant 2 alone, for every stream (variant 1 with the second pass's own kernel
//            for 8 and 16 lanes per stream, which shares a window's marked streams out among its wavefronts).
extern "C" int infl_launch_batch(const void *src, const uint64_t *src_off, void *dst,
                                 const uint64_t *dst_off, int flags, uint64_t n, int32_t *status,
                                 uint32_t *outlen, int32_t *detail, uint32_t *inpos, int lanes,
                                 int variant, void *mirror, void *stream, const uint32_t *skip) {
  return infl_launch_batch_ordered(src, src_off, dst, dst_off, flags, n, status, outlen, detail, inpos, lanes, variant,
                                   mirror, stream, skip, nullptr);
}

// The same with `order` (may be null): the sequence in which the parallel pass's workgroups take the streams, made
// by infl_launch_order.  The serial passes behind it work on what that pass marked, in index order.
extern "C" int infl_launch_batch_ordered(const void *src, const uint64_t *src_off, void *dst,
                                         const uint64_t *dst_off, int flags, uint64_t n, int32_t *status,
                                         uint32_t *outlen, int32_t *detail, uint32_t *inpos, int lanes,
                                         int variant, void *mirror, void *stream, const uint32_t *skip,
                                         const uint32_t *order) {
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  auto s = (const uint8_t *)src;
  auto d = (uint8_t *)dst;
  auto m = (uint8_t *)mirror;
#define LAUNCH(T, FMT, MODE)                                                                     \
  do {                                                                                           \
    DEFL_LAUNCH_GROUP(T);                                                                        \
    hipLaunchKernelGGL((defl::infl_batch_kernel<T, defl::FMT, MODE>),                            \
                       dim3((unsigned)((n + (64 / T) - 1) / (64 / T))), dim3(64), 0, st, s, src_off, \
                       d, dst_off, flags, n, status, outlen, detail, inpos, m, skip,             \
                       (MODE) == 3 ? order : (const uint32_t *)nullptr);                         \
  } while (0)
#ifdef DEFL_SINGLE_UNIT            /* (the test suite's host-side emulation compiles everything as one unit) */
#define LAUNCH_SMALL()                                                                           \
  do {                                                                                           \
    DEFL_LAUNCH_GROUP(64);                                                                       \
    hipLaunchKernelGGL((defl::infl_batch_kernel<64, defl::CompactFmt, 6>), dim3((unsigned)n), dim3(64), 0, st, s, \
                       src_off, d, dst_off, flags, n, status, outlen, detail, inpos, m, skip, order); \
  } while (0)
#else
#define LAUNCH_SMALL()                                                                                          \
  do {                                                                                                          \
    if (infl_launch_parallel_small(src, src_off, dst, dst_off, flags, n, status, outlen, detail, inpos, mirror, \
                                   stream, skip, order) != 0)                                                   \
      return -1;                                                                                                \
  } while (0)
#endif
#define LAUNCH_DEFERRED_PLAIN(T) LAUNCH(T, CompactFmt, 4)
#define LAUNCH_DEFERRED_WINDOW(T)                                                                              \
  do {                                                                                                         \
    DEFL_LAUNCH_GROUP(T);                                                                                      \
    hipLaunchKernelGGL((defl::infl_deferred_window_kernel<T>),                                                 \
                       dim3((unsigned)((n + defl::kDeferWindow - 1) / defl::kDeferWindow * (defl::kDeferWindow / (64 / T)))), \
                     dim3(64), 0, st, s,                                                                     \
                       src_off, d, dst_off, flags, n, status, outlen, detail, inpos, m, skip);                 \
  } while (0)
#define LAUNCH_T(T, DEFERRED)   \
  if (variant == 0) {           \
    LAUNCH(T, RefFmt, 0);       \
  } else if (variant == 1) {    \
    LAUNCH(T, CompactFmt, 1);   \
    LAUNCH(T, RefFmt, 2);       \
  } else if (variant == 5) {    \
    DEFL_LAUNCH_GROUP(64);      \
    hipLaunchKernelGGL(defl::infl_mark_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, status, n); \
    DEFERRED(T);                \
    LAUNCH(T, RefFmt, 2);       \
  } else {                      \
    if (n >= defl::PAR_SHORT_BATCH) { \
      LAUNCH(64, CompactFmt, 3); \
    } else {                    \
      LAUNCH_SMALL();           \
    }                           \
    if (variant == 2 || variant == 4) { \
      DEFERRED(T);              \
    }                           \
    if (variant == 2) {         \
      if (detail) {             \
        LAUNCH(64, CompactFmt, 5); \
      }                         \
      LAUNCH(T, RefFmt, 2);     \
    }                           \
  }
  switch (lanes) {
    case 64: LAUNCH_T(64, LAUNCH_DEFERRED_PLAIN) break;
    case 32: LAUNCH_T(32, LAUNCH_DEFERRED_PLAIN) break;
    case 16: LAUNCH_T(16, LAUNCH_DEFERRED_WINDOW) break;
    case 8:  LAUNCH_T(8, LAUNCH_DEFERRED_WINDOW)  break;
    default: return -1;
  }
#undef LAUNCH_T
#undef LAUNCH_DEFERRED_WINDOW
#undef LAUNCH_SMALL
#undef LAUNCH_DEFERRED_PLAIN
#undef LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Enqueues the counting sort of infl_order_*: work = INFL_ORDER_WORK_BYTES(n) bytes of device memory; the order is
// left at (uint32_t *)work + 256.  n < 2^32.
extern "C" int infl_launch_order(const uint64_t *src_off, const uint64_t *dst_off, uint64_t n, void *work, void *stream) {
  if (n == 0) return 0;
  if (!work || n > 0xffffffffull) return -1;
  hipStream_t st = (hipStream_t)stream;
  uint32_t   *counters = (uint32_t *)work, *order = counters + defl::ORDER_CLASSES;
  const dim3  grid((unsigned)((n + 255) / 256));
  DEFL_LAUNCH_GROUP(64);
  hipLaunchKernelGGL(defl::infl_order_clear_kernel, dim3(1), dim3(256), 0, st, counters);
  hipLaunchKernelGGL(defl::infl_order_count_kernel, grid, dim3(256), 0, st, src_off, dst_off, n, counters);
  hipLaunchKernelGGL(defl::infl_order_scan_kernel, dim3(1), dim3(256), 0, st, counters);
  hipLaunchKernelGGL(defl::infl_order_place_kernel, grid, dim3(256), 0, st, src_off, dst_off, n, counters, order);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// The decode of infl_launch_batch followed by the integrity pass: check = INFL_CHECK_ADLER32 (zlib streams) or
// INFL_CHECK_GZIP (gzip members: a header pass first).  work = 8 * n bytes of device memory (consumed input
// bytes and header lengths per stream).
extern "C" int infl_launch_checked(const void *src, const uint64_t *src_off, void *dst, const uint64_t *dst_off,
                                   int check, uint64_t n, int32_t *status, uint32_t *outlen, int32_t *detail,
                                   void *work, int lanes, int variant, void *mirror, void *stream) {
  if (n == 0) return 0;
  if (!work || (check != INFL_CHECK_ADLER32 && check != INFL_CHECK_GZIP)) return -1;
  hipStream_t st = (hipStream_t)stream;
  uint32_t *inpos = (uint32_t *)work, *skip = inpos + n;
  if (check == INFL_CHECK_GZIP)
    hipLaunchKernelGGL(defl::infl_gzip_header_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       (const uint8_t *)src, src_off, n, skip);
  if (infl_launch_batch(src, src_off, dst, dst_off, check == INFL_CHECK_ADLER32 ? 1 : 0, n, status, outlen, detail,
                        inpos, lanes, variant, mirror, stream, check == INFL_CHECK_GZIP ? skip : nullptr) != 0)
    return -1;
  const dim3 grid((unsigned)((n + 3) / 4));
  if (check == INFL_CHECK_GZIP)
    hipLaunchKernelGGL((defl::infl_verify_kernel<INFL_CHECK_GZIP>), grid, dim3(256), 0, st, (const uint8_t *)src, src_off,
                       (const uint8_t *)dst, dst_off, n, status, outlen, detail, inpos, skip);
  else
    hipLaunchKernelGGL((defl::infl_verify_kernel<INFL_CHECK_ADLER32>), grid, dim3(256), 0, st, (const uint8_t *)src, src_off,
                       (const uint8_t *)dst, dst_off, n, status, outlen, detail, inpos, skip);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int infl_launch_resume(void *src, uint32_t srclen, void *dst, uint32_t cap, int zlib,
                                  InflResume *state, const void *fresh, uint32_t fresh_at, void *hout,
                                  uint32_t hout_cap, InflResume *hstate, void *tables, void *stream) {
  DEFL_LAUNCH_GROUP(64);
  hipLaunchKernelGGL((defl::infl_resume_kernel<64>), dim3(1), dim3(64), 0, (hipStream_t)stream, (uint8_t *)src,
                     srclen, (uint8_t *)dst, cap, zlib, state, (const uint8_t *)fresh, fresh_at, (uint8_t *)hout,
                     hout_cap, hstate, (uint4 *)tables);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int infl_launch_png_unfilter(const void *in, const uint64_t *in_off, const uint32_t *in_len,
                                        const int32_t *in_status, void *out, const uint64_t *out_off,
                                        const InflPngGeom *geom, uint64_t n, int32_t *status, void *stream,
                                        uint64_t in_bytes, uint64_t out_bytes) {
  if (n == 0) return 0;
  // as many workgroups as fit on the device at once (9 tiles in a compute unit's 160 KiB of LDS)
  static std::atomic<unsigned> resident[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  unsigned slots = resident[dev].load(std::memory_order_relaxed);
  if (!slots) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return -1;
    slots = (unsigned)cus * 9u;
    resident[dev].store(slots, std::memory_order_relaxed);
  }
  const unsigned grid = n < slots ? (unsigned)n : slots;
  hipLaunchKernelGGL(defl::infl_png_unfilter_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream,
                     (const uint8_t *)in, in_off, in_len, in_status, (uint8_t *)out, out_off, geom, n, status, in_bytes,
                     out_bytes);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int infl_launch_gather(const void *blob, const uint64_t *piece_off, const uint32_t *piece_len,
                                  const uint64_t *piece_dst, uint64_t n_pieces, void *out, void *stream,
                                  uint64_t blob_bytes, uint64_t out_bytes, int32_t *piece_status) {
  if (n_pieces == 0) return 0;
  // pieces are a few bytes to a few KiB each: a grid of at most 64 Ki workgroups that take the pieces in turn
  const unsigned grid = n_pieces < 65536u ? (unsigned)n_pieces : 65536u;
  DEFL_LAUNCH_GROUP(64);
  hipLaunchKernelGGL(defl::infl_gather_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, (const uint8_t *)blob,
                     piece_off, piece_len, piece_dst, n_pieces, (uint8_t *)out, blob_bytes, out_bytes, piece_status);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

#ifdef DEFL_PAR_PROFILE
// Diagnostic builds only: read (and clear) the phase counters of huffman_block_parallel.
extern "C" __attribute__((visibility("default"))) int infl_debug_par_profile(unsigned long long *out32) {
  unsigned long long zero[32] = {};
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(defl::g_par_prof), sizeof(zero)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(defl::g_par_prof), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

#endif  // DEFL_SMALL_BATCH_UNIT
