/*
 * infl_oracle.c -- CPU restatement of the one-shot inflate path of recp/defl.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP engine: it
 * is compiled and called by tests/, by __graft_entry__.smoke() and by the
 * cpu_baseline leg of bench.py, never by the shipped library.
 *
 * What it restates (all citations are into /root/reference):
 *   - the dispatcher              src/infl/infl.c:1507-1551
 *   - the stored-only fast path   src/infl/infl.c:94-267
 *   - the packed-table fast path  src/infl/infl.c:378-1104
 *   - RFC 1951 constant tables    src/infl/apicommon.h:36-64
 *   - zlib CMF/FLG validation     src/infl/infl.c:225-239, 1027-1039
 *
 * Third-party dependency: recp/huff (git submodule deps/huff, .gitmodules:1-3)
 * is NOT present in the reference tree and its pinned revision cannot be
 * recovered (the gitlink lives in the unmounted .git).  On this path the
 * reference takes one function from it, huff_init_fast_lsb (infl.c:938), which
 * builds an 8-bit direct lookup table for the 19-symbol code-length alphabet.
 * It is restated below from RFC 1951 section 3.2.2 (canonical code, LSB-first
 * packing) and from its call site (entry = {len, sym}, len == 0 means "no
 * code", false on an over-subscribed length set; test/test_files.c:546-593
 * pins the rejection behaviour of the sibling huff_init_lsb).  Whether the
 * real function also rejects an incomplete code-length code is not pinned by
 * anything in the reference tree: this file accepts it, exactly as
 * infl_ft_build (infl.c:485-490) accepts incomplete literal/distance codes.
 *
 * Parity is pinned by the reference's own fixtures: the 139 raw/compressed
 * file pairs of test/data and the inline known-answer vectors of
 * test/test_files.c:468-476, 627-633 (see tests/test_oracle_golden.py), and is
 * cross-checked against the system zlib on valid streams.  The reference
 * itself cannot be built here (missing huff/huff.h), so no reference binary
 * takes part.
 *
 * A stream that starts with stored blocks and continues with a compressed
 * block leaves the reference's fast paths for its older huff-table decoder
 * (infl.c:256-259, 1553-1664), whose table acceptance lives entirely inside
 * the missing dependency.  Here the continuation is decoded with the
 * packed-table rules; for valid streams every conformant inflater produces the
 * same bytes, for malformed continuations parity is unpinned.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>

#include "infl_oracle.h"

/* ------------------------------------------------------------------------ */
/* RFC 1951 tables (apicommon.h:36-64)                                      */
/* ------------------------------------------------------------------------ */

static const uint16_t k_len_base[29] = {
  3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67,
  83, 99, 115, 131, 163, 195, 227, 258
};
static const uint8_t k_len_xbits[29] = {
  0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5,
  5, 5, 0
};
static const uint16_t k_dist_base[30] = {
  1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513,
  769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577
};
static const uint8_t k_dist_xbits[30] = {
  0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11,
  11, 12, 12, 13, 13
};
static const uint8_t k_clen_order[19] = {
  16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15
};

/* ------------------------------------------------------------------------ */
/* Packed decode-table entry (infl.c:385-397)                               */
/*   bits 0-4 total = codelen + xbits, 5-8 codelen, 9-12 xbits,             */
/*   13 LITERAL, 14 END, 15 SUBTABLE, 16-31 base / subtable offset          */
/* ------------------------------------------------------------------------ */
#define E_TOTAL(e)    ((unsigned)((e) & 31u))
#define E_CODELEN(e)  ((unsigned)(((e) >> 5) & 15u))
#define E_LITERAL     (1u << 13)
#define E_END         (1u << 14)
#define E_SUB         (1u << 15)
#define E_BASE(e)     ((unsigned)((e) >> 16))

#define LIT_BITS   10u
#define DIST_BITS  8u
#define LIT_CAP    2048u
#define DIST_CAP   512u

static inline uint32_t
mk_entry(unsigned base, unsigned xbits, unsigned codelen, uint32_t flags) {
  return ((uint32_t)base << 16) | flags | ((uint32_t)xbits << 9) |
         ((uint32_t)codelen << 5) | (uint32_t)(codelen + xbits);
}

static inline uint32_t
litlen_entry(unsigned sym, unsigned len) {      /* infl.c:425-439 */
  if (sym < 256)  return mk_entry(sym, 0, len, E_LITERAL);
  if (sym == 256) return mk_entry(0, 0, len, E_END);
  if (sym <= 285) return mk_entry(k_len_base[sym - 257], k_len_xbits[sym - 257], len, 0);
  return 0;                                     /* 286, 287: no entry */
}

static inline uint32_t
dist_entry(unsigned sym, unsigned len) {        /* infl.c:441-450 */
  if (sym > 29) return 0;                       /* 30, 31: no entry */
  return mk_entry(k_dist_base[sym], k_dist_xbits[sym], len, 0);
}

static inline unsigned
bitrev(unsigned v, unsigned len) {              /* infl.c:416-423 */
  unsigned r = 0;
  for (unsigned i = 0; i < len; i++) { r = (r << 1) | (v & 1u); v >>= 1; }
  return r;
}

/*
 * Two-level table build (infl.c:452-578).  Returns 0 on success.  Rejects a
 * length above 15, an over-subscribed length set and a table that does not fit
 * `cap` entries; accepts incomplete sets (also the all-zero one).
 */
int
oracle_table_build(uint32_t *table, unsigned *used_out, const uint8_t *lens,
                   unsigned nsyms, unsigned tablebits, unsigned cap, int litlen) {
  unsigned count[16] = {0}, first[16], next[16];
  uint8_t  subbits[1u << LIT_BITS];
  uint16_t subbase[1u << LIT_BITS];
  unsigned main_size = 1u << tablebits, used = main_size;
  int      left;

  if (tablebits > LIT_BITS || main_size > cap) return -1;
  memset(subbits, 0, main_size);

  for (unsigned s = 0; s < nsyms; s++) {
    if (lens[s] > 15) return -1;
    count[lens[s]]++;
  }
  left = 1;
  for (unsigned l = 1; l <= 15; l++) {
    left = (left << 1) - (int)count[l];
    if (left < 0) return -1;
  }
  first[0] = 0;
  {
    unsigned prev = 0;
    for (unsigned l = 1; l <= 15; l++) {
      /* count[0] (unused symbols) takes part for l == 1 as in infl.c:494-498:
         it only adds a multiple of 2^l to first[l], which the l-bit reversal
         below drops again, so the codes are the canonical ones */
      first[l] = (prev + count[l - 1]) << 1;
      prev = first[l];
    }
  }
  memcpy(next, first, sizeof(next));

  for (unsigned s = 0; s < nsyms; s++) {
    unsigned len = lens[s], rev, need;
    if (!len) continue;
    rev = bitrev((uint16_t)next[len]++, len);
    if (len <= tablebits) continue;
    need = len - tablebits;
    if (subbits[rev & (main_size - 1u)] < need)
      subbits[rev & (main_size - 1u)] = (uint8_t)need;
  }
  for (unsigned p = 0; p < main_size; p++) {
    unsigned size;
    if (!subbits[p]) continue;
    size = 1u << subbits[p];
    if (used + size > cap) return -1;
    subbase[p] = (uint16_t)used;
    used += size;
  }

  memset(table, 0, used * sizeof(table[0]));
  for (unsigned p = 0; p < main_size; p++)
    if (subbits[p])
      table[p] = ((uint32_t)subbase[p] << 16) | E_SUB | ((uint32_t)subbits[p] << 5) | tablebits;

  memcpy(next, first, sizeof(next));
  for (unsigned s = 0; s < nsyms; s++) {
    unsigned len = lens[s], rev;
    uint32_t e;
    if (!len) continue;
    rev = bitrev((uint16_t)next[len]++, len);
    e = litlen ? litlen_entry(s, len) : dist_entry(s, len);
    if (!e) continue;
    if (len <= tablebits) {
      for (unsigned i = rev; i < main_size; i += 1u << len) table[i] = e;
    } else {
      unsigned p = rev & (main_size - 1u), end = 1u << subbits[p];
      for (unsigned i = rev >> tablebits; i < end; i += 1u << (len - tablebits))
        table[subbase[p] + i] = e;
    }
  }
  *used_out = used;
  return 0;
}

/*
 * Stand-in for recp/huff's huff_init_fast_lsb on the 19-symbol code-length
 * alphabet (call site infl.c:938): 256-entry direct table indexed by the next
 * 8 input bits; value = sym << 8 | len, 0 = no code.
 */
static int
clen_table_build(uint16_t tab[256], const uint8_t lens[19]) {
  unsigned count[8] = {0}, next[8], code = 0;
  int left = 1;

  for (unsigned s = 0; s < 19; s++) count[lens[s]]++;   /* lens are 3-bit */
  for (unsigned l = 1; l <= 7; l++) {
    left = (left << 1) - (int)count[l];
    if (left < 0) return -1;
  }
  count[0] = 0;
  for (unsigned l = 1; l <= 7; l++) { code = (code + count[l - 1]) << 1; next[l] = code; }
  memset(tab, 0, 256 * sizeof(tab[0]));
  for (unsigned s = 0; s < 19; s++) {
    unsigned len = lens[s];
    if (!len) continue;
    for (unsigned i = bitrev(next[len]++, len); i < 256; i += 1u << len)
      tab[i] = (uint16_t)((s << 8) | len);
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* Bit reader: 64-bit LSB-first reservoir holding only real input bits      */
/* (infl.c:580-633).  Bits above `n` are always zero.                       */
/* ------------------------------------------------------------------------ */
typedef struct {
  const uint8_t *p, *end;
  uint64_t       bits;
  unsigned       n;
} rd_t;

static inline void
rd_fill(rd_t *r) {                  /* pull whole bytes until >= 56 bits or EOF */
  if (r->n >= 56) return;
  if ((size_t)(r->end - r->p) >= 8) {
    uint64_t w;
    unsigned k = (64u - r->n) >> 3;             /* bytes that fit, 1..8 */
    memcpy(&w, r->p, 8);
    if (k < 8) w &= ((uint64_t)1 << (k * 8)) - 1u;
    r->bits |= w << r->n;
    r->p    += k;
    r->n    += k * 8;
    return;
  }
  while (r->n <= 56 && r->p < r->end) {
    r->bits |= (uint64_t)*r->p++ << r->n;
    r->n    += 8;
  }
}

static inline void
rd_drop(rd_t *r, unsigned k) { r->bits >>= k; r->n -= k; }

static inline uint32_t
tab_lookup(const uint32_t *tab, unsigned mainbits, uint64_t bits) {  /* infl.c:635-657 */
  uint32_t e = tab[bits & ((1u << mainbits) - 1u)];
  if (e & E_SUB)
    e = tab[E_BASE(e) + ((bits >> mainbits) & ((1u << E_CODELEN(e)) - 1u))];
  return e;
}

typedef struct {
  uint32_t lit[LIT_CAP];
  uint32_t dist[DIST_CAP];
} tabs_t;

/* stored block reached inside the packed-table decoder (infl.c:659-715) */
static int
stored_in_stream(rd_t *r, uint8_t *dst, size_t *dpos, size_t cap) {
  unsigned len, nlen, k;
  size_t   rem;

  rd_drop(r, r->n & 7u);
  rd_fill(r);
  if (r->n < 32) return ORACLE_D_NEED_INPUT;
  len  = (unsigned)(r->bits & 0xffffu);
  nlen = (unsigned)((r->bits >> 16) & 0xffffu);
  rd_drop(r, 32);
  if (len != (nlen ^ 0xffffu)) return ORACLE_D_INVALID;
  if (len > cap - *dpos)       return ORACLE_D_FULL;

  rem = len;
  k   = r->n >> 3;
  if (k > rem) k = (unsigned)rem;
  for (unsigned i = 0; i < k; i++) { dst[*dpos + i] = (uint8_t)r->bits; rd_drop(r, 8); }
  *dpos += k;
  rem   -= k;
  if ((size_t)(r->end - r->p) < rem) return ORACLE_D_NEED_INPUT;
  if (rem) memcpy(dst + *dpos, r->p, rem);
  r->p  += rem;
  *dpos += rem;
  return ORACLE_D_OK;
}

/* optional observer of matches (statistics only) */
static void (*g_match_hook)(unsigned pos, unsigned len, unsigned dist);
void
oracle_set_match_hook(void (*hook)(unsigned, unsigned, unsigned)) {
  g_match_hook = hook;
}

/* optional observer of dynamic headers (statistics for table sizing experiments) */
static void (*g_header_hook)(const uint8_t *lens, unsigned hlit, unsigned hdist);

void
oracle_set_header_hook(void (*hook)(const uint8_t *, unsigned, unsigned)) {
  g_header_hook = hook;
}

/* dynamic block header (infl.c:903-1000) */
static int
dynamic_header(rd_t *r, tabs_t *t) {
  uint8_t  clens[19] = {0};
  uint8_t  lens[288 + 32];
  uint16_t ctab[256];
  unsigned hlit, hdist, hclen, n, i, used;

  rd_fill(r);
  if (r->n < 14) return ORACLE_D_NEED_INPUT;
  hlit  = (unsigned)(r->bits & 31u) + 257u;
  hdist = (unsigned)((r->bits >> 5) & 31u) + 1u;
  hclen = (unsigned)((r->bits >> 10) & 15u) + 4u;
  n     = hlit + hdist;
  rd_drop(r, 14);
  if (n > 320u) return ORACLE_D_INVALID;        /* cannot trigger: 288 + 32 */

  for (i = 0; i < hclen; i++) {
    rd_fill(r);
    if (r->n < 3) return ORACLE_D_NEED_INPUT;
    clens[k_clen_order[i]] = (uint8_t)(r->bits & 7u);
    rd_drop(r, 3);
  }
  if (clen_table_build(ctab, clens) != 0) return ORACLE_D_INVALID;

  memset(lens, 0, sizeof(lens));
  for (i = 0; i < n;) {
    unsigned e, sym, rep;

    rd_fill(r);
    if (r->n < 7) return ORACLE_D_NEED_INPUT;   /* infl.c:947: 7, not the code's own length */
    e = ctab[r->bits & 0xffu];
    if (!(e & 0xffu)) return ORACLE_D_INVALID;
    sym = e >> 8;
    rd_drop(r, e & 0xffu);

    if (sym < 16) {
      lens[i++] = (uint8_t)sym;
    } else if (sym == 16) {
      if (r->n < 2) return ORACLE_D_NEED_INPUT;
      if (i == 0)   return ORACLE_D_INVALID;
      rep = 3u + (unsigned)(r->bits & 3u);
      rd_drop(r, 2);
      if (i + rep > n) return ORACLE_D_INVALID;
      memset(lens + i, lens[i - 1], rep);
      i += rep;
    } else if (sym == 17) {
      if (r->n < 3) return ORACLE_D_NEED_INPUT;
      rep = 3u + (unsigned)(r->bits & 7u);
      rd_drop(r, 3);
      if (i + rep > n) return ORACLE_D_INVALID;
      i += rep;
    } else {
      if (r->n < 7) return ORACLE_D_NEED_INPUT;
      rep = 11u + (unsigned)(r->bits & 127u);
      rd_drop(r, 7);
      if (i + rep > n) return ORACLE_D_INVALID;
      i += rep;
    }
  }

  if (g_header_hook) g_header_hook(lens, hlit, hdist);
  if (oracle_table_build(t->lit, &used, lens, hlit, LIT_BITS, LIT_CAP, 1) != 0 ||
      oracle_table_build(t->dist, &used, lens + hlit, hdist, DIST_BITS, DIST_CAP, 0) != 0)
    return ORACLE_D_INVALID;
  return ORACLE_D_OK;
}

/*
 * One Huffman-coded block (infl.c:717-901).  The reference's copy kernels may
 * scribble up to 39 bytes past a match inside the destination capacity; only
 * dst[0..dpos) is part of the result, so a byte-exact forward copy is an
 * equivalent statement of them.
 */
static int
huffman_block(rd_t *r, uint8_t *dst, size_t *dpos, size_t cap, const tabs_t *t) {
  size_t pos = *dpos;

  /* Fast loop for the bulk of a block: at least 16 input bytes and 258 + 40 output bytes are
     left, so a symbol can neither run out of input nor overflow the output, and copies may
     write whole 8-byte words past the end of a match (inside the capacity), as the
     reference's own copy kernels do (infl.c:353-376).  Same decisions as the careful loop
     below, which finishes the block. */
  while ((size_t)(r->end - r->p) >= 16 && cap - pos >= 258 + 40) {
    uint32_t e;
    unsigned total, len, dist;
    uint64_t w;

    if (r->n < 48) {                               /* >= 48 bits cover literal/length + distance */
      unsigned k = (64u - r->n) >> 3;
      memcpy(&w, r->p, 8);
      if (k < 8) w &= ((uint64_t)1 << (k * 8)) - 1u;
      r->bits |= w << r->n;
      r->p    += k;
      r->n    += k * 8;
    }
    e = tab_lookup(t->lit, LIT_BITS, r->bits);
    if (!e) return ORACLE_D_INVALID;
    total = E_TOTAL(e);
    if (e & E_LITERAL) {
      rd_drop(r, total);
      dst[pos++] = (uint8_t)E_BASE(e);
      e = tab_lookup(t->lit, LIT_BITS, r->bits);   /* one literal of run-ahead */
      if ((e & E_LITERAL) && r->n >= 32) {
        rd_drop(r, E_TOTAL(e));
        dst[pos++] = (uint8_t)E_BASE(e);
      }
      continue;
    }
    if (e & E_END) {
      rd_drop(r, total);
      *dpos = pos;
      return ORACLE_D_OK;
    }
    len = E_BASE(e) + (unsigned)((r->bits & (((uint64_t)1 << total) - 1u)) >> E_CODELEN(e));
    rd_drop(r, total);
    e = tab_lookup(t->dist, DIST_BITS, r->bits);
    if (!e) return ORACLE_D_INVALID;
    total = E_TOTAL(e);
    dist = E_BASE(e) + (unsigned)((r->bits & (((uint64_t)1 << total) - 1u)) >> E_CODELEN(e));
    rd_drop(r, total);
    if (!dist || (size_t)dist > pos) return ORACLE_D_INVALID;
    if (dist >= 8) {
      const uint8_t *sp = dst + pos - dist;
      uint8_t *dp = dst + pos, *de = dp + len;
      do { memcpy(&w, sp, 8); memcpy(dp, &w, 8); sp += 8; dp += 8; } while (dp < de);
    } else if (dist == 1) {
      memset(dst + pos, dst[pos - 1], len);
    } else {
      const uint8_t *sp = dst + pos - dist;
      for (unsigned i = 0; i < len; i++) dst[pos + i] = sp[i];
    }
    pos += len;
  }

  for (;;) {
    uint32_t e;
    unsigned total, len, dist;

    rd_fill(r);
    e = tab_lookup(t->lit, LIT_BITS, r->bits);
    if (!e) return ORACLE_D_INVALID;
    total = E_TOTAL(e);
    if (r->n < total) return ORACLE_D_NEED_INPUT;

    if (e & E_LITERAL) {
      rd_drop(r, total);
      if (pos >= cap) return ORACLE_D_FULL;
      dst[pos++] = (uint8_t)E_BASE(e);
      continue;
    }
    if (e & E_END) {
      rd_drop(r, total);
      break;
    }

    len = E_BASE(e) + (unsigned)((r->bits & (((uint64_t)1 << total) - 1u)) >> E_CODELEN(e));
    rd_drop(r, total);

    rd_fill(r);
    e = tab_lookup(t->dist, DIST_BITS, r->bits);
    if (!e) return ORACLE_D_INVALID;
    total = E_TOTAL(e);
    if (r->n < total) return ORACLE_D_NEED_INPUT;
    dist = E_BASE(e) + (unsigned)((r->bits & (((uint64_t)1 << total) - 1u)) >> E_CODELEN(e));
    rd_drop(r, total);

    if (!dist || (size_t)dist > pos) return ORACLE_D_INVALID;
    if (len > cap - pos)             return ORACLE_D_FULL;
    if (g_match_hook) g_match_hook((unsigned)pos, len, dist);

    if (dist >= len) {
      memcpy(dst + pos, dst + pos - dist, len);
      pos += len;
    } else if (dist == 1) {
      memset(dst + pos, dst[pos - 1], len);
      pos += len;
    } else {
      const uint8_t *s = dst + pos - dist;
      for (unsigned i = 0; i < len; i++) dst[pos + i] = s[i];
      pos += len;
    }
  }
  *dpos = pos;
  return ORACLE_D_OK;
}

static tabs_t g_fixed;
static pthread_once_t g_fixed_once = PTHREAD_ONCE_INIT;

static void
fixed_init(void) {                               /* infl.c:1044-1052, apicommon.h:53-64 */
  uint8_t  lens[320];
  unsigned used;
  memset(lens, 8, 144);
  memset(lens + 144, 9, 112);
  memset(lens + 256, 7, 24);
  memset(lens + 280, 8, 8);
  memset(lens + 288, 5, 32);
  oracle_table_build(g_fixed.lit, &used, lens, 288, LIT_BITS, LIT_CAP, 1);
  oracle_table_build(g_fixed.dist, &used, lens + 288, 32, DIST_BITS, DIST_CAP, 0);
}

/* block loop of the packed-table decoder (infl.c:1063-1092) */
static int
block_loop(rd_t *r, uint8_t *dst, size_t *dpos, size_t cap) {
  tabs_t   dyn;
  unsigned bfinal = 0, btype;
  int      rc;

  pthread_once(&g_fixed_once, fixed_init);
  while (!bfinal) {
    rd_fill(r);
    if (r->n < 3) return ORACLE_D_NEED_INPUT;
    bfinal = (unsigned)(r->bits & 1u);
    btype  = (unsigned)((r->bits >> 1) & 3u);
    rd_drop(r, 3);
    switch (btype) {
      case 0:
        rc = stored_in_stream(r, dst, dpos, cap);
        break;
      case 1:
        rc = huffman_block(r, dst, dpos, cap, &g_fixed);
        break;
      case 2:
        rc = dynamic_header(r, &dyn);
        if (rc == ORACLE_D_OK) rc = huffman_block(r, dst, dpos, cap, &dyn);
        break;
      default:
        rc = ORACLE_D_INVALID;
        break;
    }
    if (rc != ORACLE_D_OK) return rc;
  }
  return ORACLE_D_OK;
}

static int
zlib_header_ok(const uint8_t *p) {               /* infl.c:233-238 */
  unsigned cmf = p[0], flg = p[1];
  return (cmf & 15u) == 8u && (cmf >> 4) <= 7u && ((cmf << 8) + flg) % 31u == 0 &&
         !(flg & 0x20u);
}

/*
 * infl() for one contiguous input (what infl_buf() runs, infl.h:125-145).
 * `detail` receives ORACLE_D_*: why a failing stream failed, before the
 * reference flattens every failure to UNZ_ERR (infl.c:1074-1087).
 */
int
oracle_infl_detail(const uint8_t *src, uint32_t srclen, uint8_t *dst, uint32_t dstlen,
                   int flags, uint32_t *dstpos_out, int *detail_out) {
  const uint8_t *p = src, *end = src + srclen;
  size_t   dpos = 0, cap = dstlen;
  int      zlib = flags == ORACLE_FLAG_ZLIB, try_stored = 1, detail;
  rd_t     r;

  *dstpos_out = 0;
  if (detail_out) *detail_out = ORACLE_D_OK;
  if (srclen == 0) return ORACLE_OK;             /* infl.c:1547-1550 */

  /* infl.c:1519-1531: which fast path gets the first look */
  if (!zlib)            try_stored = ((p[0] >> 1) & 3u) == 0;
  else if (srclen >= 3) try_stored = ((p[2] >> 1) & 3u) == 0;

  if (try_stored) {                              /* infl.c:206-267 */
    int go_stored = 1;
    if (zlib) {
      if (srclen < 2) { detail = ORACLE_D_NEED_INPUT; goto fail; }
      if (srclen < 3) go_stored = 0;             /* falls to the table decoder */
      else {
        if (!zlib_header_ok(p)) { detail = ORACLE_D_INVALID; goto fail; }
        p += 2;
      }
    }
    if (go_stored) {
      for (;;) {                                 /* infl.c:94-131 with nbits == 0 */
        unsigned bfinal, len, nlen;
        if (p >= end) { detail = ORACLE_D_NEED_INPUT; goto fail; }
        bfinal = p[0] & 1u;
        if (((p[0] >> 1) & 3u) != 0) break;      /* continue below from byte p */
        if ((size_t)(end - p) < 5) { detail = ORACLE_D_NEED_INPUT; goto fail; }
        len  = p[1] | ((unsigned)p[2] << 8);
        nlen = p[3] | ((unsigned)p[4] << 8);
        if (len != (nlen ^ 0xffffu)) { detail = ORACLE_D_INVALID; goto fail; }
        if (len > cap - dpos)        { detail = ORACLE_D_FULL;    goto fail; }
        p += 5;
        if ((size_t)(end - p) < len) { detail = ORACLE_D_NEED_INPUT; goto fail; }
        if (len) memcpy(dst + dpos, p, len);
        p    += len;
        dpos += len;
        if (bfinal) { *dstpos_out = (uint32_t)dpos; return ORACLE_OK; }
      }
      /* stored prefix followed by a compressed block: infl.c:256-259 hands the
         stream (dstpos = prefix) to the huff-table decoder; see file header */
      *dstpos_out = (uint32_t)dpos;
      r.p = p; r.end = end; r.bits = 0; r.n = 0;
      detail = block_loop(&r, dst, &dpos, cap);
      if (detail != ORACLE_D_OK) {
        if (detail_out) *detail_out = detail;
        return ORACLE_ERR;                       /* dstpos stays at the prefix */
      }
      *dstpos_out = (uint32_t)dpos;
      return ORACLE_OK;
    }
  }

  /* infl.c:1002-1104 */
  if (zlib) {
    if (srclen < 2) { detail = ORACLE_D_NEED_INPUT; goto fail; }
    if (!zlib_header_ok(p)) { detail = ORACLE_D_INVALID; goto fail; }
    p += 2;
  }
  r.p = p; r.end = end; r.bits = 0; r.n = 0;
  detail = block_loop(&r, dst, &dpos, cap);
  if (detail != ORACLE_D_OK) goto fail;
  *dstpos_out = (uint32_t)dpos;
  return ORACLE_OK;

fail:
  if (detail_out) *detail_out = detail;
  *dstpos_out = 0;
  return ORACLE_ERR;
}

int
oracle_infl(const uint8_t *src, uint32_t srclen, uint8_t *dst, uint32_t dstlen,
            int flags, uint32_t *dstpos_out) {
  return oracle_infl_detail(src, srclen, dst, dstlen, flags, dstpos_out, NULL);
}

/* ------------------------------------------------------------------------ */
/* Batch driver with static stream partitioning over host threads; used for  */
/* bulk comparison in tests and for the CPU baseline timing.                 */
/* ------------------------------------------------------------------------ */
typedef struct {
  const uint8_t  *src;
  const uint64_t *src_off;
  uint8_t        *dst;
  const uint64_t *dst_off;
  int             flags;
  size_t          lo, hi;
  int32_t        *status;
  uint32_t       *outlen;
  int             per_call_objects;
} job_t;

/* What infl_buf() does around the decode when it is called as it stands, one call per stream
   (include/defl/infl.h:125-145): infl_init allocates a zeroed stream object (infl.c:1677-1695;
   src/common.h:150-188: a pool of 1024 chunk-struct pointers, 32 page pointers, bit state, streaming
   state and two recp/huff tables whose size is not known here: 9 KiB stands for it), infl_include
   copies an input of at most 8 KiB into a fresh 64-byte aligned 32 KiB page (mem.c:35-78, 135-151)
   and otherwise allocates a chunk node (mem.c:185-201), infl_destroy walks the 32 + 1024 pool slots
   and frees (mem.c:331-361).  The other pattern the reference offers, one stream object reset
   between inputs (mem.c:257-275), has none of this per stream and is what the plain batch driver
   times. */
#define ORACLE_STREAM_OBJECT_BYTES 9216u
static const uint8_t *
as_is_begin(const uint8_t *src, uint32_t len, void **obj, void **page) {
  *obj  = calloc(1, ORACLE_STREAM_OBJECT_BYTES);
  *page = NULL;
  if (len <= 8192u) {
    if (posix_memalign(page, 64, 32768) != 0) *page = NULL;
    if (*page) {
      memcpy(*page, src, len);
      return (const uint8_t *)*page;
    }
  } else {
    *page = calloc(1, 64);                        /* the borrowed chunk's node */
  }
  return src;
}
static void
as_is_end(void *obj, void *page) {
  if (obj) {
    volatile const uintptr_t *slots = (volatile const uintptr_t *)obj;
    uintptr_t                 any = 0;
    for (unsigned k = 0; k < 32u + 1024u; k++) any |= slots[k];   /* the walk over the pool slots */
    (void)any;
  }
  free(page);
  free(obj);
}

static void *
batch_worker(void *arg) {
  job_t *j = (job_t *)arg;
  for (size_t i = j->lo; i < j->hi; i++) {
    uint32_t out = 0;
    if (j->per_call_objects) {
      void          *obj, *page;
      const uint32_t len = (uint32_t)(j->src_off[i + 1] - j->src_off[i]);
      const uint8_t *in  = as_is_begin(j->src + j->src_off[i], len, &obj, &page);
      j->status[i] = obj ? oracle_infl(in, len, j->dst + j->dst_off[i],
                                       (uint32_t)(j->dst_off[i + 1] - j->dst_off[i]), j->flags, &out)
                         : ORACLE_ERR;
      as_is_end(obj, page);
      j->outlen[i] = out;
      continue;
    }
    j->status[i] = oracle_infl(j->src + j->src_off[i],
                               (uint32_t)(j->src_off[i + 1] - j->src_off[i]),
                               j->dst + j->dst_off[i],
                               (uint32_t)(j->dst_off[i + 1] - j->dst_off[i]),
                               j->flags, &out);
    j->outlen[i] = out;
  }
  return NULL;
}

static int
batch_run(const uint8_t *src, const uint64_t *src_off, uint8_t *dst, const uint64_t *dst_off, int flags,
          size_t n, int32_t *status, uint32_t *outlen, int nthreads, int per_call_objects);

int
oracle_infl_batch(const uint8_t *src, const uint64_t *src_off, uint8_t *dst,
                  const uint64_t *dst_off, int flags, size_t n, int32_t *status,
                  uint32_t *outlen, int nthreads) {
  return batch_run(src, src_off, dst, dst_off, flags, n, status, outlen, nthreads, 0);
}

int
oracle_infl_batch_as_is(const uint8_t *src, const uint64_t *src_off, uint8_t *dst,
                        const uint64_t *dst_off, int flags, size_t n, int32_t *status,
                        uint32_t *outlen, int nthreads) {
  return batch_run(src, src_off, dst, dst_off, flags, n, status, outlen, nthreads, 1);
}

static int
batch_run(const uint8_t *src, const uint64_t *src_off, uint8_t *dst, const uint64_t *dst_off, int flags,
          size_t n, int32_t *status, uint32_t *outlen, int nthreads, int per_call_objects) {
  pthread_t *th;
  job_t     *jobs;

  if (nthreads < 1) nthreads = 1;
  if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
  pthread_once(&g_fixed_once, fixed_init);
  th   = (pthread_t *)calloc((size_t)nthreads, sizeof(*th));
  jobs = (job_t *)calloc((size_t)nthreads, sizeof(*jobs));
  if (!th || !jobs) { free(th); free(jobs); return -1; }
  for (int t = 0; t < nthreads; t++) {
    jobs[t] = (job_t){src, src_off, dst, dst_off, flags,
                      n * (size_t)t / (size_t)nthreads,
                      n * (size_t)(t + 1) / (size_t)nthreads, status, outlen, per_call_objects};
    if (nthreads == 1) batch_worker(&jobs[t]);
    else pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
  }
  if (nthreads > 1)
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
  return 0;
}
