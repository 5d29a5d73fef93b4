/*
 * infl_host.c -- host side of the inflate API (plain C, like the reference's
 * host code).  It owns the opaque stream object and the input piece list and
 * hands complete inputs to the GPU engine; it contains no decoder.
 *
 * Functions of the reference this file replaces:
 *   infl_init            src/infl/infl.c:1677-1695
 *   infl_include         src/infl/mem.c:114-216
 *   infl_reset_pool      src/infl/mem.c:219-232
 *   infl_reset           src/infl/mem.c:257-275
 *   infl_resize_output   src/infl/mem.c:277-288
 *   infl_output_pos      src/infl/mem.c:290-294
 *   infl_input_pos       src/infl/mem.c:296-329
 *   infl_destroy         src/infl/mem.c:331-361
 *   infl                 src/infl/infl.c:1507-1675 (dispatch only; decode is on the GPU)
 *   infl_stream          src/infl/stream.c:468-794
 * plus infl_many(), the deferred form of infl() over many streams.
 *
 * Input ownership follows the reference: a piece of at most 8 KiB
 * (UNZ_CHUNK_APPEND_THRESHOLD, src/common.h:46) is copied and coalesced into
 * 32 KiB pages owned by the stream (src/common.h:45); a larger piece is
 * borrowed by pointer and must stay valid until infl() has returned.  Where the
 * reference stops copying once its 32-page pool is used up (mem.c:152-155) this
 * implementation keeps allocating pages, which is the safer superset.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <sys/mman.h>

#include "../../include/defl/infl.h"
#include "../../include/defl/infl_batch.h"
#include "infl_device.h"

#define PAGE_SIZE        32768u   /* UNZ_CHUNK_PAGE_SIZE,        src/common.h:45 */
#define APPEND_THRESHOLD 8192u    /* UNZ_CHUNK_APPEND_THRESHOLD, src/common.h:46 */

/* Process-wide pool of the 32 KiB pages small pieces are copied into (the reference pools them per
   stream, src/infl/mem.c:35-105; its pools die with the stream, so a program that makes one stream
   per image pays for fresh pages every time).  Pages are carved from slabs aligned to 2 MiB for
   which huge pages are asked (the first touch of a page otherwise costs a fault per 4 KiB, which
   is what infl_include cost on fresh streams: 3.4 us per call against 0.7 us on reused ones), go
   back to a free list with infl_destroy() and are handed out again, already touched, to the next
   stream.  Beyond POOL_KEEP_PAGES free pages the memory of a returned page is given back to the
   system (the address stays in the list). */
#define HUGE_BYTES      (2u << 20)
#define SLAB_BYTES      (8u << 20)
#define POOL_KEEP_PAGES 8192u      /* 256 MiB of touched pages kept for reuse */

static pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;
static uint8_t       **g_pool_free;
static size_t          g_pool_nfree, g_pool_cap;
static uint8_t        *g_slab;
static size_t          g_slab_left;

static uint8_t *
page_acquire(void) {
  uint8_t *pg = NULL;
  pthread_mutex_lock(&g_pool_mu);
  if (g_pool_nfree) {
    pg = g_pool_free[--g_pool_nfree];
  } else {
    if (g_slab_left < PAGE_SIZE) {
      void *m = mmap(NULL, SLAB_BYTES + HUGE_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (m != MAP_FAILED) {
        g_slab      = (uint8_t *)(((uintptr_t)m + HUGE_BYTES - 1u) & ~(uintptr_t)(HUGE_BYTES - 1u));
        g_slab_left = SLAB_BYTES;
#ifdef MADV_HUGEPAGE
        (void)madvise(g_slab, SLAB_BYTES, MADV_HUGEPAGE);   /* (where the system declines, ordinary pages serve) */
#endif
      }
    }
    if (g_slab_left >= PAGE_SIZE) {
      pg           = g_slab;
      g_slab      += PAGE_SIZE;
      g_slab_left -= PAGE_SIZE;
    }
  }
  pthread_mutex_unlock(&g_pool_mu);
  return pg;
}

static void
page_release(uint8_t *pg) {
  pthread_mutex_lock(&g_pool_mu);
  if (g_pool_nfree == g_pool_cap) {
    size_t    cap = g_pool_cap ? g_pool_cap * 2 : 256;
    uint8_t **np  = (uint8_t **)realloc(g_pool_free, cap * sizeof(*np));
    if (np) {
      g_pool_free = np;
      g_pool_cap  = cap;
    }
  }
  if (g_pool_nfree < g_pool_cap) {
    if (g_pool_nfree >= POOL_KEEP_PAGES)
      (void)madvise(pg, PAGE_SIZE, MADV_DONTNEED);
    g_pool_free[g_pool_nfree++] = pg;
  } else {
    (void)madvise(pg, PAGE_SIZE, MADV_DONTNEED);      /* no room to remember it: the address is lost, the memory is not held */
  }
  pthread_mutex_unlock(&g_pool_mu);
}

typedef struct piece_t {
  const uint8_t *p;
  uint32_t       len;
  int32_t        page;            /* index into pages[] or -1 when borrowed */
} piece_t;

struct unz__stream_t {
  uint8_t  *dst;
  uint32_t  dstlen;
  uint32_t  dstpos;
  uint32_t  inpos;
  int       flags;

  piece_t  *pieces;
  size_t    npieces, cappieces;
  size_t    srclen;               /* sum of piece lengths */

  uint8_t **pages;                /* owned 32 KiB pages, kept across resets */
  size_t    npages, cappages;
  size_t    pages_used;
  uint32_t  page_fill;            /* bytes used in the page being appended to */
  int       appending;            /* last piece is an owned page with room left */

  uint8_t  *gather;               /* contiguous staging for multi-piece inputs */
  size_t    gather_cap;

  int       stream_done;          /* infl_stream() reached the final block */
  int       stream_started;       /* infl_stream() has begun to decode (the reference's state != NONE) */
  uint8_t  *acc;                  /* infl_stream(): the piece list so far as one contiguous range (what the device sees) */
  size_t    acc_len, acc_cap;
  size_t    mir_piece, mir_off;   /* first piece, and bytes of it, not yet copied to acc */
  void     *sctx;                 /* its device context (input, output and decode position stay on the device) */
};

static void
warn_no_device(void) {
  static int warned;
  if (!warned) {
    warned = 1;
    fprintf(stderr, "defl: no usable HIP device; the inflate engine has no CPU decoder\n");
  }
}

UNZ_EXPORT infl_stream_t *
infl_init(void * __restrict dst, uint32_t dstlen, int flags) {
  infl_stream_t *st = (infl_stream_t *)calloc(1, sizeof(*st));
  if (!st)
    return NULL;
  st->dst    = (uint8_t *)dst;
  st->dstlen = dstlen;
  st->flags  = flags;
  return st;
}

UNZ_EXPORT void
infl_reset_pool(infl_stream_t * __restrict st) {
  if (!st)
    return;
  st->npieces    = 0;
  st->srclen     = 0;
  st->pages_used = 0;
  st->page_fill  = 0;
  st->appending  = 0;
  st->acc_len    = 0;
  st->mir_piece  = 0;
  st->mir_off    = 0;
  st->stream_started = 0;
  if (st->sctx)                                   /* the incremental decode starts over with the input */
    infl_rt_stream_restart(st->sctx);
}

UNZ_EXPORT void
infl_reset(infl_stream_t * __restrict st, void * __restrict dst, uint32_t dstlen, int flags) {
  if (!st)
    return;
  infl_reset_pool(st);
  st->dst         = (uint8_t *)dst;
  st->dstlen      = dstlen;
  st->dstpos      = 0;
  st->inpos       = 0;
  st->flags       = flags;
  st->stream_done = 0;
}

UNZ_EXPORT int
infl_resize_output(infl_stream_t * __restrict st, void * __restrict dst, uint32_t dstlen) {
  if (!st || !dst || dstlen < st->dstpos)
    return UNZ_ERR;
  st->dst    = (uint8_t *)dst;
  st->dstlen = dstlen;
  return UNZ_OK;
}

UNZ_EXPORT uint32_t
infl_output_pos(const infl_stream_t * __restrict st) {
  return st ? st->dstpos : 0u;
}

UNZ_EXPORT uint32_t
infl_input_pos(const infl_stream_t * __restrict st) {
  return st ? st->inpos : 0u;
}

UNZ_EXPORT void
infl_destroy(infl_stream_t * __restrict st) {
  if (!st)
    return;
  for (size_t i = 0; i < st->npages; i++)
    page_release(st->pages[i]);
  free(st->pages);
  free(st->pieces);
  free(st->gather);
  free(st->acc);
  if (st->sctx)
    infl_rt_stream_free(st->sctx);
  free(st);
}

static piece_t *
new_piece(infl_stream_t *st) {
  if (st->npieces == st->cappieces) {
    size_t   cap = st->cappieces ? st->cappieces * 2 : 16;
    piece_t *np  = (piece_t *)realloc(st->pieces, cap * sizeof(*np));
    if (!np)
      return NULL;
    st->pieces    = np;
    st->cappieces = cap;
  }
  return &st->pieces[st->npieces++];
}

static uint8_t *
next_page(infl_stream_t *st, int32_t *index) {
  if (st->pages_used == st->npages) {
    uint8_t *pg;
    if (st->npages == st->cappages) {
      size_t    cap = st->cappages ? st->cappages * 2 : 8;
      uint8_t **np  = (uint8_t **)realloc(st->pages, cap * sizeof(*np));
      if (!np)
        return NULL;
      st->pages    = np;
      st->cappages = cap;
    }
    pg = page_acquire();
    if (!pg)
      return NULL;
    st->pages[st->npages++] = pg;
  }
  *index = (int32_t)st->pages_used;
  return st->pages[st->pages_used++];
}

UNZ_EXPORT void
infl_include(infl_stream_t * __restrict st, const void * __restrict ptr, uint32_t len) {
  piece_t *pc;

  if (!st)
    return;

  /* room left in the page being filled: append, whatever the size (mem.c:121-132) */
  if (st->appending && len <= PAGE_SIZE - st->page_fill) {
    pc = &st->pieces[st->npieces - 1];
    if (len)
      memcpy(st->pages[pc->page] + st->page_fill, ptr, len);
    st->page_fill += len;
    pc->len       += len;
    st->srclen    += len;
    return;
  }

  if (len <= APPEND_THRESHOLD) {                 /* small: copy into a fresh page */
    int32_t  idx;
    uint8_t *pg = next_page(st, &idx);
    if (pg && (pc = new_piece(st))) {
      if (len)
        memcpy(pg, ptr, len);
      pc->p         = pg;
      pc->len       = len;
      pc->page      = idx;
      st->page_fill = len;
      st->appending = 1;
      st->srclen   += len;
      return;
    }
    if (pg)
      st->pages_used--;
    /* out of memory: like the reference, fall through to borrowing */
  }

  pc = new_piece(st);                             /* large: borrow (mem.c:185-207) */
  if (!pc)
    return;                                       /* the reference drops it too (mem.c:191) */
  pc->p         = (const uint8_t *)ptr;
  pc->len       = len;
  pc->page      = -1;
  st->appending = 0;
  st->srclen   += len;
}

/* Contiguous view of everything included so far; *out points either at the
   single piece or at the stream's gather buffer. */
static int
gather_input(infl_stream_t *st, const uint8_t **out) {
  size_t at = 0;

  if (st->npieces == 1) {
    *out = st->pieces[0].p;
    return 0;
  }
  if (st->srclen > st->gather_cap) {
    size_t   cap = st->srclen + st->srclen / 2 + 64;
    uint8_t *nb  = (uint8_t *)realloc(st->gather, cap);
    if (!nb)
      return -1;
    st->gather     = nb;
    st->gather_cap = cap;
  }
  for (size_t i = 0; i < st->npieces; i++) {
    if (st->pieces[i].len)
      memcpy(st->gather + at, st->pieces[i].p, st->pieces[i].len);
    at += st->pieces[i].len;
  }
  *out = st->gather;
  return 0;
}

static int
decode_one(infl_stream_t *st, int flags, int32_t *detail) {
  const uint8_t *src;
  uint64_t       soff[2], doff[2];
  int32_t        status = UNZ_ERR;
  uint32_t       outlen = 0, inpos = 0;
  int            rc;

  if (st->srclen > UINT32_MAX || gather_input(st, &src) != 0)
    return UNZ_ERR;
  soff[0] = 0; soff[1] = st->srclen;
  doff[0] = 0; doff[1] = st->dstlen;
  rc = infl_rt_batch_host(src, soff, st->dst, doff, NULL, flags, 1, &status, &outlen, detail,
                          &inpos);
  if (rc != 0) {
    if (rc == -1)
      warn_no_device();
    return UNZ_ERR;
  }
  st->dstpos = outlen;
  st->inpos  = inpos;
  return status;
}

UNZ_EXPORT int
infl(infl_stream_t * __restrict st) {
  int32_t detail = 0;

  if (!st)
    return UNZ_ERR;
  if (st->npieces == 0)
    return UNZ_NOOP;                              /* nothing included: infl.c:1544-1545 */
  return decode_one(st, st->flags == INFL_ZLIB ? INFL_ZLIB : 0, &detail);
}

static int
acc_append(infl_stream_t *st, const void *p, size_t len) {
  if (st->acc_len + len > st->acc_cap) {
    size_t   cap = st->acc_cap ? st->acc_cap * 2 : 4096;
    uint8_t *nb;
    while (cap < st->acc_len + len)
      cap *= 2;
    nb = (uint8_t *)realloc(st->acc, cap);
    if (!nb)
      return -1;
    st->acc     = nb;
    st->acc_cap = cap;
  }
  if (len)
    memcpy(st->acc + st->acc_len, p, len);
  st->acc_len += len;
  return 0;
}

/* Incremental decode (reference src/infl/stream.c:468-794): every call adds its bytes to the input
   with infl_include(), like the reference (stream.c:481-482), so infl_include() and infl_stream()
   may be mixed on one object and a later infl() sees the same input.  The decode goes on from
   where the last call stopped (a block header or a symbol boundary; the GPU keeps input, output
   and position, see infl_resume_kernel); `acc` is the piece list as one contiguous range, extended
   by whatever the list has gained since the last call.  UNZ_UNFINISHED while the stream wants more
   input, UNZ_EFULL when the output buffer is full (go on after infl_resize_output), UNZ_OK once
   the final block is complete, UNZ_NOOP for a call without data before any decode has begun
   (stream.c:505-507). */
UNZ_EXPORT int
infl_stream(infl_stream_t * __restrict st, const void * __restrict src, uint32_t srclen) {
  InflResume rs;
  int        rc;

  if (!st)
    return UNZ_ERR;
  if (src && srclen > 0)
    infl_include(st, src, srclen);
  else if (st->npieces == 0)
    return UNZ_OK;                                /* "empty data", stream.c:489-493 */
  if (st->stream_done)
    return UNZ_OK;
  if (!src && srclen == 0 && !st->stream_started)
    return UNZ_NOOP;
  while (st->mir_piece < st->npieces) {           /* what the piece list has gained (its last piece may have grown) */
    const piece_t *pc = &st->pieces[st->mir_piece];
    if (pc->len > st->mir_off) {
      if (acc_append(st, pc->p + st->mir_off, pc->len - st->mir_off) != 0)
        return UNZ_ENOMEM;
      st->mir_off = pc->len;
    }
    if (st->mir_piece + 1 == st->npieces)
      break;
    st->mir_piece++;
    st->mir_off = 0;
  }
  st->stream_started = 1;
  if (st->acc_len > UINT32_MAX)
    return UNZ_ERR;

  rc = infl_rt_stream_step(&st->sctx, st->acc, st->acc_len, st->dst, st->dstlen, st->dstpos,
                           st->flags == INFL_ZLIB, &rs);
  if (rc != 0) {
    if (rc == -1)
      warn_no_device();
    return rc == -12 ? UNZ_ENOMEM : UNZ_ERR;
  }
  st->dstpos = rs.pos;
  st->inpos  = (uint32_t)((rs.bit + 7u) >> 3);
  if (rs.flags & INFL_RS_DONE) {
    st->stream_done = 1;
    return UNZ_OK;
  }
  switch (rs.detail) {
    case INFL_D_NEED_INPUT: return UNZ_UNFINISHED;
    case INFL_D_FULL:       return UNZ_EFULL;
    default:                return UNZ_ERR;
  }
}

/* chunk-list gather of streams [k0, k1) of a pass into its source arena */
typedef struct gather_job_t {
  infl_stream_t * const *streams;
  const size_t          *idx;
  const uint64_t        *soff;
  uint8_t               *src;
  size_t                 k0, k1;
} gather_job_t;

static void *
gather_run(void *arg) {
  const gather_job_t *g = (const gather_job_t *)arg;
  for (size_t k = g->k0; k < g->k1; k++) {
    const infl_stream_t *st = g->streams[g->idx[k]];
    size_t               at = g->soff[k];
    for (size_t j = 0; j < st->npieces; j++) {
      if (st->pieces[j].len)
        memcpy(g->src + at, st->pieces[j].p, st->pieces[j].len);
      at += st->pieces[j].len;
    }
  }
  return NULL;
}

/* one launch over the streams of `streams` whose zlib-ness equals `zlib` */
static int
many_pass(infl_stream_t * const *streams, size_t n, int zlib, int32_t *status) {
  uint64_t *soff, *doff;
  uint32_t *outlen, *inpos;
  int32_t  *st_tmp;
  size_t   *idx;
  void    **dptr;
  uint8_t  *src = NULL, *pinned = NULL;
  size_t    m = 0, stotal = 0, dtotal = 0;
  int       rc = 0;

  for (size_t i = 0; i < n; i++)
    if (streams[i] && streams[i]->npieces && (streams[i]->flags == INFL_ZLIB) == zlib)
      m++;
  if (m == 0)
    return 0;

  soff   = (uint64_t *)malloc((m + 1) * sizeof(*soff));
  doff   = (uint64_t *)malloc((m + 1) * sizeof(*doff));
  outlen = (uint32_t *)calloc(m, sizeof(*outlen));
  inpos  = (uint32_t *)calloc(m, sizeof(*inpos));
  st_tmp = (int32_t *)calloc(m, sizeof(*st_tmp));
  idx    = (size_t *)malloc(m * sizeof(*idx));
  dptr   = (void **)malloc(m * sizeof(*dptr));
  if (!soff || !doff || !outlen || !inpos || !st_tmp || !idx || !dptr) {
    rc = UNZ_ENOMEM;
    goto out;
  }
  m = 0;
  for (size_t i = 0; i < n; i++) {
    if (!(streams[i] && streams[i]->npieces && (streams[i]->flags == INFL_ZLIB) == zlib))
      continue;
    idx[m]  = i;
    dptr[m] = streams[i]->dst;
    soff[m] = stotal;
    doff[m] = dtotal;
    stotal += streams[i]->srclen;
    dtotal += streams[i]->dstlen;
    m++;
  }
  soff[m] = stotal;
  doff[m] = dtotal;

  if (stotal >= (1u << 20))                       /* large gathers go to pinned memory if it is free */
    src = pinned = (uint8_t *)infl_rt_pinned_acquire(stotal);
  if (!src)
    src = (uint8_t *)malloc(stotal ? stotal : 1);
  if (!src) {
    rc = UNZ_ENOMEM;
    goto out;
  }
  {                                               /* chunk-list gather, by a few threads when it is large */
    enum { GT = 4 };
    gather_job_t job[GT];
    pthread_t    th[GT];
    int          started[GT] = {0};
    const int    nt = (stotal >= (8u << 20) && m >= 4 * GT) ? GT : 1;
    for (int t = 0; t < nt; t++) {
      job[t].streams = streams; job[t].idx = idx; job[t].soff = soff; job[t].src = src;
      job[t].k0 = m * (size_t)t / (size_t)nt;
      job[t].k1 = m * (size_t)(t + 1) / (size_t)nt;
    }
    for (int t = 1; t < nt; t++)
      started[t] = pthread_create(&th[t], NULL, gather_run, &job[t]) == 0;
    gather_run(&job[0]);
    for (int t = 1; t < nt; t++) {
      if (started[t])
        pthread_join(th[t], NULL);
      else
        gather_run(&job[t]);                      /* no thread to be had: this one does the share */
    }
  }

  rc = infl_rt_batch_host(src, soff, NULL, doff, dptr, zlib ? INFL_ZLIB : 0, m, st_tmp, outlen,
                          NULL, inpos);
  if (rc == 0) {
    for (size_t k = 0; k < m; k++) {
      streams[idx[k]]->dstpos = outlen[k];
      streams[idx[k]]->inpos  = inpos[k];
      status[idx[k]]          = st_tmp[k];
    }
  } else {
    if (rc == -1)
      warn_no_device();
    rc = rc == -12 ? UNZ_ENOMEM : UNZ_ERR;
  }
out:
  free(soff); free(doff); free(outlen); free(inpos); free(st_tmp); free(idx); free(dptr);
  if (pinned)
    infl_rt_pinned_release(pinned);
  else
    free(src);
  return rc;
}

UNZ_EXPORT int
infl_many(infl_stream_t * const *streams, size_t n, int32_t *status) {
  int rc;

  if (n == 0)
    return 0;
  if (!streams || !status)
    return UNZ_ERR;
  for (size_t i = 0; i < n; i++)                 /* nothing included: infl.c:1544-1545 */
    status[i] = (streams[i] && streams[i]->npieces) ? UNZ_ERR : UNZ_NOOP;
  rc = many_pass(streams, n, 0, status);
  if (rc == 0)
    rc = many_pass(streams, n, 1, status);
  return rc;
}
