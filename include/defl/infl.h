/*
 * defl/infl.h -- inflate API of the MI355X engine; drop-in for recp/defl's
 * include/defl/infl.h.  Every function below replaces the reference function
 * of the same name (file:line given per declaration) and keeps its signature,
 * argument meaning, ownership rules and return codes.  Decoding runs on the
 * GPU (HIP kernels for gfx950); there is no CPU decoder behind these calls and
 * they return UNZ_ERR when no usable device is present.
 *
 * Unlike the reference header, all prototypes are inside an extern "C" block,
 * so C++ callers get C linkage without wrapping the include themselves.
 */
#ifndef infl_h
#define infl_h

#include "common.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Creates a stream that inflates into dst[0..dstlen).  flags: 0 or INFL_ZLIB.
   Returns NULL when out of memory.            (reference include/defl/infl.h:31-33,
   src/infl/infl.c:1677-1695) */
UNZ_EXPORT infl_stream_t *
infl_init(void * __restrict dst, uint32_t dstlen, int flags);

/* Re-arms a stream for another input and destination, keeping its pooled
   buffers.                                    (reference infl.h:43-48, src/infl/mem.c:257-275) */
UNZ_EXPORT void
infl_reset(infl_stream_t * __restrict stream, void * __restrict dst,
           uint32_t dstlen, int flags);

/* Swaps the destination after infl_stream() reported UNZ_EFULL; the caller has
   preserved the bytes produced so far.  UNZ_ERR if dstlen is below the current
   output position.                            (reference infl.h:60-64, mem.c:277-288) */
UNZ_EXPORT int
infl_resize_output(infl_stream_t * __restrict stream, void * __restrict dst,
                   uint32_t dstlen);

/* Appends one piece of compressed input.  Pieces of at most 8 KiB are copied
   into the stream's own pages; larger pieces are borrowed and must stay valid
   until infl() returns.                       (reference infl.h:78-82, mem.c:114-216) */
UNZ_EXPORT void
infl_include(infl_stream_t * __restrict stream, const void * __restrict ptr,
             uint32_t len);

/* Inflates everything included so far in one shot.  UNZ_OK, UNZ_ERR, or
   UNZ_NOOP when nothing was included.         (reference infl.h:92-94, infl.c:1507-1675) */
UNZ_EXPORT int
infl(infl_stream_t * __restrict stream);

/* Forgets the included input but keeps the pooled pages for reuse.
                                               (reference infl.h:102-104, mem.c:219-232) */
UNZ_EXPORT void
infl_reset_pool(infl_stream_t * __restrict stream);

/* Frees the stream; NULL is accepted.         (reference infl.h:111-113, mem.c:331-361) */
UNZ_EXPORT void
infl_destroy(infl_stream_t * __restrict stream);

/* Incremental inflate: appends src[0..srclen) and decodes as far as the input
   allows.  UNZ_OK when the final block is complete, UNZ_UNFINISHED when more
   input is needed, UNZ_EFULL when dst is full (grow it, call
   infl_resize_output(), call again), UNZ_ERR on a malformed stream.
                                               (reference infl.h:164-168, src/infl/stream.c:468-794) */
UNZ_EXPORT int
infl_stream(infl_stream_t * __restrict stream, const void * __restrict src,
            uint32_t srclen);

/* Bytes produced so far.                      (reference infl.h:175-177, mem.c:290-294) */
UNZ_EXPORT uint32_t
infl_output_pos(const infl_stream_t * __restrict stream);

/* Compressed bytes consumed so far (a partly used byte counts as consumed).
                                               (reference infl.h:187-189, mem.c:296-329) */
UNZ_EXPORT uint32_t
infl_input_pos(const infl_stream_t * __restrict stream);

/* One-call convenience wrapper, compiled into the caller exactly as in the
   reference (infl.h:125-145): init, include, infl, destroy.  -1 if the stream
   cannot be allocated. */
UNZ_INLINE int
infl_buf(const void * __restrict src, uint32_t srclen, void * __restrict dst,
         uint32_t dstlen, int flags) {
  infl_stream_t *s = infl_init(dst, dstlen, flags);
  int            rc;

  if (!s)
    return -1;
  infl_include(s, src, srclen);
  rc = infl(s);
  infl_destroy(s);
  return rc;
}

#ifdef __cplusplus
}
#endif
#endif /* infl_h */
