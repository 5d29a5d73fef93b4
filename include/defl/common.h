/*
 * defl/common.h -- shared declarations of the MI355X inflate engine.
 *
 * Binary- and source-compatible with the public header of recp/defl
 * (reference: include/defl/common.h:50-99): same result codes, same opaque
 * stream type names, same INFL_ZLIB flag, so a caller of the reference
 * recompiles against this header unchanged.
 */
#ifndef defl_common_h
#define defl_common_h

#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>
#include <errno.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_MSC_VER)
#  define UNZ_EXPORT   __declspec(dllexport)
#  define UNZ_INLINE   __forceinline
#else
#  define UNZ_EXPORT   __attribute__((visibility("default")))
#  define UNZ_INLINE   static inline __attribute__((always_inline))
#endif

/* Result codes; numeric values as in the reference (include/defl/common.h:80-90). */
typedef enum UnzResult {
  UNZ_UNFINISHED =  2,        /* infl_stream(): feed more input             */
  UNZ_NOOP       =  1,        /* nothing was included, nothing to do        */
  UNZ_OK         =  0,
  UNZ_ERR        = -1,        /* malformed, truncated or over-long stream   */
  UNZ_EFOUND     = -1000,
  UNZ_ENOMEM     = -ENOMEM,
  UNZ_EPERM      = -EPERM,
  UNZ_EBADF      = -EBADF,
  UNZ_EFULL      = -ENOBUFS   /* infl_stream(): output capacity reached     */
} UnzResult;

/* Opaque stream; the layout is private to the library (reference :94-96). */
typedef struct unz__stream_t defl_stream_t;
typedef struct unz__stream_t infl_stream_t;
typedef struct unz__chunk_t  defl_chunk_t;

/* flags of infl_init()/infl_reset()/infl_buf(): 0 = raw DEFLATE, 1 = zlib wrapper */
#define INFL_ZLIB 1

#ifdef __cplusplus
}
#endif
#endif /* defl_common_h */
