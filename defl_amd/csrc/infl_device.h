/*
 * infl_device.h -- private interface between the C host side (infl_host.c),
 * the HIP runtime glue (infl_gpu.hip) and the kernels (infl_kernels.hip).
 * Not installed; the public C ABI is include/defl/infl.h and infl_batch.h.
 */
#ifndef INFL_DEVICE_H
#define INFL_DEVICE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Why a stream stopped.  The one-shot API reports every non-OK value as
   UNZ_ERR, like the reference (src/infl/infl.c:1074-1087); the incremental API
   needs the distinction. */
#define INFL_D_OK          0
#define INFL_D_INVALID     1   /* malformed stream                          */
#define INFL_D_NEED_INPUT  2   /* input ended inside the stream             */
#define INFL_D_FULL        3   /* output capacity reached                   */
#define INFL_D_INTERNAL    4   /* safety counter ran out (engine bug)       */

/* status value the first (compact-table) pass leaves for a stream that the
   second pass has to decode; never visible to callers */
#define INFL_STATUS_DEFERRED 0x40000000
/* same, for a stream the parallel pass found to fail: only the last pass (tables in the
   reference's geometry, exact failure result) needs to look at it */
#define INFL_STATUS_DEFERRED_FAILING 0x40000001

/* kernel flag bits: bit 0 = zlib wrapper; INFL_F_PROGRESS = when a stream stops
   for lack of input or output space, report the output position it reached in
   `outlen` (the one-shot API must report 0 there, like the reference) */
#define INFL_F_ZLIB      0x001
#define INFL_F_PROGRESS  0x100

/* Enqueues the batch kernel; all pointers are device pointers; `detail` (why a
   stream stopped) and `inpos` (input bytes consumed) may be NULL.  lanes =
   lanes of a wavefront per stream (64/32/16/8).  variant 0 = one launch with
   tables in the reference's geometry; variant 1 = a first launch with compact
   16-bit tables (more streams in flight) plus a second launch in the
   reference's geometry for the streams the first one deferred; variant 2 = a
   wavefront-parallel launch first, then the two of variant 1 for what it
   deferred.  `mirror` (may be NULL) is a second arena laid out like dst,
   typically pinned host memory: every stream's output bytes are copied to the
   same offsets there by the kernel that completes the stream.  0 or -1. */
int infl_launch_batch(const void *src, const uint64_t *src_off, void *dst,
                      const uint64_t *dst_off, int flags, uint64_t n, int32_t *status,
                      uint32_t *outlen, int32_t *detail, uint32_t *inpos, int lanes,
                      int variant, void *mirror, void *stream);

/* A pinned host buffer for gathering many streams' input before infl_rt_batch_host() (copies
   from pinned memory run at the full link rate).  One buffer per process: returns NULL when it is
   in use, when no device is usable or when pinning fails; the caller then uses ordinary memory.
   Release after the batch call. */
void *infl_rt_pinned_acquire(size_t bytes);
void  infl_rt_pinned_release(void *p);

/* Host-memory batch; used by infl(), infl_many() and infl_stream().  Stream i
   reads src[src_off[i] .. src_off[i+1]) and has dst_off[i+1] - dst_off[i] bytes
   of output capacity; its output goes to dst_ptrs[i] if dst_ptrs is given, else
   to dst + dst_off[i].  detail and inpos may be NULL.
   0, -12 (no memory) or -1 (no device / launch failure). */
int infl_rt_batch_host(const void *src, const uint64_t *src_off, void *dst,
                       const uint64_t *dst_off, void * const *dst_ptrs, int flags, size_t n,
                       int32_t *status, uint32_t *outlen, int32_t *detail, uint32_t *inpos);

#ifdef __cplusplus
}
#endif
#endif
