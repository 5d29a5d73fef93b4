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
#define INFL_D_CHECKSUM    5   /* trailer mismatch (checked entry points)   */

/* integrity passes of infl_launch_checked (values of infl_batch.h's INFL_CHECK_*) */
#define INFL_CHECK_ADLER32 1
#define INFL_CHECK_GZIP    2
/* header lengths the gzip header pass leaves for streams without a usable header */
#define INFL_SKIP_SHORT 0xfffffffeu   /* input ends inside the header */
#define INFL_SKIP_BAD   0xffffffffu   /* not a gzip member            */

/* status value the first (compact-table) pass leaves for a stream that the
   second pass has to decode; never visible to callers */
#define INFL_STATUS_DEFERRED 0x40000000
/* same, for a stream the parallel pass found to fail: only the last pass (tables in the
   reference's geometry, exact failure result) needs to look at it */
#define INFL_STATUS_DEFERRED_FAILING 0x40000001
/* same, and the parallel pass has left where the last pass can take over: outlen[i] = output position, detail[i] =
   bit position of the block's header, inpos[i] = bit position of a symbol of that block (0 = its first) */
#define INFL_STATUS_DEFERRED_RESUME 0x40000002
// decoded by the windowed serial pass; the last pass turns it into 0 (until then the mark tells the window's other
// wavefronts that the stream was one of the marked ones)
#define INFL_STATUS_DEFERRED_DONE 0x40000003

/* kernel flag bits: bit 0 = zlib wrapper; INFL_F_PROGRESS = when a stream stops
   for lack of input or output space, report the output position it reached in
   `outlen` (the one-shot API must report 0 there, like the reference) */
#define INFL_F_ZLIB      0x001
#define INFL_F_PROGRESS  0x100

/* What infl_stream() keeps between calls (device side): the bit position of the symbol to go on
   with, the bit position of the header of the block it belongs to, the output position. */
#define INFL_RS_HEADER   1u    /* the zlib wrapper (if any) has been read            */
#define INFL_RS_INBLOCK  2u    /* `bit` lies inside a Huffman block, behind its header */
#define INFL_RS_DONE     4u    /* the final block is complete                          */
typedef struct InflResume {
  uint64_t bit, block_bit;
  uint32_t pos, flags;
  int32_t  detail;              /* INFL_D_* of the last call */
  uint32_t reserved;            /* the output position before the last call */
  uint64_t tables_bit;          /* 1 + bit position of the block header whose decode tables the call left in its
                                   table buffer (0 = none) */
} InflResume;
/* size of the table buffer of infl_launch_resume: both decode tables in the reference's geometry */
#define INFL_RESUME_TABLE_BYTES ((2048u + 512u) * 4u)

/* Enqueues the one-stream incremental decode: src[0 .. srclen) is all the input seen so far,
   dst[0 .. cap) the output so far and the room behind it, *state (device memory) where the last
   call stopped and where this one does.  fresh (may be NULL): host memory the device can read,
   holding input bytes fresh_at .. srclen, which the kernel moves into src first.  hout / hstate
   (may be NULL): host memory the device can write; the call's new output bytes (at most hout_cap)
   and a copy of the state go there.  The state's `reserved` word is the output position before
   the call.  tables (may be NULL): INFL_RESUME_TABLE_BYTES of device memory that belong to the
   stream; a call that stops inside a Huffman block leaves the block's decode tables there and the
   next call takes them up instead of building them again from the block's header.  0 or -1. */
int infl_launch_resume(void *src, uint32_t srclen, void *dst, uint32_t cap, int zlib,
                       InflResume *state, const void *fresh, uint32_t fresh_at, void *hout,
                       uint32_t hout_cap, InflResume *hstate, void *tables, void *stream);

/* Enqueues the batch kernel; all pointers are device pointers; `detail` (why a
   stream stopped) and `inpos` (input bytes consumed) may be NULL.  lanes =
   lanes of a wavefront per stream (64/32/16/8).  `skip` (may be NULL): bytes in front of every
   stream's DEFLATE data (a container header), inpos then counts from behind them.  variant 0 = one launch with
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
                      int variant, void *mirror, void *stream, const uint32_t *skip);

/* infl_launch_batch with the order in which the parallel pass's workgroups take the streams (n indices in
   device memory, may be NULL), as infl_launch_order leaves it at (uint32_t *)work + 256. */
int infl_launch_batch_ordered(const void *src, const uint64_t *src_off, void *dst,
                              const uint64_t *dst_off, int flags, uint64_t n, int32_t *status,
                              uint32_t *outlen, int32_t *detail, uint32_t *inpos, int lanes,
                              int variant, void *mirror, void *stream, const uint32_t *skip,
                              const uint32_t *order);
#define INFL_ORDER_WORK_BYTES(n) (1024u + 4u * (size_t)(n))
int infl_launch_order(const uint64_t *src_off, const uint64_t *dst_off, uint64_t n, void *work, void *stream);

/* infl_launch_batch plus the integrity pass (see infl_batch_dev_checked in infl_batch.h);
   work = 8 * n bytes of device memory. */
int infl_launch_checked(const void *src, const uint64_t *src_off, void *dst, const uint64_t *dst_off,
                        int check, uint64_t n, int32_t *status, uint32_t *outlen, int32_t *detail,
                        void *work, int lanes, int variant, void *mirror, void *stream);

/* Enqueues the PNG un-filter pass (see infl_png_unfilter_dev in infl_batch.h, which holds the same
   definition of InflPngGeom). */
#ifndef INFL_PNG_GEOM_DEFINED
#define INFL_PNG_GEOM_DEFINED
typedef struct InflPngGeom {
  uint32_t rows, rowbytes, bpp, reserved;
} InflPngGeom;
#endif
/* in_bytes / out_bytes: sizes of the two arenas where the caller knows them (else ~0): an image whose
   ranges reach beyond them fails with status -1 and is not touched. */
int infl_launch_png_unfilter(const void *in, const uint64_t *in_off, const uint32_t *in_len,
                             const int32_t *in_status, void *out, const uint64_t *out_off,
                             const InflPngGeom *geom, uint64_t n, int32_t *status, void *stream,
                             uint64_t in_bytes, uint64_t out_bytes);

/* blob_bytes / out_bytes as above; piece_status (may be NULL) gets 0 or -1 per piece. */
int infl_launch_gather(const void *blob, const uint64_t *piece_off, const uint32_t *piece_len,
                       const uint64_t *piece_dst, uint64_t n_pieces, void *out, void *stream,
                       uint64_t blob_bytes, uint64_t out_bytes, int32_t *piece_status);

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

/* One step of infl_stream(): *ctxp is the stream's device context (NULL before the first call,
   set by the call, released with infl_rt_stream_free); acc[0 .. acc_len) is all the input fed so
   far, dst/dstlen the caller's output buffer, dstpos the output position before the call.  The new
   output bytes are copied to dst and *out says where the decode stands.  0, -12 or -1. */
int  infl_rt_stream_step(void **ctxp, const void *acc, size_t acc_len, void *dst, uint32_t dstlen,
                         uint32_t dstpos, int zlib, InflResume *out);
void infl_rt_stream_free(void *ctx);
/* The stream starts over with its input (infl_reset_pool): the context keeps its device buffers and the next step
   begins at the first bit again. */
void infl_rt_stream_restart(void *ctx);

#ifdef __cplusplus
}
#endif
#endif
