/*
 * defl/infl_batch.h -- batched entry points of the MI355X inflate engine.
 *
 * The reference decodes one stream per call (infl(), src/infl/infl.c:1507-1675);
 * one stream cannot occupy a GPU, so the engine adds calls that decode many
 * independent streams with one kernel launch.  Each stream is decoded with the
 * exact semantics of the reference's infl() on one contiguous input: the
 * per-stream status is what infl() would return, outlen what infl_output_pos()
 * would report afterwards, and dst[dst_off[i] .. dst_off[i] + outlen[i]) the
 * bytes it would have produced.
 *
 * Plain C ABI: pointers and sizes only.  "dev" pointers are HIP device
 * pointers (hipMalloc / a framework's device allocation); hip_stream is a
 * hipStream_t passed as void* (NULL = the default stream).
 */
#ifndef infl_batch_h
#define infl_batch_h

#include "common.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Selects the HIP device used by the calling thread for all later calls and
   prepares the engine on it.  0, or UNZ_ERR if the device is unusable.  Replaces
   nothing in the reference (it has no device notion). */
UNZ_EXPORT int
infl_gpu_init(int device);

/* Number of usable HIP devices (0 if none / no driver). */
UNZ_EXPORT int
infl_gpu_device_count(void);

/*
 * Decodes n independent streams that are already resident in device memory.
 *   dev_src      compressed arena
 *   dev_src_off  n + 1 offsets into it; stream i is [off[i], off[i+1])
 *   dev_dst      output arena
 *   dev_dst_off  n + 1 offsets; stream i owns [off[i], off[i+1]) as capacity
 *   flags        0 raw DEFLATE, INFL_ZLIB zlib wrapper (all streams alike)
 *   dev_status   n result codes (UnzResult), as infl() returns them
 *   dev_outlen   n output lengths, as infl_output_pos() reports them
 * Asynchronous: the kernel is enqueued on hip_stream and the call returns.
 * Returns 0 when enqueued, UNZ_ERR on a launch failure or bad argument.
 * Per stream this replaces infl_buf() (reference include/defl/infl.h:125-145).
 */
UNZ_EXPORT int
infl_batch_dev(const void *dev_src, const uint64_t *dev_src_off,
               void *dev_dst, const uint64_t *dev_dst_off,
               int flags, size_t n,
               int32_t *dev_status, uint32_t *dev_outlen,
               void *hip_stream);

/*
 * infl_batch_dev() for batches whose streams differ widely in size and come in no particular order
 * (file systems, HTTP payloads): the streams are first put into size classes by a counting sort on
 * the device, and the decode takes them class by class, largest first, so that the wavefronts that
 * run side by side work on streams of similar length (1 KiB .. 256 KiB log-uniform in random order:
 * 203 -> 279 GB/s decompressed).  Results are those of infl_batch_dev(): they do not depend on the
 * order streams are decoded in.  dev_work: INFL_BATCH_SORT_WORK_BYTES(n) bytes of device memory
 * for the call's own use (n < 2^32).  The reference has no counterpart (one stream per call).
 */
#define INFL_BATCH_SORT_WORK_BYTES(n) (1024u + 4u * (size_t)(n))
UNZ_EXPORT int
infl_batch_dev_sorted(const void *dev_src, const uint64_t *dev_src_off,
                      void *dev_dst, const uint64_t *dev_dst_off,
                      int flags, size_t n,
                      int32_t *dev_status, uint32_t *dev_outlen,
                      void *dev_work, void *hip_stream);

/*
 * infl_batch_dev() plus one more output: dev_detail[i] says why stream i
 * stopped (0 complete, 1 malformed, 2 input ended inside the stream, 3 output
 * capacity reached, 4 internal safety stop).  The reference folds 1-3 into
 * UNZ_ERR on this path (src/infl/infl.c:1074-1087); the extra array is for
 * diagnosis and tests.
 */
UNZ_EXPORT int
infl_batch_dev_detail(const void *dev_src, const uint64_t *dev_src_off,
                      void *dev_dst, const uint64_t *dev_dst_off,
                      int flags, size_t n,
                      int32_t *dev_status, uint32_t *dev_outlen,
                      int32_t *dev_detail, void *hip_stream);

/*
 * Same for host memory: uploads the arenas, decodes, downloads the results
 * and returns when they are in place.  All pointers are host pointers.
 * Returns 0, UNZ_ENOMEM or UNZ_ERR (no device).
 */
UNZ_EXPORT int
infl_batch(const void *src, const uint64_t *src_off,
           void *dst, const uint64_t *dst_off,
           int flags, size_t n, int32_t *status, uint32_t *outlen);

/*
 * Opt-in integrity and framing.  The reference stops at the final block: it never reads the zlib
 * trailer (src/infl/zlib.h:45-96 parses the two header bytes only, src/infl/infl.c:1094-1101 hands
 * the position back) and knows no gzip, so these checks are entry points of their own and the ten
 * reference functions and the calls above keep the reference's behaviour bit for bit.
 *   check = INFL_CHECK_ADLER32  streams are zlib streams (RFC 1950); after the decode the Adler-32
 *                               of the output is compared with the four bytes behind the final block;
 *   check = INFL_CHECK_GZIP     streams are gzip members (RFC 1952): the header (with its optional
 *                               extra / name / comment / header-CRC fields) is skipped, and CRC-32
 *                               and ISIZE of the output are compared with the eight trailer bytes.
 * status[i] = UNZ_OK only if the stream decodes AND its trailer matches.  detail[i] (may be NULL)
 * tells the rest: 0 fine, 1 malformed (also: not a gzip header), 2 input ended (also: inside the
 * header or the trailer), 3 output capacity reached, 5 trailer mismatch; the output bytes of a
 * stream whose trailer does not match stay in place, outlen[i] says how many.
 * dev_work: 8 * n bytes of device memory for the call's own use.
 */
#define INFL_CHECK_ADLER32 1
#define INFL_CHECK_GZIP    2
UNZ_EXPORT int
infl_batch_dev_checked(const void *dev_src, const uint64_t *dev_src_off,
                       void *dev_dst, const uint64_t *dev_dst_off,
                       int check, size_t n,
                       int32_t *dev_status, uint32_t *dev_outlen, int32_t *dev_detail,
                       void *dev_work, void *hip_stream);

/* The same for arenas in host memory (upload, decode and check, download). */
UNZ_EXPORT int
infl_batch_checked(const void *src, const uint64_t *src_off,
                   void *dst, const uint64_t *dst_off,
                   int check, size_t n,
                   int32_t *status, uint32_t *outlen, int32_t *detail);

/*
 * PNG un-filter on the GPU: the step that follows inflate when the streams are PNG image data
 * ("undo filters", which the reference leaves to its caller: README.md:150-152; it holds no code for
 * it, the rule is the PNG specification's, section 9).  Image i is geom[i].rows scanlines of one
 * filter-type byte (0 None, 1 Sub, 2 Up, 3 Average, 4 Paeth) and geom[i].rowbytes filtered bytes at
 * dev_in + dev_in_off[i]; its rows * rowbytes rebuilt bytes go to dev_out + dev_out_off[i].  bpp =
 * bytes per complete pixel, rounded up to 1 (1, 2, 3, 4, 6 or 8 in a PNG; 1..8 accepted).
 * To run it behind infl_batch_dev() pass that call's dev_dst / dev_dst_off as dev_in / dev_in_off,
 * its dev_outlen as dev_in_len (bytes present; NULL = all of dev_in_off[i+1] - dev_in_off[i]) and
 * its dev_status as dev_in_status (NULL = none): an image whose stream failed fails here too.
 * dev_status[i] = 0, or -1 (stream failed, fewer than rows * (rowbytes + 1) bytes present, no room
 * for the output, bpp out of range, or a filter type above 4; the image's output is then
 * unspecified).  dev_out must not overlap dev_in.  The call enqueues on hip_stream and returns.
 */
#ifndef INFL_PNG_GEOM_DEFINED
#define INFL_PNG_GEOM_DEFINED
typedef struct InflPngGeom {
  uint32_t rows, rowbytes, bpp, reserved;
} InflPngGeom;
#endif
UNZ_EXPORT int
infl_png_unfilter_dev(const void *dev_in, const uint64_t *dev_in_off,
                      const uint32_t *dev_in_len, const int32_t *dev_in_status,
                      void *dev_out, const uint64_t *dev_out_off,
                      const InflPngGeom *dev_geom, size_t n,
                      int32_t *dev_status, void *hip_stream);

/*
 * infl_include() for input that already lies in device memory (a PNG file uploaded whole, its IDAT
 * payloads scattered through it): the device-side counterpart of a run of infl_include() calls, each of
 * which appends one byte range to a stream's logical input (reference src/infl/mem.c:114-216; the
 * README's PNG loop, README.md:120-155).  Piece p is the dev_piece_len[p] bytes at dev_blob +
 * dev_piece_off[p] and is copied to dev_src + dev_piece_dst[p]; the caller, who walked the chunk
 * headers, lays the pieces of a stream one behind the other and the streams one behind the other, and
 * passes dev_src with the matching dev_src_off to infl_batch_dev().  Pieces of any length (0 included)
 * at any alignment; dev_src must not overlap the pieces it is gathered from.  No bounds are known to
 * this call: offsets and lengths must lie inside the caller's buffers (a descriptor that does not is
 * a memory fault on the device; callers that take descriptors from files use the _checked form).
 * Enqueues on hip_stream and returns; 0, or -1 for a NULL argument or a launch that failed.
 */
UNZ_EXPORT int
infl_gather_dev(const void *dev_blob, const uint64_t *dev_piece_off,
                const uint32_t *dev_piece_len, const uint64_t *dev_piece_dst,
                size_t n_pieces, void *dev_src, void *hip_stream);

/*
 * The two descriptor-driven calls with the sizes of their buffers, for descriptors that come from
 * untrusted input (chunk headers of a file): blob_bytes / src_bytes are the sizes of dev_blob and
 * dev_src, in_bytes / out_bytes those of dev_in and dev_out.  A piece that does not lie inside
 * both of its buffers is not copied and gets dev_piece_status[p] = -1 (0 otherwise; the array may
 * be NULL); an image whose input or output range reaches beyond its arena gets dev_status[i] = -1
 * and is not touched.  No counterpart in the reference (mem.c:114-216 trusts its caller too).
 */
UNZ_EXPORT int
infl_gather_dev_checked(const void *dev_blob, uint64_t blob_bytes,
                        const uint64_t *dev_piece_off, const uint32_t *dev_piece_len,
                        const uint64_t *dev_piece_dst, size_t n_pieces,
                        void *dev_src, uint64_t src_bytes,
                        int32_t *dev_piece_status, void *hip_stream);
UNZ_EXPORT int
infl_png_unfilter_dev_checked(const void *dev_in, uint64_t in_bytes, const uint64_t *dev_in_off,
                              const uint32_t *dev_in_len, const int32_t *dev_in_status,
                              void *dev_out, uint64_t out_bytes, const uint64_t *dev_out_off,
                              const InflPngGeom *dev_geom, size_t n,
                              int32_t *dev_status, void *hip_stream);

/*
 * infl_batch() for callers that cannot allocate: tight serving loops, harnesses that forbid
 * hipMalloc.  It has no counterpart in the reference.  Same arguments and results as infl_batch(),
 * but the device staging (source and destination arenas, offset tables, result arrays) is a fixed
 * array inside the library's code object, one copy per GPU, held until the process ends; the call
 * allocates and frees nothing (it runs on the stream infl_gpu_init() made for the device, on the
 * null stream if that call was never made).  Limits: n <= INFL_NOALLOC_MAX_STREAMS, src_off[n]
 * <= INFL_NOALLOC_MAX_SRC, dst_off[n] <= INFL_NOALLOC_MAX_DST; beyond them the call returns
 * UNZ_EFULL and decodes nothing; offsets that decrease are refused with UNZ_ERR.
 * dst[dst_off[i] + outlen[i] .. dst_off[i + 1]) is zero afterwards (infl_batch() leaves it
 * untouched): the array keeps earlier calls' bytes, and none of them reaches a later caller.
 * Calls on the same GPU share one array: the library serialises host threads of one process on it
 * (calls that bypass that, e.g. from a signal handler, race); callers in different processes each
 * have their own copy.
 */
#define INFL_NOALLOC_MAX_STREAMS 4096u
#define INFL_NOALLOC_MAX_SRC     (4u << 20)
#define INFL_NOALLOC_MAX_DST     (16u << 20)
UNZ_EXPORT int
infl_batch_noalloc(const void *src, const uint64_t *src_off,
                   void *dst, const uint64_t *dst_off,
                   int flags, size_t n, int32_t *status, uint32_t *outlen);

/*
 * Deferred form of infl() for many streams that were filled with
 * infl_include(): gathers every stream's chunk list into one staging arena,
 * decodes all of them with one launch and stores each stream's result as
 * infl() would have (status[i] is its return value; output position and bytes
 * land in the stream's own destination).  Replaces a loop of infl() calls
 * (reference src/infl/infl.c:1507-1675) over streams fed by infl_include()
 * (src/infl/mem.c:114-216).
 */
UNZ_EXPORT int
infl_many(infl_stream_t * const *streams, size_t n, int32_t *status);

/* Device-memory helpers so that a plain C caller (or ctypes) can stage data
   without linking HIP itself.  Thin wrappers over hipMalloc / hipFree /
   hipMemcpy / hipDeviceSynchronize; 0 on success. */
UNZ_EXPORT void *infl_gpu_malloc(size_t bytes);
UNZ_EXPORT void  infl_gpu_free(void *dev_ptr);
UNZ_EXPORT int   infl_gpu_upload(void *dev_dst, const void *host_src, size_t bytes);
UNZ_EXPORT int   infl_gpu_download(void *host_dst, const void *dev_src, size_t bytes);
UNZ_EXPORT int   infl_gpu_sync(void);

/* Tuning knob for measurements: lanes of a wavefront that cooperate on one
   stream in the serial passes (64, 32, 16 or 8).  Returns the value in effect. */
UNZ_EXPORT int
infl_gpu_set_lanes_per_stream(int lanes);

/* Tuning knob for measurements.  Results are identical for every value.
   0 = one launch, symbols decoded one after the other per stream, decode
       tables in the reference's geometry (32-bit entries, 10 KiB of LDS per
       stream);
   1 = a first launch with compact 16-bit tables (2.7 KiB per stream, four
       times the streams in flight) plus a second launch as 0 for the streams
       the first one could not finish;
   2 = (default) a first launch in which a whole wavefront decodes each
       Huffman block in parallel (64 speculative sub-sequences that
       re-synchronise, matches resolved in LDS; stretches of very long matches
       symbol by symbol), then the two launches of 1 for the streams it hands
       over: anything malformed or out of space, tables too large for the
       16-bit format, streams of fewer than 8 bytes.
   Returns the value in effect. */
UNZ_EXPORT int
infl_gpu_set_table_variant(int variant);

#ifdef __cplusplus
}
#endif
#endif /* infl_batch_h */
