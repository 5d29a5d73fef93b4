/*
 * infl_oracle.h -- interface of the CPU restatement in infl_oracle.c.
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker, never by the shipped library.
 */
#ifndef INFL_ORACLE_H
#define INFL_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* return codes of the reference (include/defl/common.h:80-90) */
#define ORACLE_OK          0
#define ORACLE_NOOP        1
#define ORACLE_UNFINISHED  2
#define ORACLE_ERR        (-1)
#define ORACLE_EFULL      (-105)

#define ORACLE_FLAG_ZLIB   1     /* INFL_ZLIB, include/defl/common.h:99 */

/* why a stream failed, before infl() flattens it to ORACLE_ERR */
#define ORACLE_D_OK          0
#define ORACLE_D_INVALID     1   /* malformed stream                     */
#define ORACLE_D_NEED_INPUT  2   /* input ended inside the stream        */
#define ORACLE_D_FULL        3   /* output capacity reached              */

int oracle_infl(const uint8_t *src, uint32_t srclen, uint8_t *dst, uint32_t dstlen,
                int flags, uint32_t *dstpos_out);

int oracle_infl_detail(const uint8_t *src, uint32_t srclen, uint8_t *dst,
                       uint32_t dstlen, int flags, uint32_t *dstpos_out,
                       int *detail_out);

/* stream i reads src[src_off[i] .. src_off[i+1]) and owns dst[dst_off[i] ..
   dst_off[i+1]) as its output capacity */
int oracle_infl_batch(const uint8_t *src, const uint64_t *src_off, uint8_t *dst,
                      const uint64_t *dst_off, int flags, size_t n, int32_t *status,
                      uint32_t *outlen, int nthreads);

/* The same decode with, around every stream, the allocations and the copy the reference's infl_buf() makes when
   it is called once per stream as it stands (include/defl/infl.h:125-145); the plain driver above corresponds to
   the reference's other pattern, one stream object reset between inputs (src/infl/mem.c:257-275).  For timing. */
int oracle_infl_batch_as_is(const uint8_t *src, const uint64_t *src_off, uint8_t *dst,
                            const uint64_t *dst_off, int flags, size_t n, int32_t *status,
                            uint32_t *outlen, int nthreads);

/* observer called for every match decoded (NULL = off) */
void oracle_set_match_hook(void (*hook)(unsigned pos, unsigned len, unsigned dist));

/* observer called with the code lengths of every dynamic block header decoded (NULL = off) */
void oracle_set_header_hook(void (*hook)(const uint8_t *lens, unsigned hlit, unsigned hdist));

int oracle_table_build(uint32_t *table, unsigned *used_out, const uint8_t *lens,
                       unsigned nsyms, unsigned tablebits, unsigned cap, int litlen);

/* PNG un-filter (png_unfilter.c): rows x (1 + rowbytes) filtered bytes -> rows x rowbytes bytes.
   0, or -1 when a row names a filter type above 4. */
int oracle_png_unfilter(const uint8_t *in, uint8_t *out, uint32_t rows, uint32_t rowbytes, uint32_t bpp);

#ifdef __cplusplus
}
#endif
#endif
