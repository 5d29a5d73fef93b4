/*
 * dropin_check.c -- a plain C caller of <defl/infl.h>, linked against libdefl_amd.so instead of
 * the reference's libdefl: what a user of recp/defl does when switching engines.  Test
 * infrastructure.  It walks the golden file pairs the way the reference's own test program does
 * (test/test_files.c:174-257 one-shot, :260-331 input fed in 1/2/4/8-byte pieces, :688-790
 * streaming in 64..1024-byte feeds, :460-544 error vectors, :618-686 regression vectors), with
 * the same calls and the same pass criteria: return code and memcmp against the raw twin.
 *
 *   dropin_check <dir with raw/ and compressed/> [max streaming files]
 *
 * Prints one line per group and "dropin ok" at the end; exit code 1 on the first mismatch.
 */
#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <defl/infl.h>
#include <defl/infl_batch.h>

static unsigned char *slurp(const char *dir, const char *kind, const char *name, size_t *len) {
  char path[1024];
  snprintf(path, sizeof path, "%s/%s/%s", dir, kind, name);
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char *p = malloc(n > 0 ? (size_t)n : 1);
  if (n > 0 && fread(p, 1, (size_t)n, f) != (size_t)n) {
    fclose(f);
    free(p);
    return NULL;
  }
  fclose(f);
  *len = (size_t)n;
  return p;
}

static int fail(const char *what, const char *name, int rc) {
  fprintf(stderr, "FAILED %s: %s (rc %d)\n", what, name, rc);
  return 1;
}

static int cmp_names(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const char *dir = argv[1];
  int max_stream  = argc > 2 ? atoi(argv[2]) : 10;
  if (infl_gpu_device_count() < 1 || infl_gpu_init(0) != 0) {
    fprintf(stderr, "no usable HIP device\n");
    return 3;
  }

  char  cdir[1024], *names[1024];
  int   n = 0;
  snprintf(cdir, sizeof cdir, "%s/compressed", dir);
  DIR *d = opendir(cdir);
  if (!d) return 2;
  for (struct dirent *e; (e = readdir(d)) && n < 1024;)
    if (e->d_name[0] != '.') names[n++] = strdup(e->d_name);
  closedir(d);
  qsort(names, (size_t)n, sizeof names[0], cmp_names);

  /* 1. every pair, one infl_include + infl (raw DEFLATE, flags 0), and the same through infl_buf */
  int pairs = 0;
  for (int i = 0; i < n; i++) {
    size_t rl = 0, cl = 0;
    unsigned char *raw = slurp(dir, "raw", names[i], &rl), *comp = slurp(dir, "compressed", names[i], &cl);
    if (!raw || !comp) {                         /* no raw twin: skipped, like the reference does */
      free(raw);
      free(comp);
      continue;
    }
    unsigned char *out = calloc(1, rl + 1000);
    infl_stream_t *st  = infl_init(out, (uint32_t)rl + 1000, 0);
    if (!st) return fail("init", names[i], 0);
    infl_include(st, comp, (uint32_t)cl);
    int rc = infl(st);
    if (rc != UNZ_OK || memcmp(raw, out, rl) != 0 || infl_output_pos(st) != rl) return fail("one-shot", names[i], rc);
    infl_destroy(st);
    memset(out, 0, rl + 1000);
    rc = infl_buf(comp, (uint32_t)cl, out, (uint32_t)rl + 1000, 0);
    if (rc != UNZ_OK || memcmp(raw, out, rl) != 0) return fail("infl_buf", names[i], rc);
    pairs++;

    /* 2. input in 1, 2, 4 and 8-byte pieces (PNG IDAT simulation), files up to 40 KB */
    if (cl <= 40000) {
      for (uint32_t piece = 1; piece <= 8; piece *= 2) {
        memset(out, 0, rl + 1000);
        st = infl_init(out, (uint32_t)rl + 1000, 0);
        for (size_t p = 0; p < cl; p += piece) infl_include(st, comp + p, (uint32_t)(cl - p < piece ? cl - p : piece));
        rc = infl(st);
        if (cl == 0 ? (rc != UNZ_OK && rc != UNZ_NOOP) : (rc != UNZ_OK || memcmp(raw, out, rl) != 0))
          return fail("pieces", names[i], rc);
        infl_destroy(st);
      }
    }
    free(out);
    free(raw);
    free(comp);
  }
  printf("%d file pairs: one-shot, infl_buf and 1/2/4/8-byte pieces equal to the raw files\n", pairs);

  /* 3. streaming in 64/128/256/512/1024-byte feeds */
  int streamed = 0;
  for (int i = 0; i < n && streamed < max_stream; i++) {
    size_t rl = 0, cl = 0;
    unsigned char *raw = slurp(dir, "raw", names[i], &rl), *comp = slurp(dir, "compressed", names[i], &cl);
    if (!raw || !comp || cl < 200 || cl > 20000) {
      free(raw);
      free(comp);
      continue;
    }
    unsigned char *out = calloc(1, rl + 1000);
    infl_stream_t *st  = infl_init(out, (uint32_t)rl + 1000, 0);
    size_t pos = 0;
    int    rc = UNZ_UNFINISHED, k = 0;
    while (pos < cl) {
      size_t feed = (size_t)64 << (k++ % 5);
      if (pos + feed > cl) feed = cl - pos;
      rc = infl_stream(st, comp + pos, (uint32_t)feed);
      pos += feed;
      if (rc != UNZ_UNFINISHED) break;
    }
    for (int t = 0; rc == UNZ_UNFINISHED && t < 5; t++) rc = infl_stream(st, NULL, 0);
    if (rc != UNZ_OK || memcmp(raw, out, rl) != 0) return fail("streaming", names[i], rc);
    infl_destroy(st);
    free(out);
    free(raw);
    free(comp);
    streamed++;
  }
  printf("%d files streamed in 64..1024-byte feeds\n", streamed);

  /* 4. error vectors: each must not return UNZ_OK */
  {
    unsigned char out[64];
    /* (the bytes of tests/golden/kats.json, i.e. of the reference's test/test_files.c:468-476) */
    static const unsigned char bad_btype[] = {0x07};
    static const unsigned char stored10[]  = {0x01, 0x0a, 0x00, 0xf5, 0xff, 0x41, 0x42, 0x43, 0x44, 0x45,
                                              0x46, 0x47, 0x48, 0x49, 0x4a};
    static const unsigned char hdr_only[]  = {0x78, 0x9c};
    static const unsigned char bad_cm[]    = {0x00, 0x00, 0x03, 0x00};
    static const unsigned char bad_fchk[]  = {0x78, 0x00, 0x03, 0x00};
    static const unsigned char fdict[]     = {0x78, 0xbb, 0x00, 0x00, 0x00, 0x00, 0x03, 0x00};
    if (infl_buf(bad_btype, sizeof bad_btype, out, sizeof out, 0) == UNZ_OK) return fail("error vector", "BTYPE 3", 0);
    if (infl_buf(stored10, sizeof stored10, out, 5, 0) == UNZ_OK) return fail("error vector", "output too small", 0);
    if (infl_buf(hdr_only, sizeof hdr_only, out, sizeof out, INFL_ZLIB) == UNZ_OK) return fail("error vector", "header only", 0);
    if (infl_buf(bad_cm, sizeof bad_cm, out, sizeof out, INFL_ZLIB) == UNZ_OK) return fail("error vector", "CM", 0);
    if (infl_buf(bad_fchk, sizeof bad_fchk, out, sizeof out, INFL_ZLIB) == UNZ_OK) return fail("error vector", "FCHECK", 0);
    if (infl_buf(fdict, sizeof fdict, out, sizeof out, INFL_ZLIB) == UNZ_OK) return fail("error vector", "FDICT", 0);
    /* 5. regression vectors */
    static const unsigned char stored_a[] = {0x01, 0x01, 0x00, 0xfe, 0xff, 0x41};
    static const unsigned char abcdef[]   = {0x00, 0x03, 0x00, 0xfc, 0xff, 0x61, 0x62, 0x63, 0x4b, 0x49, 0x4d, 0x03, 0x00};
    memset(out, 0, sizeof out);
    if (infl_buf(stored_a, sizeof stored_a, out, sizeof out, 0) != UNZ_OK || out[0] != 'A') return fail("regression", "stored A", 0);
    memset(out, 0, sizeof out);
    if (infl_buf(abcdef, sizeof abcdef, out, sizeof out, 0) != UNZ_OK || memcmp(out, "abcdef", 6) != 0)
      return fail("regression", "stored prefix then fixed", 0);
    printf("6 error vectors rejected, 2 regression vectors decoded\n");
  }
  printf("dropin ok\n");
  return 0;
}
