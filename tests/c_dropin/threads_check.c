/*
 * threads_check.c -- the library called from several host threads at once, each on streams of its
 * own (the reference allows exactly that: a stream is single-threaded state, distinct streams on
 * distinct threads are fine once its lazily built tables exist, src/infl/infl.c:1004-1006,
 * 1510-1511).  Test infrastructure.  Every thread loops over the golden file pairs and decodes
 * them through infl_buf(), infl_include() pieces + infl_many(), infl_stream() feeds, infl_batch()
 * and infl_batch_noalloc(), comparing every result with the raw twin; the first mismatch ends the
 * program with exit code 1.
 *
 *   threads_check <dir with raw/ and compressed/> [threads] [rounds]
 */
#include <dirent.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <defl/infl.h>
#include <defl/infl_batch.h>

typedef struct pair_t {
  char           name[128];
  unsigned char *raw, *comp;
  size_t         rawlen, complen;
} pair_t;

static pair_t pairs[256];
static int    npairs;
static int    rounds = 3;
static volatile int failed;

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

static int bad(int tid, const char *what, const char *name, int rc) {
  fprintf(stderr, "FAILED thread %d %s: %s (rc %d)\n", tid, what, name, rc);
  failed = 1;
  return 1;
}

static int one_shot(int tid, const pair_t *p) {
  unsigned char *out = calloc(1, p->rawlen + 1000);
  int rc = infl_buf(p->comp, (uint32_t)p->complen, out, (uint32_t)p->rawlen + 1000, 0);
  int r  = (rc != UNZ_OK || memcmp(out, p->raw, p->rawlen) != 0) ? bad(tid, "infl_buf", p->name, rc) : 0;
  free(out);
  return r;
}

static int streamed(int tid, const pair_t *p, size_t feed) {
  unsigned char *out = calloc(1, p->rawlen + 1000);
  infl_stream_t *st  = infl_init(out, (uint32_t)p->rawlen + 1000, 0);
  int    rc = UNZ_UNFINISHED;
  size_t pos = 0;
  if (!p->complen) rc = infl_stream(st, NULL, 0);
  while (pos < p->complen && rc == UNZ_UNFINISHED) {
    size_t n = feed < p->complen - pos ? feed : p->complen - pos;
    rc = infl_stream(st, p->comp + pos, (uint32_t)n);
    pos += n;
  }
  int r = (rc != UNZ_OK || infl_output_pos(st) != p->rawlen || memcmp(out, p->raw, p->rawlen) != 0)
              ? bad(tid, "infl_stream", p->name, rc) : 0;
  infl_destroy(st);
  free(out);
  return r;
}

enum { GROUP = 12 };

static int many(int tid, int first) {
  infl_stream_t *st[GROUP];
  unsigned char *out[GROUP];
  int32_t        status[GROUP];
  int            r = 0;
  for (int k = 0; k < GROUP; k++) {
    const pair_t *p = &pairs[(first + k) % npairs];
    out[k] = calloc(1, p->rawlen + 1000);
    st[k]  = infl_init(out[k], (uint32_t)p->rawlen + 1000, 0);
    for (size_t pos = 0; pos < p->complen;) {               /* pieces of 1 .. 3000 bytes */
      size_t n = 1 + (pos * 7 + (size_t)k * 131) % 3000;
      if (n > p->complen - pos) n = p->complen - pos;
      infl_include(st[k], p->comp + pos, (uint32_t)n);
      pos += n;
    }
  }
  int rc = infl_many(st, GROUP, status);
  for (int k = 0; k < GROUP; k++) {
    const pair_t *p = &pairs[(first + k) % npairs];
    const int want = p->complen ? UNZ_OK : UNZ_NOOP;
    if (!r && (rc != 0 || status[k] != want || infl_output_pos(st[k]) != p->rawlen || memcmp(out[k], p->raw, p->rawlen) != 0))
      r = bad(tid, "infl_many", p->name, rc ? rc : status[k]);
    infl_destroy(st[k]);
    free(out[k]);
  }
  return r;
}

static int batch(int tid, int first, int noalloc) {
  uint64_t so[GROUP + 1] = {0}, dof[GROUP + 1] = {0};
  int32_t  status[GROUP];
  uint32_t outlen[GROUP];
  int      r = 0;
  for (int k = 0; k < GROUP; k++) {
    const pair_t *p = &pairs[(first + k) % npairs];
    so[k + 1]  = so[k] + p->complen;
    dof[k + 1] = dof[k] + p->rawlen + 8;
  }
  unsigned char *src = malloc(so[GROUP] + 1), *dst = malloc(dof[GROUP] + 1);
  memset(dst, 0xA5, dof[GROUP] + 1);
  for (int k = 0; k < GROUP; k++) {
    const pair_t *p = &pairs[(first + k) % npairs];
    memcpy(src + so[k], p->comp, p->complen);
  }
  int rc = noalloc ? infl_batch_noalloc(src, so, dst, dof, 0, GROUP, status, outlen)
                   : infl_batch(src, so, dst, dof, 0, GROUP, status, outlen);
  for (int k = 0; k < GROUP && !r; k++) {
    const pair_t *p = &pairs[(first + k) % npairs];
    if (rc != 0 || status[k] != UNZ_OK || outlen[k] != p->rawlen || memcmp(dst + dof[k], p->raw, p->rawlen) != 0)
      r = bad(tid, noalloc ? "infl_batch_noalloc" : "infl_batch", p->name, rc ? rc : status[k]);
  }
  free(src);
  free(dst);
  return r;
}

static void *worker(void *arg) {
  const int tid = (int)(intptr_t)arg;
  for (int round = 0; round < rounds && !failed; round++) {
    for (int i = 0; i < npairs && !failed; i++) {
      const int     k = (i * 5 + tid * 37 + round * 11) % npairs;    /* every thread walks the pairs in its own order */
      const pair_t *p = &pairs[k];
      if (one_shot(tid, p)) return NULL;
      if (p->complen <= 20000 && streamed(tid, p, 64u << (i % 5))) return NULL;
      if (i % GROUP == 0) {
        if (many(tid, k)) return NULL;
        if (batch(tid, k, 0)) return NULL;
        if (batch(tid, k, 1)) return NULL;
      }
    }
  }
  return NULL;
}

static int by_name(const void *a, const void *b) { return strcmp(((const pair_t *)a)->name, ((const pair_t *)b)->name); }

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const char *dir = argv[1];
  int nthreads = argc > 2 ? atoi(argv[2]) : 4;
  if (argc > 3) rounds = atoi(argv[3]);
  if (nthreads < 1 || nthreads > 16) return 2;
  if (infl_gpu_device_count() < 1 || infl_gpu_init(0) != 0) {
    fprintf(stderr, "no usable HIP device\n");
    return 3;
  }
  char cdir[1024];
  snprintf(cdir, sizeof cdir, "%s/compressed", dir);
  DIR *d = opendir(cdir);
  if (!d) return 2;
  for (struct dirent *e; (e = readdir(d)) && npairs < 256;) {
    if (e->d_name[0] == '.' || strlen(e->d_name) >= sizeof pairs[0].name) continue;
    pair_t *p = &pairs[npairs];
    strcpy(p->name, e->d_name);
    if (!(p->raw = slurp(dir, "raw", e->d_name, &p->rawlen))) continue;      /* (streams without a raw twin) */
    if (!(p->comp = slurp(dir, "compressed", e->d_name, &p->complen))) return 2;
    if (p->rawlen > 120000) {                    /* keep a thread's batches inside the fixed workspace of infl_batch_noalloc */
      free(p->raw);
      free(p->comp);
      continue;
    }
    npairs++;
  }
  closedir(d);
  qsort(pairs, (size_t)npairs, sizeof pairs[0], by_name);
  if (npairs < 100) return 2;

  pthread_t th[16];
  for (int t = 0; t < nthreads; t++)
    if (pthread_create(&th[t], NULL, worker, (void *)(intptr_t)t) != 0) return 2;
  for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  if (failed) return 1;
  printf("threads ok: %d threads x %d rounds over %d file pairs\n", nthreads, rounds, npairs);
  return 0;
}
