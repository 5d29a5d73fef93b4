/*
 * include_cost.c -- what infl_include() costs a C caller (no device needed: the call only files the
 * piece, reference src/infl/mem.c:114-216).  Streams are fed the way a PNG reader feeds IDAT chunks:
 * half of the pieces 8192 bytes, a quarter 1..64, a quarter 64..4096 (the mix of the benchmark's PNG
 * configuration).  Prints calls, bytes and nanoseconds per call, for fresh streams (their 32 KiB pages are
 * allocated and touched for the first time) and again after infl_reset_pool() (pages kept: the steady state of
 * a caller that reuses its stream objects), and for a second generation of fresh streams made after the first
 * was destroyed (its pages come back from the library's page pool, already touched: the steady state of a
 * caller that makes a stream per image); infl_init / infl_destroy are timed apart.
 *   include_cost [streams] [bytes per stream]
 */
#define _POSIX_C_SOURCE 199309L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include <defl/infl.h>

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static uint32_t rnd(uint32_t *s) {
  *s ^= *s << 13;
  *s ^= *s >> 17;
  *s ^= *s << 5;
  return *s;
}

int main(int argc, char **argv) {
  const size_t   nstreams = argc > 1 ? (size_t)atol(argv[1]) : 4096;
  const uint32_t total    = argc > 2 ? (uint32_t)atol(argv[2]) : 48000;
  uint8_t *src = malloc(total), *dst = malloc(65536);
  infl_stream_t **st = malloc(nstreams * sizeof *st);
  if (!src || !dst || !st) return 2;
  for (uint32_t i = 0; i < total; i++) src[i] = (uint8_t)(i * 131u);

  double t0 = now();
  for (size_t i = 0; i < nstreams; i++)
    if (!(st[i] = infl_init(dst, 65536, INFL_ZLIB))) return 3;
  double t1 = now();
  uint64_t calls = 0, bytes = 0, small = 0;
  uint32_t seed = 12345;
  for (size_t i = 0; i < nstreams; i++) {
    uint32_t pos = 0;
    while (pos < total) {
      const uint32_t k = rnd(&seed) & 3u;
      uint32_t len = k < 2 ? 8192u : k == 2 ? 1u + rnd(&seed) % 64u : 64u + rnd(&seed) % 4033u;
      if (len > total - pos) len = total - pos;
      infl_include(st[i], src + pos, len);
      pos += len;
      calls++;
      bytes += len;
      small += len <= 8192u;
    }
  }
  double t2 = now();
  for (size_t i = 0; i < nstreams; i++) infl_reset_pool(st[i]);
  seed = 12345;
  double w0 = now();
  for (size_t i = 0; i < nstreams; i++) {
    uint32_t pos = 0;
    while (pos < total) {
      const uint32_t k = rnd(&seed) & 3u;
      uint32_t len = k < 2 ? 8192u : k == 2 ? 1u + rnd(&seed) % 64u : 64u + rnd(&seed) % 4033u;
      if (len > total - pos) len = total - pos;
      infl_include(st[i], src + pos, len);
      pos += len;
    }
  }
  double w1 = now();
  for (size_t i = 0; i < nstreams; i++) infl_destroy(st[i]);
  double t3 = now();
  for (size_t i = 0; i < nstreams; i++)
    if (!(st[i] = infl_init(dst, 65536, INFL_ZLIB))) return 3;
  seed = 12345;
  double g0 = now();
  for (size_t i = 0; i < nstreams; i++) {
    uint32_t pos = 0;
    while (pos < total) {
      const uint32_t k = rnd(&seed) & 3u;
      uint32_t len = k < 2 ? 8192u : k == 2 ? 1u + rnd(&seed) % 64u : 64u + rnd(&seed) % 4033u;
      if (len > total - pos) len = total - pos;
      infl_include(st[i], src + pos, len);
      pos += len;
    }
  }
  double g1 = now();
  for (size_t i = 0; i < nstreams; i++) infl_destroy(st[i]);
  printf("{\"streams\": %zu, \"include_calls\": %llu, \"bytes\": %llu, \"ns_per_include\": %.1f, "
         "\"include_GBps\": %.2f, \"ns_per_include_reused\": %.1f, \"include_GBps_reused\": %.2f, "
         "\"ns_per_include_second_generation\": %.1f, \"ns_per_init\": %.1f, \"ns_per_destroy\": %.1f}\n",
         nstreams, (unsigned long long)calls, (unsigned long long)bytes, 1e9 * (t2 - t1) / (double)calls,
         (double)bytes / (t2 - t1) / 1e9, 1e9 * (w1 - w0) / (double)calls, (double)bytes / (w1 - w0) / 1e9,
         1e9 * (g1 - g0) / (double)calls,
         1e9 * (t1 - t0) / (double)nstreams, 1e9 * (t3 - w1) / (double)nstreams);
  (void)small;
  free(src);
  free(dst);
  free(st);
  return 0;
}
