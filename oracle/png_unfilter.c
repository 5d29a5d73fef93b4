/*
 * png_unfilter.c -- CPU checker for the PNG un-filter kernel (TEST INFRASTRUCTURE ONLY: used by
 * tests/ as the checker, never by the shipped library).
 *
 * The step that follows inflate in a PNG decoder ("undo filters", reference README.md:150-152);
 * the reference itself holds no code and no fixture for it (it lives in its caller, im), so this is
 * a restatement of the PNG specification (ISO/IEC 15948:2003, section 9 "Filtering": filter types
 * 0 None, 1 Sub, 2 Up, 3 Average, 4 Paeth over bytes, bpp = bytes per complete pixel, rounded up to
 * 1) and nothing in the reference pins it: PARITY UNPINNED BY THE REFERENCE.  tests/ pin it against
 * an independent Python restatement and against filter round trips.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

static int
paeth(int a, int b, int c) {
  int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  if (pb <= pc) return b;
  return c;
}

/* in: rows x (1 + rowbytes) filtered bytes; out: rows x rowbytes.  0, or -1 on a filter type > 4. */
int
oracle_png_unfilter(const uint8_t *in, uint8_t *out, uint32_t rows, uint32_t rowbytes, uint32_t bpp) {
  for (uint32_t r = 0; r < rows; r++) {
    const uint8_t *f = in + (size_t)r * (rowbytes + 1u);
    uint8_t       *o = out + (size_t)r * rowbytes;
    const uint8_t *up = r ? o - rowbytes : NULL;
    if (f[0] > 4) return -1;
    for (uint32_t x = 0; x < rowbytes; x++) {
      int a = x >= bpp ? o[x - bpp] : 0;
      int b = up ? up[x] : 0;
      int c = up && x >= bpp ? up[x - bpp] : 0;
      int v = f[1 + x];
      switch (f[0]) {
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: break;
      }
      o[x] = (uint8_t)v;
    }
  }
  return 0;
}
