// Micro-benchmarks of the gfx950 properties the parallel inflate kernel is designed around:
// integer VALU issue rate per SIMD at 1..8 waves, LDS accesses at unaligned byte addresses,
// byte-granular LDS stores, unaligned 16-byte global gathers.  Diagnostic tool, not part of the library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k_valu(uint32_t *out, int iters, uint64_t *cyc) {
  uint32_t a[16];
  for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 17 + i + out[0];
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = (a[i] << 1) ^ (a[i] + 0x9e3779b9u);   // v_lshl_add / v_xor style: 2 ops
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint32_t s = 0;
  for (int i = 0; i < 16; i++) s += a[i];
  if (s == 0x12345678u) out[1] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// One dependent chain per wave, 64 links per iteration: cycles per link for one resident wave per SIMD.
// KIND 0 v_add_u32; 1 v_sad_u8; 2 v_cmp + v_cndmask through vcc; 3 v_mov_dpp wave_shr:1 + v_add; 4 v_mov_dpp row_shr:1 + v_add;
// 5 v_and_or_b32; 6 ds_bpermute + v_add; 7 v_add_u32 with SDWA byte select
template <int KIND>
__global__ __launch_bounds__(64) void k_chain(uint32_t *out, int iters, uint64_t *cyc) {
  uint32_t a = threadIdx.x * 17 + out[0], b = threadIdx.x + 3, c = 0;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 64; i++) {
      if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      if (KIND == 1) asm volatile("v_sad_u8 %0, %0, %1, 0" : "+v"(a) : "v"(b));
      if (KIND == 2) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_add_u32 %0, %0, %1" : "+v"(a) : "v"(b) : "vcc");
      if (KIND == 3) asm volatile("s_nop 1\n\tv_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32 %0, %1, %2" : "+v"(a), "+v"(c) : "v"(b));
      if (KIND == 4) asm volatile("s_nop 1\n\tv_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32 %0, %1, %2" : "+v"(a), "+v"(c) : "v"(b));
      if (KIND == 5) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
      if (KIND == 6) { a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x + 63u) & 63u) << 2), (int)a) + b; }
      if (KIND == 7) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(a) : "v"(b));
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (a + c == 0x12345678u) out[1] = a;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// Issue cost: 16 independent chains per wave, 4 waves per SIMD; cycles of SIMD time per instruction.  KIND as in k_chain.
template <int KIND>
__global__ __launch_bounds__(64) void k_rate(uint32_t *out, int iters, uint64_t *cyc) {
  uint32_t a[16], b = threadIdx.x + 3, c[16];
  for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 17 + i + out[0], c[i] = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (KIND == 1) asm volatile("v_sad_u8 %0, %0, %1, 0" : "+v"(a[i]) : "v"(b));
        if (KIND == 2) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
        if (KIND == 3) asm volatile("v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]), "+v"(c[i]));
        if (KIND == 4) asm volatile("v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]), "+v"(c[i]));
        if (KIND == 5) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
        if (KIND == 6) asm volatile("v_sad_u16 %0, %0, %1, 0" : "+v"(a[i]) : "v"(b));
        if (KIND == 7) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(a[i]) : "v"(b));
        if (KIND == 8) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));
        if (KIND == 9) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (KIND == 10) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
        if (KIND == 11) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a[i]) : "v"(b));
      }
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < 16; i++) s += a[i] + c[i];
  if (s == 0x12345678u) out[1] = s;
}

// The PNG un-filter step sequence (bpp 4) on registers only: cycles per step at 1, 2 and 3 waves per SIMD.
__global__ __launch_bounds__(64) void k_png_steps(uint32_t *out, int iters) {
  const uint32_t lane = threadIdx.x, ft = (lane + blockIdx.x) % 5u;
  const uint32_t m_sub = ft == 1 ? 0xffu : 0u, m_up = ft == 2 ? 0xffu : 0u, m_avg = ft == 3 ? 0xffu : 0u, m_paeth = ft == 4 ? 0xffu : 0u;
  uint32_t own[4] = {0, 0, 0, 0}, above[4] = {0, 0, 0, 0}, prev = 0, acc = 0;
  uint32_t cur[4] = {lane * 0x01010101u + out[0], lane * 77u, lane * 0x9e3779b9u, lane + 5u}, upc[4] = {1, 2, 3, 4};
  for (int it = 0; it < iters; it++) {
    uint32_t byte[16], bu[16], o[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t v = (cur[j >> 2] >> (8 * (j & 3))) & 0xffu;
      bu[j] = (uint32_t)__builtin_amdgcn_update_dpp((int)((upc[j >> 2] >> (8 * (j & 3))) & 0xffu), (int)prev, 0x138, 0xf, 0xf, false);
      const uint32_t a = j >= 4 ? byte[j >= 4 ? j - 4 : 0] : own[j < 4 ? j : 0];
      const uint32_t c = j >= 4 ? bu[j >= 4 ? j - 4 : 0] : above[j < 4 ? j : 0];
      const uint32_t b = bu[j], ab = a + b;
      const uint32_t pa = __builtin_amdgcn_sad_u8(b, c, 0), pb = __builtin_amdgcn_sad_u8(a, c, 0), pc = __builtin_amdgcn_sad_u16(ab, c << 1, 0);
      const uint32_t paeth = pa <= (pb < pc ? pb : pc) ? a : pb <= pc ? b : c;
      const uint32_t pred = (a & m_sub) | (b & m_up) | ((ab >> 1) & m_avg) | (paeth & m_paeth);
      byte[j] = (v + pred) & 0xffu;
      prev = byte[j];
      o[j >> 2] |= byte[j] << (8 * (j & 3));
    }
#pragma unroll
    for (int j = 0; j < 4; j++) own[j] = byte[12 + j], above[j] = bu[12 + j];
#pragma unroll
    for (int w = 0; w < 4; w++) { acc ^= o[w]; cur[w] = cur[w] * 1664525u + 1013904223u; }
  }
  if (acc == 0x12345678u) out[1] = acc;
}

// MODE 0 aligned b32, 1 unaligned b32, 2 aligned b64, 3 unaligned b64, 4 aligned b128, 5 unaligned b128 (random addresses)
template <int MODE>
__global__ __launch_bounds__(64) void k_lds_read(uint32_t *out, int iters, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) uint8_t buf[8192 + 64];
  for (int i = threadIdx.x; i < 8192 + 64; i += 64) buf[i] = (uint8_t)(i * 13 + seed);
  __syncthreads();
  uint32_t x = threadIdx.x * 2654435761u + seed, acc = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      x = x * 1664525u + 1013904223u;
      uint32_t off = (x >> 8) & 8191u;
      if (MODE == 0) { off &= ~3u; uint32_t v; __builtin_memcpy(&v, buf + off, 4); acc += v; }
      if (MODE == 1) { off |= 1u; uint32_t v; __builtin_memcpy(&v, buf + off, 4); acc += v; }
      if (MODE == 2) { off &= ~7u; uint64_t v; __builtin_memcpy(&v, (uint64_t *)__builtin_assume_aligned(buf + off, 8), 8); acc += (uint32_t)v + (uint32_t)(v >> 32); }
      if (MODE == 3) { off |= 1u; uint64_t v; __builtin_memcpy(&v, buf + off, 8); acc += (uint32_t)v + (uint32_t)(v >> 32); }
      if (MODE == 4) { off &= ~15u; uint4 v; __builtin_memcpy(&v, (uint4 *)__builtin_assume_aligned(buf + off, 16), 16); acc += v.x + v.w; }
      if (MODE == 5) { off |= 1u; uint4 v; __builtin_memcpy(&v, buf + off, 16); acc += v.x + v.w; }
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

// MODE 0 b8 scatter, 1 unaligned b32 scatter, 2 unaligned b64 scatter, 3 aligned b32 scatter
template <int MODE>
__global__ __launch_bounds__(64) void k_lds_write(uint32_t *out, int iters, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) uint8_t buf[8192 + 64];
  uint32_t x = threadIdx.x * 2654435761u + seed;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      x = x * 1664525u + 1013904223u;
      uint32_t off = (x >> 8) & 8191u;
      if (MODE == 0) buf[off] = (uint8_t)x;
      if (MODE == 1) { off |= 1u; __builtin_memcpy(buf + off, &x, 4); }
      if (MODE == 2) { off |= 1u; uint64_t v = x * 0x100000001ull; __builtin_memcpy(buf + off, &v, 8); }
      if (MODE == 3) { off &= ~3u; __builtin_memcpy(buf + off, &x, 4); }
    }
  }
  __syncthreads();
  out[blockIdx.x * 64 + threadIdx.x] = buf[threadIdx.x];
}

// every lane loads 16 bytes at a random unaligned offset inside its wave's 64 KiB window (a stream's recent output)
template <int BYTES>
__global__ __launch_bounds__(64) void k_gather(const uint8_t *src, uint32_t *out, int iters, uint32_t seed) {
  const uint8_t *win = src + (size_t)blockIdx.x * 65536;
  uint32_t x = (threadIdx.x + blockIdx.x * 64) * 2654435761u + seed, acc = 0;
  for (int it = 0; it < iters; it++) {
    x = x * 1664525u + 1013904223u;
    uint32_t off = (x >> 8) % 65000u;
    if (BYTES == 16) { uint4 v; __builtin_memcpy(&v, win + off, 16); acc += v.x ^ v.w; }
    if (BYTES == 1) { acc += win[off]; }
    if (BYTES == 4) { uint32_t v; __builtin_memcpy(&v, win + off, 4); acc += v; }
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <class F>
static float timeit(F f, int reps = 3) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    CK(hipEventRecord(a));
    f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  uint32_t *out; uint64_t *cyc;
  CK(hipMalloc(&out, 64 << 20)); CK(hipMemset(out, 0, 64 << 20));
  CK(hipMalloc(&cyc, 64));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  const int CUS = p.multiProcessorCount;
  for (int w : {1, 2, 3, 4, 6, 8}) {
    const int iters = 20000, grid = CUS * 4 * w;
    float ms = timeit([&] { hipLaunchKernelGGL(k_valu, dim3(grid), dim3(64), 0, 0, out, iters, cyc); });
    uint64_t c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    double instr = (double)iters * 48.0 * grid;           // 3 VALU per element and iteration (shift, add, xor)
    printf("valu waves/SIMD %d: %.3f ms, wave cycles %llu (%.2f GHz), %.3f VALU instr per ns per SIMD = %.3f per cycle at 2.4 GHz\n",
           w, ms, (unsigned long long)c, c / (ms * 1e6), instr / (ms * 1e6) / (CUS * 4), instr / (ms * 1e6) / (CUS * 4) / 2.4);
  }
  {
    const char *cn[] = {"v_add_u32", "v_sad_u8", "v_cmp+v_cndmask+v_add", "s_nop 1+dpp wave_shr:1+v_add", "s_nop 1+dpp row_shr:1+v_add", "v_and_or_b32",
                        "ds_bpermute+v_add", "v_add_u32 sdwa"};
    for (int w : {1, 2}) {
      const int iters = 2000, grid = CUS * 4 * w;
      float ms[8];
      uint64_t cy[8];
#define CHAIN(K) ms[K] = timeit([&] { hipLaunchKernelGGL(k_chain<K>, dim3(grid), dim3(64), 0, 0, out, iters, cyc); }); CK(hipMemcpy(&cy[K], cyc, 8, hipMemcpyDeviceToHost));
      CHAIN(0) CHAIN(1) CHAIN(2) CHAIN(3) CHAIN(4) CHAIN(5) CHAIN(6) CHAIN(7)
      for (int k = 0; k < 8; k++)
        printf("dependent chain, %d wave(s)/SIMD, %-30s: %.3f ms = %.1f cycles per link at 2.4 GHz (s_memtime: %.1f ticks per link)\n", w, cn[k], ms[k],
               ms[k] * 2.4e6 / (iters * 64.0), (double)cy[k] / (iters * 64.0));
    }
  }
  {
    const char *cn[] = {"v_add_u32", "v_sad_u8", "v_cmp+v_cndmask (2 instr)", "v_mov_dpp wave_shr:1", "v_mov_dpp row_shr:1", "v_and_or_b32", "v_sad_u16",
                        "v_add_u32 sdwa", "v_bfe_u32", "v_min_u32", "v_or3_b32", "v_lshl_or_b32"};
    const int iters = 2000, grid = CUS * 16;
    float ms[12];
#define RATE(K) ms[K] = timeit([&] { hipLaunchKernelGGL(k_rate<K>, dim3(grid), dim3(64), 0, 0, out, iters, cyc); });
    RATE(0) RATE(1) RATE(2) RATE(3) RATE(4) RATE(5) RATE(6) RATE(7) RATE(8) RATE(9) RATE(10) RATE(11)
    for (int k = 0; k < 12; k++)
      printf("issue cost, 4 waves/SIMD, 16 chains, %-26s: %.3f ms = %.2f SIMD cycles per asm statement at 2.4 GHz\n", cn[k], ms[k],
             ms[k] * 2.4e6 / (iters * 64.0 * 4));
  }
  for (int w : {1, 2, 3, 4}) {
    const int iters = 4000, g2 = CUS * 4 * w;
    float ms = timeit([&] { hipLaunchKernelGGL(k_png_steps, dim3(g2), dim3(64), 0, 0, out, iters); });
    printf("png steps on registers, %d wave(s)/SIMD: %.3f ms = %.1f cycles per step and wave, %.1f SIMD cycles per step at 2.4 GHz\n", w, ms,
           ms * 2.4e6 / (iters * 16.0), ms * 2.4e6 / (iters * 16.0) / w);
  }
  const int grid = CUS * 16, iters = 4000;
  const char *rn[] = {"read b32 aligned", "read b32 unaligned", "read b64 aligned", "read b64 unaligned", "read b128 aligned", "read b128 unaligned"};
  float rm[6];
  rm[0] = timeit([&] { hipLaunchKernelGGL(k_lds_read<0>, dim3(grid), dim3(64), 0, 0, out, iters, 1u); });
  rm[1] = timeit([&] { hipLaunchKernelGGL(k_lds_read<1>, dim3(grid), dim3(64), 0, 0, out, iters, 1u); });
  rm[2] = timeit([&] { hipLaunchKernelGGL(k_lds_read<2>, dim3(grid), dim3(64), 0, 0, out, iters, 1u); });
  rm[3] = timeit([&] { hipLaunchKernelGGL(k_lds_read<3>, dim3(grid), dim3(64), 0, 0, out, iters, 1u); });
  rm[4] = timeit([&] { hipLaunchKernelGGL(k_lds_read<4>, dim3(grid), dim3(64), 0, 0, out, iters, 1u); });
  rm[5] = timeit([&] { hipLaunchKernelGGL(k_lds_read<5>, dim3(grid), dim3(64), 0, 0, out, iters, 1u); });
  for (int m = 0; m < 6; m++)
    printf("lds %s (random addresses, 16 waves/CU): %.3f ms = %.1f cycles per wave-instruction per CU at 2.4 GHz\n", rn[m], rm[m],
           rm[m] * 1e-3 * 2.4e9 / ((double)iters * 8 * 16));
  const char *wn[] = {"write b8", "write b32 unaligned", "write b64 unaligned", "write b32 aligned"};
  float wm[4];
  wm[0] = timeit([&] { hipLaunchKernelGGL(k_lds_write<0>, dim3(grid), dim3(64), 0, 0, out, iters, 1u); });
  wm[1] = timeit([&] { hipLaunchKernelGGL(k_lds_write<1>, dim3(grid), dim3(64), 0, 0, out, iters, 1u); });
  wm[2] = timeit([&] { hipLaunchKernelGGL(k_lds_write<2>, dim3(grid), dim3(64), 0, 0, out, iters, 1u); });
  wm[3] = timeit([&] { hipLaunchKernelGGL(k_lds_write<3>, dim3(grid), dim3(64), 0, 0, out, iters, 1u); });
  for (int m = 0; m < 4; m++)
    printf("lds %s (random addresses, 16 waves/CU): %.3f ms = %.1f cycles per wave-instruction per CU at 2.4 GHz\n", wn[m], wm[m],
           wm[m] * 1e-3 * 2.4e9 / ((double)iters * 8 * 16));
  // correctness of unaligned LDS reads: compare one result with the host
  {
    hipLaunchKernelGGL(k_lds_read<3>, dim3(1), dim3(64), 0, 0, out, 1, 7u);
    std::vector<uint32_t> h(64); CK(hipMemcpy(h.data(), out, 256, hipMemcpyDeviceToHost));
    std::vector<uint8_t> buf(8192 + 64);
    for (int i = 0; i < 8192 + 64; i++) buf[i] = (uint8_t)(i * 13 + 7);
    int bad = 0;
    for (int t = 0; t < 64; t++) {
      uint32_t x = t * 2654435761u + 7u, acc = 0;
      for (int u = 0; u < 8; u++) {
        x = x * 1664525u + 1013904223u;
        uint32_t off = ((x >> 8) & 8191u) | 1u;
        uint64_t v; __builtin_memcpy(&v, &buf[off], 8); acc += (uint32_t)v + (uint32_t)(v >> 32);
      }
      bad += acc != h[t];
    }
    printf("unaligned ds_read_b64 check: %d lanes differ from the host\n", bad);
  }
  uint8_t *src; const int gw = CUS * 16;
  CK(hipMalloc(&src, (size_t)gw * 65536)); CK(hipMemset(src, 3, (size_t)gw * 65536));
  for (int bytes : {1, 4, 16}) {
    const int git = 2000;
    float ms = bytes == 1 ? timeit([&] { hipLaunchKernelGGL(k_gather<1>, dim3(gw), dim3(64), 0, 0, src, out, git, 5u); })
             : bytes == 4 ? timeit([&] { hipLaunchKernelGGL(k_gather<4>, dim3(gw), dim3(64), 0, 0, src, out, git, 5u); })
                          : timeit([&] { hipLaunchKernelGGL(k_gather<16>, dim3(gw), dim3(64), 0, 0, src, out, git, 5u); });
    printf("global gather %2d B per lane, random unaligned inside a 64 KiB window per wave, 16 waves/CU (256 MiB in all): %.3f ms = %.0f cycles per wave-instruction per CU, %.1f G lane-loads/s\n",
           bytes, ms, ms * 1e-3 * 2.4e9 / ((double)git * 16), (double)git * gw * 64 / (ms * 1e6));
  }
  return 0;
}
