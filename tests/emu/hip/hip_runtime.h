// Stand-in for <hip/hip_runtime.h> used ONLY by the host-side kernel emulation in tests/emu:
// it lets g++ compile the kernel source unchanged and run every "lane" as a host thread, so
// that the kernel's own logic (indexing, bounds, lockstep hand-offs) can be checked on the CPU
// under AddressSanitizer before a GPU ever runs it.  Test infrastructure, never shipped.
#pragma once
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <functional>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
typedef void *hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipDeviceAttributeMultiprocessorCount = 1 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int *v, int, int) { *v = 1; return hipSuccess; }   // a one-CU device: small grids

#define __device__
#define __constant__ static
#define __global__
#define __shared__ static
#define __forceinline__ inline
#define __noinline__
#define __launch_bounds__(...)

namespace emu {
struct Group;
extern dim3   t_threadIdx, t_blockIdx, t_gridDim;   // the running lane's (lanes are fibers of one host thread)
extern Group *t_group;
extern int    t_lane;
extern int    g_lanes_per_group;                     // lanes that share barriers, ballots and permutes (a stream's lanes)
void     group_barrier();
void     block_barrier();
uint64_t ballot(bool pred);
int      bpermute(int byte_addr, int value);
void     launch(unsigned grid, unsigned block, const std::function<void()> &fn);
}  // namespace emu

#define threadIdx emu::t_threadIdx
#define blockIdx emu::t_blockIdx
#define gridDim emu::t_gridDim

#define DEFL_LAUNCH_GROUP(T) emu::g_lanes_per_group = (T)
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
  emu::launch((grid).x, (block).x, [=] { kern(__VA_ARGS__); })

#define __builtin_amdgcn_fence(order, scope) emu::group_barrier()
#define __builtin_amdgcn_wave_barrier() emu::group_barrier()
#define __syncthreads() emu::block_barrier()
// v_dot4_u32_u8: c + the four byte products of a and b
static inline uint32_t __builtin_amdgcn_udot4(uint32_t a, uint32_t b, uint32_t c, bool) {
  for (int k = 0; k < 4; k++) c += ((a >> (8 * k)) & 0xffu) * ((b >> (8 * k)) & 0xffu);
  return c;
}
#define __ballot(p) emu::ballot(p)
#define __popcll(x) __builtin_popcountll(x)
#define __builtin_amdgcn_ds_bpermute(a, v) emu::bpermute(a, v)
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t s) {
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31u));
}
// v_sad_u8 / v_sad_u16: c + the sum of absolute differences of the four bytes / two halves of a and b
static inline uint32_t __builtin_amdgcn_sad_u8(uint32_t a, uint32_t b, uint32_t c) {
  for (int k = 0; k < 4; k++) {
    const int x = (a >> (8 * k)) & 0xff, y = (b >> (8 * k)) & 0xff;
    c += (uint32_t)(x > y ? x - y : y - x);
  }
  return c;
}
static inline uint32_t __builtin_amdgcn_sad_u16(uint32_t a, uint32_t b, uint32_t c) {
  for (int k = 0; k < 2; k++) {
    const int x = (a >> (16 * k)) & 0xffff, y = (b >> (16 * k)) & 0xffff;
    c += (uint32_t)(x > y ? x - y : y - x);
  }
  return c;
}
// DPP controls used by the kernels: row_shr:n (0x110 + n), the two row broadcasts and wave_shr:1 (0x138); lanes
// whose source lies outside their row of 16 (outside the wavefront for wave_shr), or whose row is masked off, get `old`.
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int, bool) {
  const int lane = emu::t_lane, row = lane >> 4;
  int from = -1;
  if (ctrl >= 0x111 && ctrl <= 0x11f) {
    const int n = ctrl - 0x110;
    if ((lane & 15) >= n) from = lane - n;
  } else if (ctrl == 0x138) {                      // wave_shr:1
    if (lane & 63) from = lane - 1;
  } else if (ctrl == 0x142) {                      // row_bcast:15
    if (row >= 1) from = row * 16 - 1;
  } else if (ctrl == 0x143) {                      // row_bcast:31
    if (row >= 2) from = 31;
  }
  const int got = emu::bpermute((from < 0 ? lane : from) << 2, src);   // every lane takes part
  return (from >= 0 && ((row_mask >> row) & 1)) ? got : old;
}
// v_mbcnt_lo / v_mbcnt_hi: base + the set bits of the mask half that belong to lanes below this one
static inline uint32_t __builtin_amdgcn_mbcnt_lo(uint32_t m, uint32_t base) {
  const int lane = emu::t_lane & 63;
  return base + (uint32_t)__builtin_popcount(lane >= 32 ? m : m & ((1u << lane) - 1u));
}
static inline uint32_t __builtin_amdgcn_mbcnt_hi(uint32_t m, uint32_t base) {
  const int lane = emu::t_lane & 63;
  return base + (uint32_t)__builtin_popcount(lane <= 32 ? 0u : m & ((1u << (lane - 32)) - 1u));
}
#define __builtin_amdgcn_readfirstlane(v) (v)
#define __builtin_amdgcn_readlane(v, l) emu::bpermute((int)((l) << 2), (int)(v))

static inline uint32_t __brev(uint32_t v) {
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
  return __builtin_bswap32(v);
}
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline uint32_t atomicAnd(uint32_t *p, uint32_t v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
// v_alignbyte_b32: bytes [s, s + 4) of the 8-byte value hi:lo (s taken modulo 4)
// v_bfe_u32: offset and width come from the low five bits of their operands; width 0 gives 0
static inline uint32_t __builtin_amdgcn_ubfe(uint32_t v, uint32_t off, uint32_t width) {
  off &= 31u;
  width &= 31u;
  return width ? (v >> off) & ((1u << width) - 1u) : 0u;
}
static inline uint32_t __builtin_amdgcn_alignbyte(uint32_t hi, uint32_t lo, uint32_t s) {
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * (s & 3u)));
}
// v_mul_u32_u24: the product of the low 24 bits of both operands
static inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }

// v_perm_b32: every selector byte picks one of the 8 bytes of s0:s1 (0-3 from s1, 4-7 from s0); 12 gives 0x00,
// above 12 gives 0xff, 8-11 replicate a sign bit (not used by the kernel)
static inline uint32_t __builtin_amdgcn_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
  const uint64_t src = ((uint64_t)s0 << 32) | s1;
  uint32_t out = 0;
  for (int b = 0; b < 4; b++) {
    const uint32_t k = (sel >> (8 * b)) & 0xffu;
    uint32_t v;
    if (k < 8) v = (uint32_t)(src >> (8 * k)) & 0xffu;
    else if (k < 12) v = ((src >> (16 * (k - 8) + 15)) & 1u) ? 0xffu : 0u;
    else v = k == 12 ? 0u : 0xffu;
    out |= v << (8 * b);
  }
  return out;
}
static inline uint32_t atomicMax(uint32_t *p, uint32_t v) {
  uint32_t old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
  }
  return old;
}
