// infl_gpu.hip -- HIP runtime glue behind the C ABI of include/defl/infl_batch.h:
// device selection, device-memory helpers, the device-resident batch call and
// the host-memory batch call (upload, one launch, download).  No decode logic
// lives here and there is no CPU decoder to fall back to: without a usable HIP
// device every decode entry point fails with UNZ_ERR.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/defl/infl_batch.h"
#include "infl_device.h"

namespace {

std::mutex g_mu;            // guards the cached workspace below
int        g_lanes = 16;    // lanes of a wavefront per stream
int        g_variant = 2;   // 0 reference-geometry tables, 1 compact tables + deferred pass, 2 parallel first

struct Buf {
  void  *p   = nullptr;
  size_t cap = 0;
};

// Workspace of the host-memory path, grown on demand and kept for reuse.
// One set per device the process has used.
struct Workspace {
  Buf   src, dst, soff, doff, status, outlen, detail, inpos;   // device
  Buf   hdst;                                            // pinned host staging
};
constexpr int MAX_DEV = 16;
Workspace     g_ws[MAX_DEV];

bool dev_reserve(Buf &b, size_t bytes) {
  if (bytes <= b.cap) return true;
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  size_t want = bytes + bytes / 4 + 4096;
  if (hipMalloc(&b.p, want) != hipSuccess) {
    (void)hipGetLastError();
    if (hipMalloc(&b.p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      b.p = nullptr;
      return false;
    }
    want = bytes;
  }
  b.cap = want;
  return true;
}

bool host_reserve(Buf &b, size_t bytes) {
  if (bytes <= b.cap) return true;
  if (b.p) (void)hipHostFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  size_t want = bytes + bytes / 4 + 4096;
  if (hipHostMalloc(&b.p, want, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    b.p = nullptr;
    return false;
  }
  b.cap = want;
  return true;
}

bool device_ready() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return false;
  }
  return true;
}

}  // namespace

extern "C" {

int infl_gpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int infl_gpu_init(int device) {
  if (device < 0 || device >= infl_gpu_device_count()) return -1;
  if (hipSetDevice(device) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return hipFree(nullptr) == hipSuccess ? 0 : -1;    // forces context creation
}

void *infl_gpu_malloc(size_t bytes) {
  void *p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void infl_gpu_free(void *dev_ptr) {
  if (dev_ptr) (void)hipFree(dev_ptr);
}

int infl_gpu_upload(void *dev_dst, const void *host_src, size_t bytes) {
  if (!bytes) return 0;
  return hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}

int infl_gpu_download(void *host_dst, const void *dev_src, size_t bytes) {
  if (!bytes) return 0;
  return hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

int infl_gpu_sync(void) { return hipDeviceSynchronize() == hipSuccess ? 0 : -1; }

int infl_gpu_set_lanes_per_stream(int lanes) {
  if (lanes == 64 || lanes == 32 || lanes == 16 || lanes == 8) g_lanes = lanes;
  return g_lanes;
}

int infl_gpu_set_table_variant(int variant) {
  if (variant >= 0 && variant <= 2) g_variant = variant;
  return g_variant;
}

int infl_batch_dev(const void *dev_src, const uint64_t *dev_src_off, void *dev_dst,
                   const uint64_t *dev_dst_off, int flags, size_t n, int32_t *dev_status,
                   uint32_t *dev_outlen, void *hip_stream) {
  if (n == 0) return 0;
  if (!dev_src_off || !dev_dst_off || !dev_status || !dev_outlen) return -1;
  return infl_launch_batch(dev_src, dev_src_off, dev_dst, dev_dst_off, flags, n, dev_status,
                           dev_outlen, nullptr, nullptr, g_lanes, g_variant, hip_stream);
}

// Same with the per-stream detail codes (INFL_D_*); for tests and diagnosis.
int infl_batch_dev_detail(const void *dev_src, const uint64_t *dev_src_off, void *dev_dst,
                          const uint64_t *dev_dst_off, int flags, size_t n, int32_t *dev_status,
                          uint32_t *dev_outlen, int32_t *dev_detail, void *hip_stream) {
  if (n == 0) return 0;
  if (!dev_src_off || !dev_dst_off || !dev_status || !dev_outlen) return -1;
  return infl_launch_batch(dev_src, dev_src_off, dev_dst, dev_dst_off, flags, n, dev_status,
                           dev_outlen, dev_detail, nullptr, g_lanes, g_variant, hip_stream);
}

static Buf  g_pin;                 // see infl_rt_pinned_acquire
static bool g_pin_busy = false;
static std::mutex g_pin_mu;

void *infl_rt_pinned_acquire(size_t bytes) {
  if (!device_ready()) return nullptr;
  std::lock_guard<std::mutex> lock(g_pin_mu);
  if (g_pin_busy || !host_reserve(g_pin, bytes + 16)) return nullptr;
  g_pin_busy = true;
  return g_pin.p;
}

void infl_rt_pinned_release(void *p) {
  std::lock_guard<std::mutex> lock(g_pin_mu);
  if (p && p == g_pin.p) g_pin_busy = false;
}

int infl_rt_batch_host(const void *src, const uint64_t *src_off, void *dst,
                       const uint64_t *dst_off, void *const *dst_ptrs, int flags, size_t n,
                       int32_t *status, uint32_t *outlen, int32_t *detail, uint32_t *inpos) {
  if (n == 0) return 0;
  if (!device_ready()) return -1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return -1;

  std::lock_guard<std::mutex> lock(g_mu);
  Workspace &w = g_ws[dev];
  const size_t src_bytes = (size_t)src_off[n], dst_bytes = (size_t)dst_off[n];
  const size_t off_bytes = (n + 1) * sizeof(uint64_t);

  if (!dev_reserve(w.src, src_bytes + 16) || !dev_reserve(w.dst, dst_bytes + 16) ||
      !dev_reserve(w.soff, off_bytes) || !dev_reserve(w.doff, off_bytes) ||
      !dev_reserve(w.status, n * 4) || !dev_reserve(w.outlen, n * 4) ||
      !dev_reserve(w.detail, n * 4) || !dev_reserve(w.inpos, n * 4) ||
      !host_reserve(w.hdst, dst_bytes + 16))
    return -12;

  hipStream_t st = nullptr;
  bool ok = true;
  if (src_bytes) ok &= hipMemcpyAsync(w.src.p, src, src_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
  ok &= hipMemcpyAsync(w.soff.p, src_off, off_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
  ok &= hipMemcpyAsync(w.doff.p, dst_off, off_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
  if (!ok) return -1;
  if (infl_launch_batch(w.src.p, (const uint64_t *)w.soff.p, w.dst.p, (const uint64_t *)w.doff.p,
                        flags, n, (int32_t *)w.status.p, (uint32_t *)w.outlen.p,
                        (int32_t *)w.detail.p, (uint32_t *)w.inpos.p, g_lanes, g_variant, st) != 0)
    return -1;
  ok &= hipMemcpyAsync(status, w.status.p, n * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
  ok &= hipMemcpyAsync(outlen, w.outlen.p, n * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
  if (detail) ok &= hipMemcpyAsync(detail, w.detail.p, n * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
  if (inpos) ok &= hipMemcpyAsync(inpos, w.inpos.p, n * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
  if (dst_bytes)
    ok &= hipMemcpyAsync(w.hdst.p, w.dst.p, dst_bytes, hipMemcpyDeviceToHost, st) == hipSuccess;
  ok &= hipStreamSynchronize(st) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    return -1;
  }
  // Hand back exactly the bytes each stream produced; the rest of the caller's
  // destination is left untouched.
  auto scatter = [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
      size_t len = outlen[i], cap = (size_t)(dst_off[i + 1] - dst_off[i]);
      if (len > cap) len = cap;
      uint8_t *to = dst_ptrs ? (uint8_t *)dst_ptrs[i] : (uint8_t *)dst + dst_off[i];
      if (len) memcpy(to, (const uint8_t *)w.hdst.p + dst_off[i], len);
    }
  };
  // a few host threads for large batches: one core copies about 10 GB/s, the link twice that
  const size_t nthreads = dst_bytes >= (32u << 20) && n >= 64 ? 8 : 1;
  if (nthreads == 1) {
    scatter(0, n);
  } else {
    std::vector<std::thread> pool;
    size_t lo = 0;
    for (size_t t = 1; t <= nthreads; t++) {         // equal byte shares, cut at stream boundaries
      size_t hi = n;
      if (t < nthreads) {
        const uint64_t want = dst_off[n] / nthreads * t;
        hi = (size_t)(std::lower_bound(dst_off, dst_off + n + 1, want) - dst_off);
        hi = std::min(std::max(hi, lo), n);
      }
      if (hi > lo) pool.emplace_back(scatter, lo, hi);
      lo = hi;
    }
    for (auto &th : pool) th.join();
  }
  return 0;
}

int infl_batch(const void *src, const uint64_t *src_off, void *dst, const uint64_t *dst_off,
               int flags, size_t n, int32_t *status, uint32_t *outlen) {
  if (n == 0) return 0;
  if (!src_off || !dst_off || !status || !outlen) return -1;
  return infl_rt_batch_host(src, src_off, dst, dst_off, nullptr, flags == 1 ? 1 : 0, n, status, outlen,
                            nullptr, nullptr);
}

}  // extern "C"
