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
#include <new>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/defl/infl_batch.h"
#include "infl_device.h"

namespace {

std::atomic<int> g_lanes{16};    // lanes of a wavefront per stream
std::atomic<int> g_variant{2};   // 0 reference-geometry tables, 1 compact tables + deferred pass, 2 parallel first

struct Buf {
  void  *p   = nullptr;
  size_t cap = 0;
};

// Workspace of the host-memory path, grown on demand and kept for reuse.
// One set per device the process has used.
struct Workspace {
  Buf   src, dst, soff, doff, status, outlen, detail, inpos, order;   // device
  Buf   hdst, hmeta, hsrc;                               // pinned host staging: output bytes, the four result arrays, input
};
constexpr int MAX_DEV = 16;
Workspace     g_ws[MAX_DEV];
std::mutex    g_mu[MAX_DEV];   // one per device: host threads that drive different GPUs do not wait for each other

// Streams and events of the sliced host-memory path (default priority), made on first use.
constexpr int      kPipeSlices     = 16;               // at most this many slices per call
constexpr int      kPipeStreams    = 4;                // slices in flight
constexpr size_t   kPipeMinBytes   = 64u << 20;        // smaller batches are not sliced
constexpr size_t   kPipeMinStreams = 256;              // a slice holds at least this many streams
constexpr size_t   kOrderMinStreams = 4096;            // batches (or slices) of at least this many streams are put into size classes
struct Pipe {
  hipStream_t st[kPipeStreams] = {}, up = nullptr;        // decode + download, in turn; uploads
  hipEvent_t  ev[kPipeSlices] = {}, evu[kPipeSlices] = {};  // slice downloaded; slice uploaded
  int         state               = 0;                 // 0 not tried, 1 ready, -1 failed
};
Pipe g_pipe[MAX_DEV];

// Staging of infl_batch_noalloc(): device memory behind the inputs and outputs of a host-memory
// batch (source arena, destination arena, the two offset tables, the two result arrays), inside a
// file-scope array of the code object instead of hipMalloc'ed buffers.  Every GPU the library runs
// on holds its own copy of the array, at its own address, until the process ends; in host code the
// array's name is only a handle for hipGetSymbolAddress, never a pointer a kernel could use.
constexpr size_t kNaSrcBytes = INFL_NOALLOC_MAX_SRC + 64;             // (+ a little slack behind each arena)
constexpr size_t kNaDstBytes = INFL_NOALLOC_MAX_DST + 64;
constexpr size_t kNaOffBytes = (INFL_NOALLOC_MAX_STREAMS + 1) * sizeof(uint64_t);
constexpr size_t kNaResBytes = INFL_NOALLOC_MAX_STREAMS * sizeof(uint32_t);
constexpr size_t kNaSrcAt = 0, kNaDstAt = kNaSrcAt + kNaSrcBytes, kNaSoffAt = kNaDstAt + kNaDstBytes,
                 kNaDoffAt = kNaSoffAt + kNaOffBytes, kNaStatusAt = kNaDoffAt + kNaOffBytes,
                 kNaOutlenAt = kNaStatusAt + kNaResBytes, kNaTotal = kNaOutlenAt + kNaResBytes;
static_assert(kNaSrcBytes % 64 == 0 && kNaDstBytes % 64 == 0 && kNaOffBytes % 8 == 0, "regions stay aligned");
static __device__ __align__(256) unsigned char scratch[kNaTotal];

unsigned char *g_scratch_ptr[MAX_DEV] = {};            // device address of `scratch` per device (under g_mu[dev])

// One stream of the library's own per device (default priority, not synchronising with the null
// stream) for the calls that upload, decode, download and wait: infl_stream() steps and
// infl_batch_noalloc().  Made by infl_gpu_init() or on first use; caller holds g_mu[dev].
hipStream_t g_aux_stream[MAX_DEV] = {};
hipStream_t aux_stream_on(int dev) {
  if (!g_aux_stream[dev] && hipStreamCreateWithFlags(&g_aux_stream[dev], hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    g_aux_stream[dev] = nullptr;
  }
  return g_aux_stream[dev];
}

// Looks the array's address up on the current device (once per device); caller holds g_mu[dev].
unsigned char *scratch_on(int dev) {
  if (!g_scratch_ptr[dev]) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) {
      (void)hipGetLastError();
      return nullptr;
    }
    g_scratch_ptr[dev] = (unsigned char *)p;
  }
  return g_scratch_ptr[dev];
}

bool pipe_ready(int dev) {                             // caller holds g_mu[dev]
  Pipe &p = g_pipe[dev];
  if (p.state) return p.state > 0;
  p.state = -1;
  for (auto &s : p.st)
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return false;
  if (hipStreamCreateWithFlags(&p.up, hipStreamNonBlocking) != hipSuccess) return false;
  for (auto &e : p.ev)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
  for (auto &e : p.evu)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
  p.state = 1;
  return true;
}

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
  if (hipFree(nullptr) != hipSuccess) return -1;     // forces context creation
  if (device < MAX_DEV) {
    std::lock_guard<std::mutex> lock(g_mu[device]);
    if (!scratch_on(device) || !aux_stream_on(device)) return -1;
  }
  return 0;
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
  if (lanes == 64 || lanes == 32 || lanes == 16 || lanes == 8) g_lanes.store(lanes);
  return g_lanes.load();
}

int infl_gpu_set_table_variant(int variant) {
#ifdef DEFL_PAR_PROFILE
  constexpr int kMaxVariant = 3;   // diagnostic builds: the parallel pass alone, handed-over streams keep their mark
#else
  constexpr int kMaxVariant = 2;
#endif
  if ((variant >= 0 && variant <= kMaxVariant) || variant == 5) g_variant.store(variant);
  return g_variant.load();
}

int infl_batch_dev(const void *dev_src, const uint64_t *dev_src_off, void *dev_dst,
                   const uint64_t *dev_dst_off, int flags, size_t n, int32_t *dev_status,
                   uint32_t *dev_outlen, void *hip_stream) {
  if (n == 0) return 0;
  if (!dev_src_off || !dev_dst_off || !dev_status || !dev_outlen) return -1;
  return infl_launch_batch(dev_src, dev_src_off, dev_dst, dev_dst_off, flags, n, dev_status,
                           dev_outlen, nullptr, nullptr, g_lanes.load(), g_variant.load(), nullptr, hip_stream, nullptr);
}

// infl_batch_dev with the batch put into size classes first (a counting sort on the device), so that streams of
// widely different sizes side by side do not slow the decode down.
int infl_batch_dev_sorted(const void *dev_src, const uint64_t *dev_src_off, void *dev_dst,
                          const uint64_t *dev_dst_off, int flags, size_t n, int32_t *dev_status,
                          uint32_t *dev_outlen, void *dev_work, void *hip_stream) {
  if (n == 0) return 0;
  if (!dev_src_off || !dev_dst_off || !dev_status || !dev_outlen || !dev_work || n > 0xffffffffull) return -1;
  if (infl_launch_order(dev_src_off, dev_dst_off, n, dev_work, hip_stream) != 0) return -1;
  return infl_launch_batch_ordered(dev_src, dev_src_off, dev_dst, dev_dst_off, flags, n, dev_status, dev_outlen,
                                   nullptr, nullptr, g_lanes.load(), g_variant.load(), nullptr, hip_stream, nullptr,
                                   (const uint32_t *)dev_work + 256);
}

// Same with the per-stream detail codes (INFL_D_*); for tests and diagnosis.
int infl_batch_dev_detail(const void *dev_src, const uint64_t *dev_src_off, void *dev_dst,
                          const uint64_t *dev_dst_off, int flags, size_t n, int32_t *dev_status,
                          uint32_t *dev_outlen, int32_t *dev_detail, void *hip_stream) {
  if (n == 0) return 0;
  if (!dev_src_off || !dev_dst_off || !dev_status || !dev_outlen) return -1;
  return infl_launch_batch(dev_src, dev_src_off, dev_dst, dev_dst_off, flags, n, dev_status,
                           dev_outlen, dev_detail, nullptr, g_lanes.load(), g_variant.load(), nullptr, hip_stream, nullptr);
}

int infl_batch_dev_checked(const void *dev_src, const uint64_t *dev_src_off, void *dev_dst,
                           const uint64_t *dev_dst_off, int check, size_t n, int32_t *dev_status,
                           uint32_t *dev_outlen, int32_t *dev_detail, void *dev_work, void *hip_stream) {
  if (n == 0) return 0;
  if (!dev_src_off || !dev_dst_off || !dev_status || !dev_outlen || !dev_work) return -1;
  if (check != INFL_CHECK_ADLER32 && check != INFL_CHECK_GZIP) return -1;
  return infl_launch_checked(dev_src, dev_src_off, dev_dst, dev_dst_off, check, n, dev_status, dev_outlen,
                             dev_detail, dev_work, g_lanes.load(), g_variant.load(), nullptr, hip_stream);
}

int infl_png_unfilter_dev(const void *dev_in, const uint64_t *dev_in_off, const uint32_t *dev_in_len,
                          const int32_t *dev_in_status, void *dev_out, const uint64_t *dev_out_off,
                          const InflPngGeom *dev_geom, size_t n, int32_t *dev_status, void *hip_stream) {
  if (n == 0) return 0;
  if (!dev_in || !dev_in_off || !dev_out || !dev_out_off || !dev_geom || !dev_status) return -1;
  return infl_launch_png_unfilter(dev_in, dev_in_off, dev_in_len, dev_in_status, dev_out, dev_out_off, dev_geom, n,
                                  dev_status, hip_stream, ~0ull, ~0ull);
}

int infl_png_unfilter_dev_checked(const void *dev_in, uint64_t in_bytes, const uint64_t *dev_in_off,
                                  const uint32_t *dev_in_len, const int32_t *dev_in_status, void *dev_out,
                                  uint64_t out_bytes, const uint64_t *dev_out_off, const InflPngGeom *dev_geom, size_t n,
                                  int32_t *dev_status, void *hip_stream) {
  if (n == 0) return 0;
  if (!dev_in || !dev_in_off || !dev_out || !dev_out_off || !dev_geom || !dev_status) return -1;
  return infl_launch_png_unfilter(dev_in, dev_in_off, dev_in_len, dev_in_status, dev_out, dev_out_off, dev_geom, n,
                                  dev_status, hip_stream, in_bytes, out_bytes);
}

int infl_gather_dev(const void *dev_blob, const uint64_t *dev_piece_off, const uint32_t *dev_piece_len,
                    const uint64_t *dev_piece_dst, size_t n_pieces, void *dev_src, void *hip_stream) {
  if (n_pieces == 0) return 0;
  if (!dev_blob || !dev_piece_off || !dev_piece_len || !dev_piece_dst || !dev_src) return -1;
  return infl_launch_gather(dev_blob, dev_piece_off, dev_piece_len, dev_piece_dst, n_pieces, dev_src, hip_stream,
                            ~0ull, ~0ull, nullptr);
}

int infl_gather_dev_checked(const void *dev_blob, uint64_t blob_bytes, const uint64_t *dev_piece_off,
                            const uint32_t *dev_piece_len, const uint64_t *dev_piece_dst, size_t n_pieces, void *dev_src,
                            uint64_t src_bytes, int32_t *dev_piece_status, void *hip_stream) {
  if (n_pieces == 0) return 0;
  if (!dev_blob || !dev_piece_off || !dev_piece_len || !dev_piece_dst || !dev_src) return -1;
  return infl_launch_gather(dev_blob, dev_piece_off, dev_piece_len, dev_piece_dst, n_pieces, dev_src, hip_stream,
                            blob_bytes, src_bytes, dev_piece_status);
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

  std::lock_guard<std::mutex> lock(g_mu[dev]);
  Workspace &w = g_ws[dev];
  const int lanes = g_lanes.load(), variant = g_variant.load();
  const size_t src_bytes = (size_t)src_off[n], dst_bytes = (size_t)dst_off[n];
  const size_t off_bytes = (n + 1) * sizeof(uint64_t);

  if (!dev_reserve(w.src, src_bytes + 16) || !dev_reserve(w.dst, dst_bytes + 16) ||
      !dev_reserve(w.soff, off_bytes) || !dev_reserve(w.doff, off_bytes) ||
      !dev_reserve(w.status, n * 4) || !dev_reserve(w.outlen, n * 4) ||
      !dev_reserve(w.detail, n * 4) || !dev_reserve(w.inpos, n * 4) ||
      !host_reserve(w.hdst, dst_bytes + 16) || !host_reserve(w.hmeta, 4 * n * 4))
    return -12;
  // (room for the size-class order of a batch of many streams, or of each of its slices; the decode does without)
  const bool have_order = n >= kOrderMinStreams && n <= 0xffffffffull &&
                          dev_reserve(w.order, INFL_ORDER_WORK_BYTES(n) + 1024u * kPipeSlices);

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
  // first stream index at or behind which `want` destination bytes of [lo, n) lie
  auto cut_at = [&](uint64_t want, size_t lo) {
    size_t hi = (size_t)(std::lower_bound(dst_off, dst_off + n + 1, want) - dst_off);
    return std::min(std::max(hi, lo), n);
  };
  // one core copies about 10 GB/s, the link five times that: `threads` equal byte shares of
  // [lo, hi), cut at stream boundaries
  auto scatter_threads = [&](std::vector<std::thread> &pool, size_t lo, size_t hi, size_t threads) {
    const uint64_t b0 = dst_off[lo], b1 = dst_off[hi];
    for (size_t t = 1; t <= threads && lo < hi; t++) {
      const size_t cut = t < threads ? std::min(cut_at(b0 + (b1 - b0) / threads * t, lo), hi) : hi;
      if (cut > lo) {
        try {
          pool.emplace_back(scatter, lo, cut);
        } catch (...) {                                  // no thread to be had: this one does the share
          scatter(lo, cut);
        }
      }
      lo = cut;
    }
  };

  std::vector<std::thread> pool;
  bool ok = true;
  if (dst_bytes < kPipeMinBytes || n < 2 * kPipeMinStreams || !pipe_ready(dev)) {
    // small batch: upload, then decode (the kernels write the output to pinned memory), on the default stream
    hipStream_t st = nullptr;
    if (src_bytes) ok &= hipMemcpyAsync(w.src.p, src, src_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
    ok &= hipMemcpyAsync(w.soff.p, src_off, off_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
    ok &= hipMemcpyAsync(w.doff.p, dst_off, off_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
    if (!ok) {
      (void)hipStreamSynchronize(st);
      (void)hipGetLastError();
      return -1;
    }
    const uint32_t *order = nullptr;
    if (have_order &&
        infl_launch_order((const uint64_t *)w.soff.p, (const uint64_t *)w.doff.p, n, w.order.p, st) == 0)
      order = (const uint32_t *)w.order.p + 256;
    if (infl_launch_batch_ordered(w.src.p, (const uint64_t *)w.soff.p, w.dst.p, (const uint64_t *)w.doff.p,
                                  flags, n, (int32_t *)w.status.p, (uint32_t *)w.outlen.p,
                                  detail ? (int32_t *)w.detail.p : nullptr, inpos ? (uint32_t *)w.inpos.p : nullptr,
                                  lanes, variant, w.hdst.p, st, nullptr, order) != 0) {
      (void)hipStreamSynchronize(st);                    // the uploads queued above still read the caller's buffers
      return -1;
    }
    ok &= hipMemcpyAsync(status, w.status.p, n * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
    ok &= hipMemcpyAsync(outlen, w.outlen.p, n * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
    if (detail) ok &= hipMemcpyAsync(detail, w.detail.p, n * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
    if (inpos) ok &= hipMemcpyAsync(inpos, w.inpos.p, n * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
    ok &= hipStreamSynchronize(st) == hipSuccess;    // (the kernels wrote the output bytes to w.hdst themselves)
    if (!ok) {
      (void)hipGetLastError();
      return -1;
    }
    if (dst_bytes >= (32u << 20) && n >= 64) {
      scatter_threads(pool, 0, n, 8);
      for (auto &th : pool) th.join();
    } else {
      scatter(0, n);
    }
    return 0;
  }

  // Large batch: slices of whole streams go through upload -> decode on a few streams in
  // turn; the decode kernels write every finished stream's bytes to the pinned buffer
  // themselves (the `mirror` of infl_launch_batch), so the link carries data both ways while
  // the kernels run, and host threads hand a slice's bytes to the caller while the next
  // slices are still on their way.  The caller's pageable input is copied into pinned memory
  // by a few threads, slice by slice (the runtime's own staging of pageable memory moves
  // about 7 GB/s, one seventh of the link); the result arrays come back through pinned
  // memory as well, so that no call here holds this thread until a slice's kernels are done.
  Pipe      &pp     = g_pipe[dev];
  bool staged;                                         // (the gather buffer of infl_rt_pinned_acquire is pinned already)
  {
    std::lock_guard<std::mutex> pin_lock(g_pin_mu);
    staged = src != g_pin.p;
  }
  if (staged && !host_reserve(w.hsrc, src_bytes + 16)) return -12;
  const uint8_t *from = staged ? (const uint8_t *)w.hsrc.p : (const uint8_t *)src;
  uint32_t      *hm   = (uint32_t *)w.hmeta.p;
  size_t         cuts[kPipeSlices + 1];
  int            nsl = 0, done = 0;
  cuts[0] = 0;
  // hands the slices whose download has completed to scatter threads (all of them if `wait`)
  auto collect = [&](bool wait, bool last_too) {
    while (ok && done < nsl) {
      const hipError_t e = wait ? hipEventSynchronize(pp.ev[done]) : hipEventQuery(pp.ev[done]);
      if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        return;
      }
      if (e != hipSuccess) {
        ok = false;
        return;
      }
      const size_t lo = cuts[done], hi = cuts[done + 1], cnt = (hi - lo) * 4;
      memcpy(status + lo, hm + lo, cnt);
      memcpy(outlen + lo, hm + n + lo, cnt);
      if (detail) memcpy(detail + lo, hm + 2 * n + lo, cnt);
      if (inpos) memcpy(inpos + lo, hm + 3 * n + lo, cnt);
      scatter_threads(pool, lo, hi, last_too && hi == n ? 8 : 2);
      done++;
    }
  };
  ok &= hipMemcpyAsync(w.soff.p, src_off, off_bytes, hipMemcpyHostToDevice, pp.up) == hipSuccess;
  ok &= hipMemcpyAsync(w.doff.p, dst_off, off_bytes, hipMemcpyHostToDevice, pp.up) == hipSuccess;
  const uint64_t slice_bytes = std::max<uint64_t>(kPipeMinBytes * 3 / 4, dst_bytes / (kPipeSlices - 1));
  while (ok && cuts[nsl] < n && nsl < kPipeSlices) {
    const size_t lo = cuts[nsl];
    size_t       hi = nsl + 1 == kPipeSlices ? n : cut_at(dst_off[lo] + slice_bytes, lo);
    hi = std::min(std::max(hi, std::min(lo + kPipeMinStreams, n)), n);
    hipStream_t    st = pp.st[nsl % kPipeStreams];
    const uint64_t s0 = src_off[lo], s1 = src_off[hi], d0 = dst_off[lo], d1 = dst_off[hi];
    if (s1 > s0) {
      if (staged) {
        constexpr int T   = 4;
        const size_t  len = s1 - s0;
        std::thread   helper[T - 1];
        auto part = [&](int t) {
          const size_t a = len / T * t, b = t + 1 == T ? len : len / T * (t + 1);
          memcpy((uint8_t *)w.hsrc.p + s0 + a, (const uint8_t *)src + s0 + a, b - a);
        };
        const bool many = len >= (1u << 20);
        for (int t = 1; t < T && many; t++) {
          try {
            helper[t - 1] = std::thread(part, t);
          } catch (...) {
            part(t);
          }
        }
        if (many) part(0); else memcpy((uint8_t *)w.hsrc.p + s0, (const uint8_t *)src + s0, len);
        for (int t = 1; t < T && many; t++)
          if (helper[t - 1].joinable()) helper[t - 1].join();
      }
      ok &= hipMemcpyAsync((uint8_t *)w.src.p + s0, from + s0, s1 - s0, hipMemcpyHostToDevice, pp.up) == hipSuccess;
    }
    ok &= hipEventRecord(pp.evu[nsl], pp.up) == hipSuccess;
    ok &= hipStreamWaitEvent(st, pp.evu[nsl], 0) == hipSuccess;
    // (a slice of many streams is put into size classes first; slice k's counters and indices lie at 1024 k + 4 lo)
    const uint32_t *order = nullptr;
    if (ok && have_order && hi - lo >= kOrderMinStreams) {
      uint8_t *work = (uint8_t *)w.order.p + 1024u * (size_t)nsl + 4u * lo;
      if (infl_launch_order((const uint64_t *)w.soff.p + lo, (const uint64_t *)w.doff.p + lo, hi - lo, work, st) == 0)
        order = (const uint32_t *)work + 256;
    }
    ok = ok && infl_launch_batch_ordered(w.src.p, (const uint64_t *)w.soff.p + lo, w.dst.p, (const uint64_t *)w.doff.p + lo,
                                         flags, hi - lo, (int32_t *)w.status.p + lo, (uint32_t *)w.outlen.p + lo,
                                         detail ? (int32_t *)w.detail.p + lo : nullptr,
                                         inpos ? (uint32_t *)w.inpos.p + lo : nullptr, lanes, variant,
                                         w.hdst.p, st, nullptr, order) == 0;
    const size_t cnt = (hi - lo) * 4;
    ok &= hipMemcpyAsync(hm + lo, (int32_t *)w.status.p + lo, cnt, hipMemcpyDeviceToHost, st) == hipSuccess;
    ok &= hipMemcpyAsync(hm + n + lo, (uint32_t *)w.outlen.p + lo, cnt, hipMemcpyDeviceToHost, st) == hipSuccess;
    if (detail) ok &= hipMemcpyAsync(hm + 2 * n + lo, (int32_t *)w.detail.p + lo, cnt, hipMemcpyDeviceToHost, st) == hipSuccess;
    if (inpos) ok &= hipMemcpyAsync(hm + 3 * n + lo, (uint32_t *)w.inpos.p + lo, cnt, hipMemcpyDeviceToHost, st) == hipSuccess;
    ok &= hipEventRecord(pp.ev[nsl], st) == hipSuccess;
    cuts[++nsl] = hi;
    collect(false, false);
  }
  collect(true, true);
  if (!ok) {                                           // let everything queued drain before the buffers are reused
    (void)hipStreamSynchronize(pp.up);
    for (auto &q : pp.st) (void)hipStreamSynchronize(q);
    (void)hipGetLastError();
  }
  for (auto &th : pool) th.join();
  return ok && cuts[nsl] == n ? 0 : -1;
}

// ---- infl_stream(): one device context per stream object ------------------------------------
// The input seen so far and the output produced so far stay on the device between calls.  A call
// is one launch and one wait: the new input bytes lie in a pinned buffer the kernel reads itself,
// and the kernel stores the new output bytes and its state to pinned memory too (a feed or an
// output larger than the staging takes the copy engine for the rest).  Calls on one GPU take
// turns (the per-device lock): they share the library's stream there.
constexpr size_t kStageIn = 64u << 10, kStageOut = 64u << 10;
struct StreamCtx {
  void       *d_src = nullptr, *d_dst = nullptr;
  size_t      src_cap = 0, src_up = 0, dst_cap = 0;
  InflResume *d_state = nullptr;
  void       *d_tables = nullptr;         // the decode tables of the block the last call stopped in
  uint8_t    *pin = nullptr;              // [InflResume | pad to 64][kStageIn input][kStageOut output]
  int         dev = 0;
  bool        restart = false;            // the next step begins the stream again (state cleared on the device first)
};

void infl_rt_stream_free(void *ctx) {
  StreamCtx *c = (StreamCtx *)ctx;
  if (!c) return;
  if (c->d_src) (void)hipFree(c->d_src);
  if (c->d_dst) (void)hipFree(c->d_dst);
  if (c->d_state) (void)hipFree(c->d_state);
  if (c->d_tables) (void)hipFree(c->d_tables);
  if (c->pin) (void)hipHostFree(c->pin);
  delete c;
}

// Called by infl_reset_pool(): the buffers stay (a stream object that is reset for every input would otherwise pay
// three device allocations and a pinned one per input); the state is cleared by the next step, on its stream.
void infl_rt_stream_restart(void *ctx) {
  StreamCtx *c = (StreamCtx *)ctx;
  if (c) c->restart = true;
}

int infl_rt_stream_step(void **ctxp, const void *acc, size_t acc_len, void *dst, uint32_t dstlen,
                        uint32_t dstpos, int zlib, InflResume *out) {
  if (!ctxp || !out || acc_len > 0xffffffffu) return -1;
  if (!device_ready()) return -1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return -1;
  std::lock_guard<std::mutex> lock(g_mu[dev]);
  hipStream_t st = aux_stream_on(dev);
  if (!st) return -1;
  StreamCtx *c = (StreamCtx *)*ctxp;
  if (c && c->restart && c->dev != dev) {             // a fresh start on another GPU: nothing of the old context serves
    infl_rt_stream_free(c);
    *ctxp = c = nullptr;
  }
  if (c && c->dev != dev) return -1;                  // a stream stays on the GPU it began on
  if (c && c->restart) {
    if (hipMemsetAsync(c->d_state, 0, sizeof(InflResume), st) != hipSuccess) {
      (void)hipGetLastError();
      return -1;
    }
    c->src_up  = 0;
    c->restart = false;
  }
  if (!c) {
    c = new (std::nothrow) StreamCtx();
    if (!c) return -12;
    c->dev = dev;
    if (hipMalloc((void **)&c->d_state, sizeof(InflResume)) != hipSuccess ||
        hipMalloc(&c->d_tables, INFL_RESUME_TABLE_BYTES) != hipSuccess ||
        hipMemsetAsync(c->d_state, 0, sizeof(InflResume), st) != hipSuccess ||
        hipHostMalloc((void **)&c->pin, 64 + kStageIn + kStageOut, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      infl_rt_stream_free(c);
      return -12;
    }
    *ctxp = c;
  }
  InflResume *hstate = (InflResume *)c->pin;
  uint8_t    *hin = c->pin + 64, *hout = hin + kStageIn;
  if (acc_len + 16 > c->src_cap) {                    // grow the input arena: everything goes up again
    const size_t want = std::max(acc_len + 16 + acc_len / 2, (size_t)4096);
    void *n = nullptr;
    if (hipMalloc(&n, want) != hipSuccess) {
      (void)hipGetLastError();
      return -12;
    }
    if (c->d_src) {
      (void)hipStreamSynchronize(st);
      (void)hipFree(c->d_src);
    }
    c->d_src   = n;
    c->src_cap = want;
    c->src_up  = 0;
  }
  if ((size_t)dstlen + 16 > c->dst_cap) {             // a larger output buffer (infl_resize_output): keep what is there
    const size_t want = (size_t)dstlen + 16;
    void *n = nullptr;
    if (hipMalloc(&n, want) != hipSuccess) {
      (void)hipGetLastError();
      return -12;
    }
    bool ok = true;
    if (c->d_dst && dstpos)
      ok = hipMemcpyAsync(n, c->d_dst, std::min((size_t)dstpos, c->dst_cap), hipMemcpyDeviceToDevice, st) == hipSuccess &&
           hipStreamSynchronize(st) == hipSuccess;
    if (c->d_dst) (void)hipFree(c->d_dst);
    c->d_dst   = n;
    c->dst_cap = want;
    if (!ok) return -1;
  }
  bool           ok = true;
  const uint8_t *fresh = nullptr;
  const size_t   fresh_at = c->src_up, fresh_len = acc_len - c->src_up;
  if (fresh_len > kStageIn) {                         // a large feed: the copy engine
    ok &= hipMemcpyAsync((uint8_t *)c->d_src + fresh_at, (const uint8_t *)acc + fresh_at, fresh_len,
                         hipMemcpyHostToDevice, st) == hipSuccess;
  } else if (fresh_len) {
    memcpy(hin, (const uint8_t *)acc + fresh_at, fresh_len);
    fresh = hin;
  }
  c->src_up = acc_len;
  ok = ok && infl_launch_resume(c->d_src, (uint32_t)acc_len, c->d_dst, dstlen, zlib ? 1 : 0, c->d_state, fresh,
                                (uint32_t)fresh_at, hout, (uint32_t)kStageOut, hstate, c->d_tables, st) == 0;
  ok = ok && hipStreamSynchronize(st) == hipSuccess;
  if (ok) {
    *out = *hstate;
    ok = out->reserved == dstpos && out->pos >= dstpos && out->pos <= dstlen;   // (the device began where the host stands)
  }
  if (ok && out->pos > dstpos) {
    const size_t got = out->pos - dstpos, staged = std::min(got, kStageOut);
    memcpy((uint8_t *)dst + dstpos, hout, staged);
    if (got > staged)
      ok = hipMemcpyAsync((uint8_t *)dst + dstpos + staged, (const uint8_t *)c->d_dst + dstpos + staged, got - staged,
                          hipMemcpyDeviceToHost, st) == hipSuccess &&
           hipStreamSynchronize(st) == hipSuccess;
  }
  if (!ok) {
    (void)hipStreamSynchronize(st);
    (void)hipGetLastError();
    return -1;
  }
  return 0;
}

// Clears what lies behind every stream's output in the destination arena of infl_batch_noalloc (one
// wavefront per stream), so that the one copy that hands the arena back carries nothing of earlier calls.
__global__ __launch_bounds__(64) void infl_clear_tail_kernel(unsigned char *__restrict__ dst,
                                                            const uint64_t *__restrict__ dst_off,
                                                            const uint32_t *__restrict__ outlen, uint64_t n) {
  for (uint64_t i = blockIdx.x; i < n; i += gridDim.x) {
    const uint64_t d0 = dst_off[i], cap = dst_off[i + 1] - d0;
    const uint64_t from = outlen[i] < cap ? outlen[i] : cap;
    for (uint64_t k = from + threadIdx.x; k < cap; k += 64) dst[d0 + k] = 0;
  }
}

int infl_batch_noalloc(const void *src, const uint64_t *src_off, void *dst, const uint64_t *dst_off,
                       int flags, size_t n, int32_t *status, uint32_t *outlen) {
  if (n == 0) return 0;
  if (!src_off || !dst_off || !status || !outlen) return -1;
  const uint64_t src_bytes = src_off[n], dst_bytes = dst_off[n];
  if (n > INFL_NOALLOC_MAX_STREAMS || src_bytes > INFL_NOALLOC_MAX_SRC || dst_bytes > INFL_NOALLOC_MAX_DST) {
    fprintf(stderr, "defl: infl_batch_noalloc: %zu streams, %llu source and %llu destination bytes exceed the fixed "
                    "workspace (%u streams, %u and %u bytes)\n", n, (unsigned long long)src_bytes,
            (unsigned long long)dst_bytes, (unsigned)INFL_NOALLOC_MAX_STREAMS, (unsigned)INFL_NOALLOC_MAX_SRC,
            (unsigned)INFL_NOALLOC_MAX_DST);
    return -105;                                       // UNZ_EFULL
  }
  if ((src_bytes && !src) || (dst_bytes && !dst)) return -1;
  for (size_t i = 0; i < n; i++)                       // the regions of the one array lie side by side: a stream's
    if (src_off[i] > src_off[i + 1] || dst_off[i] > dst_off[i + 1]) return -1;   // range must not reach into another
  if (!device_ready()) return -1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return -1;

  // Calls on one GPU share that GPU's array: the lock keeps host threads apart (a caller that
  // bypasses it, e.g. from two processes' worth of streams, would race).  Nothing in the array
  // has to start at zero: every stream's status, length and bytes are written by the kernels
  // before they are read back, and what lies behind a stream's bytes is cleared before the
  // arena goes back, whatever the last call left behind.
  std::lock_guard<std::mutex> lock(g_mu[dev]);
  unsigned char *base = scratch_on(dev);
  if (!base) return -1;
  const size_t   off_bytes = (n + 1) * sizeof(uint64_t);
  hipStream_t    st = g_aux_stream[dev];               // made by infl_gpu_init() (nothing is created here); without
  bool ok = true;                                      // that call the null stream serves
  if (src_bytes) ok &= hipMemcpyAsync(base + kNaSrcAt, src, src_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
  ok &= hipMemcpyAsync(base + kNaSoffAt, src_off, off_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
  ok &= hipMemcpyAsync(base + kNaDoffAt, dst_off, off_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
  ok = ok && infl_launch_batch(base + kNaSrcAt, (const uint64_t *)(base + kNaSoffAt), base + kNaDstAt,
                               (const uint64_t *)(base + kNaDoffAt), flags == 1 ? 1 : 0, n,
                               (int32_t *)(base + kNaStatusAt), (uint32_t *)(base + kNaOutlenAt), nullptr, nullptr,
                               g_lanes.load(), g_variant.load(), nullptr, st, nullptr) == 0;
  if (ok) {
    hipLaunchKernelGGL(infl_clear_tail_kernel, dim3((unsigned)n), dim3(64), 0, st, base + kNaDstAt,
                       (const uint64_t *)(base + kNaDoffAt), (const uint32_t *)(base + kNaOutlenAt), (uint64_t)n);
    ok &= hipGetLastError() == hipSuccess;
    ok &= hipMemcpyAsync(status, base + kNaStatusAt, n * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
    ok &= hipMemcpyAsync(outlen, base + kNaOutlenAt, n * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
    // the whole destination arena in one copy (zero behind every stream's bytes)
    if (dst_bytes) ok &= hipMemcpyAsync(dst, base + kNaDstAt, dst_bytes, hipMemcpyDeviceToHost, st) == hipSuccess;
  }
  const bool drained = hipStreamSynchronize(st) == hipSuccess;
  if (!ok || !drained) {
    (void)hipGetLastError();
    return -1;
  }
  return 0;
}

// Host-memory form of infl_batch_dev_checked: upload, decode + integrity pass, download (one launch,
// the device workspace of infl_batch(); only the bytes a stream produced are handed back).
int infl_batch_checked(const void *src, const uint64_t *src_off, void *dst, const uint64_t *dst_off,
                       int check, size_t n, int32_t *status, uint32_t *outlen, int32_t *detail) {
  if (n == 0) return 0;
  if (!src_off || !dst_off || !status || !outlen) return -1;
  if (check != INFL_CHECK_ADLER32 && check != INFL_CHECK_GZIP) return -1;
  if (!device_ready()) return -1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return -1;
  std::lock_guard<std::mutex> lock(g_mu[dev]);
  Workspace &w = g_ws[dev];
  const size_t src_bytes = (size_t)src_off[n], dst_bytes = (size_t)dst_off[n];
  const size_t off_bytes = (n + 1) * sizeof(uint64_t);
  if (!dev_reserve(w.src, src_bytes + 16) || !dev_reserve(w.dst, dst_bytes + 16) || !dev_reserve(w.soff, off_bytes) ||
      !dev_reserve(w.doff, off_bytes) || !dev_reserve(w.status, n * 4) || !dev_reserve(w.outlen, n * 4) ||
      !dev_reserve(w.detail, n * 4) || !dev_reserve(w.inpos, n * 8) || !host_reserve(w.hdst, dst_bytes + 16))
    return -12;
  hipStream_t st = nullptr;
  bool ok = true;
  if (src_bytes) ok &= hipMemcpyAsync(w.src.p, src, src_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
  ok &= hipMemcpyAsync(w.soff.p, src_off, off_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
  ok &= hipMemcpyAsync(w.doff.p, dst_off, off_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
  ok = ok && infl_launch_checked(w.src.p, (const uint64_t *)w.soff.p, w.dst.p, (const uint64_t *)w.doff.p, check, n,
                                 (int32_t *)w.status.p, (uint32_t *)w.outlen.p, (int32_t *)w.detail.p, w.inpos.p,
                                 g_lanes.load(), g_variant.load(), nullptr, st) == 0;
  if (ok) {
    ok &= hipMemcpyAsync(status, w.status.p, n * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
    ok &= hipMemcpyAsync(outlen, w.outlen.p, n * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
    if (detail) ok &= hipMemcpyAsync(detail, w.detail.p, n * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
    if (dst_bytes) ok &= hipMemcpyAsync(w.hdst.p, w.dst.p, dst_bytes, hipMemcpyDeviceToHost, st) == hipSuccess;
  }
  const bool drained = hipStreamSynchronize(st) == hipSuccess;
  if (!ok || !drained) {
    (void)hipGetLastError();
    return -1;
  }
  for (size_t i = 0; i < n; i++) {
    size_t len = outlen[i], cap = (size_t)(dst_off[i + 1] - dst_off[i]);
    if (len > cap) len = cap;
    if (len) memcpy((uint8_t *)dst + dst_off[i], (const uint8_t *)w.hdst.p + dst_off[i], len);
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

