"""ctypes binding of libdefl_amd.so, mirroring the C API one to one.

`load()` raises if the shared library has not been built: the engine has no fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get("DEFL_AMD_LIB") or os.path.join(HERE, "libdefl_amd.so")   # override: diagnostic builds
CSRC = os.path.join(HERE, "csrc")

UNZ_UNFINISHED, UNZ_NOOP, UNZ_OK, UNZ_ERR, UNZ_EFULL = 2, 1, 0, -1, -105
INFL_ZLIB = 1
D_OK, D_INVALID, D_NEED_INPUT, D_FULL, D_INTERNAL, D_CHECKSUM = 0, 1, 2, 3, 4, 5
CHECK_ADLER32, CHECK_GZIP = 1, 2

# every symbol include/defl/infl.h and include/defl/infl_batch.h declare
EXPORTS = [
    "infl_init", "infl_reset", "infl_resize_output", "infl_include", "infl", "infl_reset_pool",
    "infl_destroy", "infl_stream", "infl_output_pos", "infl_input_pos",
    "infl_gpu_init", "infl_gpu_device_count", "infl_batch_dev", "infl_batch_dev_detail", "infl_batch",
    "infl_many", "infl_gpu_malloc", "infl_gpu_free", "infl_gpu_upload", "infl_gpu_download",
    "infl_gpu_sync", "infl_gpu_set_lanes_per_stream", "infl_gpu_set_table_variant",
    "infl_batch_noalloc", "infl_batch_dev_checked", "infl_batch_checked", "infl_png_unfilter_dev", "infl_gather_dev",
    "infl_gather_dev_checked", "infl_png_unfilter_dev_checked", "infl_batch_dev_sorted",
]


def build(jobs: int = 8):
    """Compile the HIP kernels for gfx950 and link the shared library in-tree."""
    subprocess.check_call(["make", "-s", "-j", str(jobs), "-C", CSRC])
    return SO


class DeviceBatch:
    """A batch resident in device memory (arenas + offset tables + result arrays)."""

    def __init__(self, eng, src, src_off, dst_off):
        self.eng = eng
        self.n = len(src_off) - 1
        self.src_off = np.ascontiguousarray(src_off, dtype=np.uint64)
        self.dst_off = np.ascontiguousarray(dst_off, dtype=np.uint64)
        src = np.ascontiguousarray(src, dtype=np.uint8)
        self.src_bytes = int(self.src_off[-1])
        self.dst_bytes = int(self.dst_off[-1])
        m = eng.lib.infl_gpu_malloc
        self.d_src = m(self.src_bytes + 16)
        self.d_dst = m(self.dst_bytes + 16)
        self.d_soff = m(self.src_off.nbytes)
        self.d_doff = m(self.dst_off.nbytes)
        self.d_status = m(4 * max(1, self.n))
        self.d_outlen = m(4 * max(1, self.n))
        self.d_detail = m(4 * max(1, self.n))
        if not all((self.d_src, self.d_dst, self.d_soff, self.d_doff, self.d_status, self.d_outlen,
                    self.d_detail)):
            raise MemoryError("infl_gpu_malloc failed")
        up = eng.lib.infl_gpu_upload
        assert up(self.d_src, src.ctypes.data, self.src_bytes) == 0
        assert up(self.d_soff, self.src_off.ctypes.data, self.src_off.nbytes) == 0
        assert up(self.d_doff, self.dst_off.ctypes.data, self.dst_off.nbytes) == 0

    def run(self, flags=0, stream=None, detail=True):
        lib = self.eng.lib
        if detail:
            rc = lib.infl_batch_dev_detail(self.d_src, self.d_soff, self.d_dst, self.d_doff, flags, self.n,
                                           self.d_status, self.d_outlen, self.d_detail, stream)
        else:
            rc = lib.infl_batch_dev(self.d_src, self.d_soff, self.d_dst, self.d_doff, flags, self.n,
                                    self.d_status, self.d_outlen, stream)
        if rc != 0:
            raise RuntimeError("infl_batch_dev failed to launch")

    def results(self, want_dst=True):
        lib = self.eng.lib
        assert lib.infl_gpu_sync() == 0
        status = np.zeros(max(1, self.n), dtype=np.int32)
        outlen = np.zeros(max(1, self.n), dtype=np.uint32)
        detail = np.zeros(max(1, self.n), dtype=np.int32)
        dn = lib.infl_gpu_download
        assert dn(status.ctypes.data, self.d_status, 4 * self.n) == 0
        assert dn(outlen.ctypes.data, self.d_outlen, 4 * self.n) == 0
        assert dn(detail.ctypes.data, self.d_detail, 4 * self.n) == 0
        dst = None
        if want_dst:
            dst = np.zeros(max(1, self.dst_bytes), dtype=np.uint8)
            assert dn(dst.ctypes.data, self.d_dst, self.dst_bytes) == 0
        return dst, status[:self.n], outlen[:self.n], detail[:self.n]

    def free(self):
        for name in ("d_src", "d_dst", "d_soff", "d_doff", "d_status", "d_outlen", "d_detail"):
            p = getattr(self, name, None)
            if p:
                self.eng.lib.infl_gpu_free(p)
                setattr(self, name, None)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    def __init__(self, lib):
        self.lib = lib
        vp, u32, i32, sz = C.c_void_p, C.c_uint32, C.c_int, C.c_size_t
        sig = {
            "infl_init": (vp, [vp, u32, i32]),
            "infl_reset": (None, [vp, vp, u32, i32]),
            "infl_resize_output": (i32, [vp, vp, u32]),
            "infl_include": (None, [vp, vp, u32]),
            "infl": (i32, [vp]),
            "infl_reset_pool": (None, [vp]),
            "infl_destroy": (None, [vp]),
            "infl_stream": (i32, [vp, vp, u32]),
            "infl_output_pos": (u32, [vp]),
            "infl_input_pos": (u32, [vp]),
            "infl_gpu_init": (i32, [i32]),
            "infl_gpu_device_count": (i32, []),
            "infl_batch_dev": (i32, [vp, vp, vp, vp, i32, sz, vp, vp, vp]),
            "infl_batch_dev_detail": (i32, [vp, vp, vp, vp, i32, sz, vp, vp, vp, vp]),
            "infl_batch_dev_sorted": (i32, [vp, vp, vp, vp, i32, sz, vp, vp, vp, vp]),
            "infl_batch": (i32, [vp, vp, vp, vp, i32, sz, vp, vp]),
            "infl_batch_noalloc": (i32, [vp, vp, vp, vp, i32, sz, vp, vp]),
            "infl_batch_dev_checked": (i32, [vp, vp, vp, vp, i32, sz, vp, vp, vp, vp, vp]),
            "infl_batch_checked": (i32, [vp, vp, vp, vp, i32, sz, vp, vp, vp]),
            "infl_png_unfilter_dev": (i32, [vp, vp, vp, vp, vp, vp, vp, sz, vp, vp]),
            "infl_gather_dev": (i32, [vp, vp, vp, vp, sz, vp, vp]),
            "infl_gather_dev_checked": (i32, [vp, C.c_uint64, vp, vp, vp, sz, vp, C.c_uint64, vp, vp]),
            "infl_png_unfilter_dev_checked": (i32, [vp, C.c_uint64, vp, vp, vp, vp, C.c_uint64, vp, vp, sz, vp, vp]),
            "infl_many": (i32, [vp, sz, vp]),
            "infl_gpu_malloc": (vp, [sz]),
            "infl_gpu_free": (None, [vp]),
            "infl_gpu_upload": (i32, [vp, vp, sz]),
            "infl_gpu_download": (i32, [vp, vp, sz]),
            "infl_gpu_sync": (i32, []),
            "infl_gpu_set_lanes_per_stream": (i32, [i32]),
            "infl_gpu_set_table_variant": (i32, [i32]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args

    # -- the reference's one-call wrapper, infl_buf() (include/defl/infl.h:125-145) --
    def infl_buf(self, src: bytes, dstlen: int, flags: int = 0):
        """-> (rc, output position, bytes[0:position])"""
        src = bytes(src)
        sbuf = (C.c_uint8 * max(1, len(src))).from_buffer_copy(src or b"\0")
        dbuf = (C.c_uint8 * max(1, dstlen))()
        st = self.lib.infl_init(dbuf, dstlen, flags)
        if not st:
            return -1, 0, b""
        self.lib.infl_include(st, sbuf, len(src))
        rc = self.lib.infl(st)
        pos = self.lib.infl_output_pos(st)
        self.lib.infl_destroy(st)
        return rc, pos, bytes(dbuf[:pos])

    def batch_host(self, src, src_off, dst_off, flags=0):
        """infl_batch() on host arrays -> (dst arena, status, outlen)"""
        n = len(src_off) - 1
        src = np.ascontiguousarray(src, dtype=np.uint8)
        src_off = np.ascontiguousarray(src_off, dtype=np.uint64)
        dst_off = np.ascontiguousarray(dst_off, dtype=np.uint64)
        dst = np.zeros(max(1, int(dst_off[-1])), dtype=np.uint8)
        status = np.full(max(1, n), -99, dtype=np.int32)
        outlen = np.zeros(max(1, n), dtype=np.uint32)
        rc = self.lib.infl_batch(src.ctypes.data, src_off.ctypes.data, dst.ctypes.data, dst_off.ctypes.data,
                                 flags, n, status.ctypes.data, outlen.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"infl_batch failed: {rc}")
        return dst, status[:n], outlen[:n]

    def batch_checked(self, src, src_off, dst_off, check):
        """infl_batch_checked() on host arrays -> (dst arena, status, outlen, detail)"""
        n = len(src_off) - 1
        src = np.ascontiguousarray(src, dtype=np.uint8)
        src_off = np.ascontiguousarray(src_off, dtype=np.uint64)
        dst_off = np.ascontiguousarray(dst_off, dtype=np.uint64)
        dst = np.zeros(max(1, int(dst_off[-1])), dtype=np.uint8)
        status = np.full(max(1, n), -99, dtype=np.int32)
        outlen = np.zeros(max(1, n), dtype=np.uint32)
        detail = np.full(max(1, n), -99, dtype=np.int32)
        rc = self.lib.infl_batch_checked(src.ctypes.data, src_off.ctypes.data, dst.ctypes.data, dst_off.ctypes.data,
                                         check, n, status.ctypes.data, outlen.ctypes.data, detail.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"infl_batch_checked failed: {rc}")
        return dst, status[:n], outlen[:n], detail[:n]

    def gather_into(self, batch: "DeviceBatch", blob, piece_off, piece_len, piece_dst):
        """infl_gather_dev(): uploads `blob` and the three piece tables and gathers the pieces into the source arena of
        `batch` (whose src_off must describe the gathered streams).  Enqueues and synchronises."""
        lib = self.lib
        blob = np.ascontiguousarray(np.frombuffer(bytes(blob), dtype=np.uint8))
        po = np.ascontiguousarray(piece_off, dtype=np.uint64)
        pl = np.ascontiguousarray(piece_len, dtype=np.uint32)
        pd = np.ascontiguousarray(piece_dst, dtype=np.uint64)
        n = len(po)
        bufs = [lib.infl_gpu_malloc(max(16, a.nbytes + 16)) for a in (blob, po, pl, pd)]
        try:
            if not all(bufs):
                raise MemoryError("infl_gpu_malloc failed")
            for b, a in zip(bufs, (blob, po, pl, pd)):
                if a.nbytes:
                    assert lib.infl_gpu_upload(b, a.ctypes.data, a.nbytes) == 0
            if lib.infl_gather_dev(bufs[0], bufs[1], bufs[2], bufs[3], n, batch.d_src, None) != 0:
                raise RuntimeError("infl_gather_dev failed to launch")
            assert lib.infl_gpu_sync() == 0
        finally:
            for b in bufs:
                if b:
                    lib.infl_gpu_free(b)

    def png_unfilter(self, batch: "DeviceBatch", geom, chained=True):
        """infl_png_unfilter_dev() behind a decoded DeviceBatch: image i is the output of stream i, geom a list
        of (rows, rowbytes, bpp).  chained = pass the decode's outlen and status along.
        -> (list of per-image bytes, status)"""
        lib, n = self.lib, batch.n
        g = np.zeros((max(1, n), 4), dtype=np.uint32)
        for i, (rows, rowbytes, bpp) in enumerate(geom):
            g[i, :3] = rows, rowbytes, bpp
        ooff = np.zeros(n + 1, dtype=np.uint64)
        ooff[1:] = np.cumsum([int(r) * int(rb) for r, rb, _ in geom], dtype=np.uint64)
        total = int(ooff[-1])
        d_geom, d_ooff, d_out, d_st = (lib.infl_gpu_malloc(x) for x in (g.nbytes, ooff.nbytes, total + 16, 4 * max(1, n)))
        try:
            if not all((d_geom, d_ooff, d_out, d_st)):
                raise MemoryError("infl_gpu_malloc failed")
            assert lib.infl_gpu_upload(d_geom, g.ctypes.data, g.nbytes) == 0
            assert lib.infl_gpu_upload(d_ooff, ooff.ctypes.data, ooff.nbytes) == 0
            rc = lib.infl_png_unfilter_dev(batch.d_dst, batch.d_doff, batch.d_outlen if chained else None,
                                           batch.d_status if chained else None, d_out, d_ooff, d_geom, n, d_st, None)
            if rc != 0:
                raise RuntimeError("infl_png_unfilter_dev failed to launch")
            assert lib.infl_gpu_sync() == 0
            out = np.zeros(max(1, total), dtype=np.uint8)
            st = np.zeros(max(1, n), dtype=np.int32)
            assert lib.infl_gpu_download(out.ctypes.data, d_out, total) == 0
            assert lib.infl_gpu_download(st.ctypes.data, d_st, 4 * n) == 0
        finally:
            for p in (d_geom, d_ooff, d_out, d_st):
                if p:
                    lib.infl_gpu_free(p)
        return [out[int(ooff[i]):int(ooff[i + 1])].tobytes() for i in range(n)], st[:n]

    def device_batch(self, src, src_off, dst_off) -> DeviceBatch:
        return DeviceBatch(self, src, src_off, dst_off)


_cached = None


def load() -> Engine:
    global _cached
    if _cached is None:
        if not os.path.exists(SO):
            raise RuntimeError(f"{SO} is missing: build it with defl_amd.capi.build() "
                               "(make -C defl_amd/csrc); the engine has no fallback path")
        _cached = Engine(C.CDLL(SO))
    return _cached
