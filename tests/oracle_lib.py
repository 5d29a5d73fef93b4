"""ctypes binding of oracle/liboracle.so (the CPU checker; test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "liboracle.so")

OK, NOOP, UNFINISHED, ERR, EFULL = 0, 1, 2, -1, -105
D_OK, D_INVALID, D_NEED_INPUT, D_FULL = 0, 1, 2, 3
FLAG_ZLIB = 1


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        u8p, u32p, u64p, i32p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_uint64, C.c_int32))
        lib.oracle_infl_detail.restype = C.c_int
        lib.oracle_infl_detail.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int,
                                           u32p, C.POINTER(C.c_int)]
        lib.oracle_infl_batch.restype = C.c_int
        lib.oracle_infl_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        lib.oracle_infl_batch_as_is.restype = C.c_int
        lib.oracle_infl_batch_as_is.argtypes = lib.oracle_infl_batch.argtypes
        lib.oracle_table_build.restype = C.c_int
        lib.oracle_table_build.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.c_void_p, C.c_uint,
                                           C.c_uint, C.c_uint, C.c_int]

    def png_unfilter(self, data: bytes, rows: int, rowbytes: int, bpp: int):
        """-> (rc, unfiltered bytes)"""
        self.lib.oracle_png_unfilter.restype = C.c_int
        self.lib.oracle_png_unfilter.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        ibuf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(bytes(data) or b"\0")
        obuf = (C.c_uint8 * max(1, rows * rowbytes))()
        rc = self.lib.oracle_png_unfilter(ibuf, obuf, rows, rowbytes, bpp)
        return rc, bytes(obuf[: rows * rowbytes])

    def infl(self, src: bytes, dstlen: int, flags: int = 0):
        """-> (rc, dstpos, out bytes[0:dstpos], detail)"""
        src = bytes(src)
        sbuf = (C.c_uint8 * max(1, len(src))).from_buffer_copy(src or b"\0")
        dbuf = (C.c_uint8 * max(1, dstlen))()
        pos, detail = C.c_uint32(0), C.c_int(0)
        rc = self.lib.oracle_infl_detail(sbuf, len(src), dbuf, dstlen, flags, C.byref(pos), C.byref(detail))
        return rc, pos.value, bytes(dbuf[:pos.value]), detail.value

    def infl_batch(self, src: np.ndarray, src_off: np.ndarray, dst_off: np.ndarray, flags: int = 0,
                   nthreads: int = 8, as_is: bool = False):
        """src: uint8 arena, offsets uint64[n+1] -> (dst arena, status int32[n], outlen uint32[n]).
        as_is: with the per-call allocations of the reference's infl_buf() around every stream (for timing)."""
        n = len(src_off) - 1
        src = np.ascontiguousarray(src, dtype=np.uint8)
        src_off = np.ascontiguousarray(src_off, dtype=np.uint64)
        dst_off = np.ascontiguousarray(dst_off, dtype=np.uint64)
        dst = np.zeros(max(1, int(dst_off[-1])), dtype=np.uint8)
        status = np.zeros(max(1, n), dtype=np.int32)
        outlen = np.zeros(max(1, n), dtype=np.uint32)
        fn = self.lib.oracle_infl_batch_as_is if as_is else self.lib.oracle_infl_batch
        rc = fn(src.ctypes.data, src_off.ctypes.data, dst.ctypes.data, dst_off.ctypes.data, flags, n,
                status.ctypes.data, outlen.ctypes.data, nthreads)
        assert rc == 0
        return dst, status[:n], outlen[:n]

    def table_build(self, lens, tablebits, cap, litlen):
        lens = np.ascontiguousarray(lens, dtype=np.uint8)
        table = np.zeros(cap, dtype=np.uint32)
        used = C.c_uint(0)
        rc = self.lib.oracle_table_build(table.ctypes.data, C.byref(used), lens.ctypes.data, len(lens),
                                         tablebits, cap, int(litlen))
        return rc, used.value, table


_cached = None


def load() -> Oracle:
    global _cached
    if _cached is None:
        src_m = max(os.path.getmtime(os.path.join(ORACLE_DIR, f)) for f in ("infl_oracle.c", "infl_oracle.h", "png_unfilter.c"))
        if not os.path.exists(SO) or os.path.getmtime(SO) < src_m:
            build()
        _cached = Oracle(C.CDLL(SO))
    return _cached
