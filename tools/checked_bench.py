#!/usr/bin/env python3
"""What the opt-in integrity pass costs: infl_batch_dev against infl_batch_dev_checked (Adler-32 of zlib streams,
CRC-32 + ISIZE of gzip members) on device-resident batches of 64 KiB text streams; wall clock around `reps` calls."""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import workloads as wl                                             # noqa: E402
from defl_amd import capi                                          # noqa: E402


def gzip_member(raw):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = c.compress(raw) + c.flush()
    return b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + body + zlib.crc32(raw).to_bytes(4, "little") + (len(raw) & 0xFFFFFFFF).to_bytes(4, "little")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=32768)
    ap.add_argument("--distinct", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    eng = capi.load()
    lib = eng.lib
    assert lib.infl_gpu_init(0) == 0
    n = a.streams
    raws = [wl.text_stream(9000 + i, 65536) for i in range(a.distinct)]
    out = {"streams": n, "decompressed_bytes": n * 65536}
    for name, check, comps in (("zlib", capi.CHECK_ADLER32, [wl.deflate(r) for r in raws]),
                               ("gzip", capi.CHECK_GZIP, [gzip_member(r) for r in raws])):
        src, so, do = wl.pack([comps[i % a.distinct] for i in range(n)], [65536] * n)
        b = eng.device_batch(src, so, do)
        work = lib.infl_gpu_malloc(8 * n)
        assert work

        def plain():
            assert lib.infl_batch_dev(b.d_src, b.d_soff, b.d_dst, b.d_doff, 1, n, b.d_status, b.d_outlen, None) == 0

        def checked():
            assert lib.infl_batch_dev_checked(b.d_src, b.d_soff, b.d_dst, b.d_doff, check, n, b.d_status, b.d_outlen,
                                              b.d_detail, work, None) == 0

        def timed(fn):
            fn(); assert lib.infl_gpu_sync() == 0
            t = time.perf_counter()
            for _ in range(a.reps):
                fn()
            assert lib.infl_gpu_sync() == 0
            return (time.perf_counter() - t) / a.reps

        if name == "zlib":
            out["plain_ms"] = round(timed(plain) * 1e3, 3)
        t = timed(checked)
        _, status, outlen, detail = b.results(want_dst=False)
        assert (status == 0).all() and (outlen == 65536).all(), (name, int((status != 0).sum()))
        out[name + "_checked_ms"] = round(t * 1e3, 3)
        out[name + "_checked_GBps"] = round(n * 65536 / t / 1e9, 1)
        lib.infl_gpu_free(work)
        b.free()
    out["plain_GBps"] = round(n * 65536 / (out["plain_ms"] * 1e-3) / 1e9, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
