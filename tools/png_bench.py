#!/usr/bin/env python3
"""Times infl_png_unfilter_dev behind infl_batch_dev on the PNG-shaped configuration (zlib streams of
128 x 128 RGBA scanlines with shuffled filter types): wall clock around `reps` launches, device-resident data.
Prints the decode alone, the un-filter alone and both, in GB/s of filtered bytes."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import workloads as wl                                             # noqa: E402
from defl_amd import capi                                          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=16384)
    ap.add_argument("--distinct", type=int, default=64)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--width", type=int, default=128)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--bpp", type=int, default=4)
    a = ap.parse_args()
    eng = capi.load()
    lib = eng.lib
    assert lib.infl_gpu_init(0) == 0
    n, rb = a.images, a.width * a.bpp
    base = [wl.png_filtered(7000 + i, a.width, a.height, a.bpp) for i in range(a.distinct)]
    comp = [wl.deflate(f) for _, f, _ in base]
    fbytes = a.height * (rb + 1)
    src, so, do = wl.pack([comp[i % a.distinct] for i in range(n)], [fbytes] * n)
    b = eng.device_batch(src, so, do)
    g = np.zeros((n, 4), dtype=np.uint32)
    g[:, 0], g[:, 1], g[:, 2] = a.height, rb, a.bpp
    ooff = (np.arange(n + 1, dtype=np.uint64) * np.uint64(a.height * rb))
    d_geom, d_ooff, d_out, d_st = (lib.infl_gpu_malloc(x) for x in (g.nbytes, ooff.nbytes, int(ooff[-1]) + 16, 4 * n))
    assert all((d_geom, d_ooff, d_out, d_st))
    assert lib.infl_gpu_upload(d_geom, g.ctypes.data, g.nbytes) == 0
    assert lib.infl_gpu_upload(d_ooff, ooff.ctypes.data, ooff.nbytes) == 0

    def decode():
        assert lib.infl_batch_dev(b.d_src, b.d_soff, b.d_dst, b.d_doff, 1, n, b.d_status, b.d_outlen, None) == 0

    def unfilter():
        assert lib.infl_png_unfilter_dev(b.d_dst, b.d_doff, b.d_outlen, b.d_status, d_out, d_ooff, d_geom, n, d_st, None) == 0

    def timed(fn):
        fn(); assert lib.infl_gpu_sync() == 0
        t = time.perf_counter()
        for _ in range(a.reps):
            fn()
        assert lib.infl_gpu_sync() == 0
        return (time.perf_counter() - t) / a.reps

    t_dec = timed(decode)
    t_unf = timed(unfilter)
    t_both = timed(lambda: (decode(), unfilter()))
    st = np.zeros(n, dtype=np.int32)
    assert lib.infl_gpu_download(st.ctypes.data, d_st, 4 * n) == 0 and (st == 0).all()
    out = np.zeros(a.height * rb, dtype=np.uint8)
    for i in (0, n // 2 + 1, n - 1):
        assert lib.infl_gpu_download(out.ctypes.data, d_out + i * a.height * rb, out.nbytes) == 0
        assert out.tobytes() == base[i % a.distinct][0], i
    total = n * fbytes
    print(json.dumps({"images": n, "filtered_bytes": total, "compressed_bytes": int(so[-1]),
                      "decode_ms": round(t_dec * 1e3, 3), "unfilter_ms": round(t_unf * 1e3, 3), "both_ms": round(t_both * 1e3, 3),
                      "decode_GBps": round(total / t_dec / 1e9, 1), "unfilter_GBps": round(total / t_unf / 1e9, 1),
                      "both_GBps": round(total / t_both / 1e9, 1),
                      "unfilter_hbm_frac": round((total + n * a.height * rb) / t_unf / 8e12, 4)}))
    for p in (d_geom, d_ooff, d_out, d_st):
        lib.infl_gpu_free(p)
    b.free()


if __name__ == "__main__":
    main()
