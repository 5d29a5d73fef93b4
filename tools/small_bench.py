#!/usr/bin/env python3
"""Device-resident batches of small text streams (default 1 KiB and 4 KiB, 65 536 streams each): ms per launch and
GB/s decompressed through infl_batch_dev.  `--size N` runs one size only (for counter runs under rocprofv3)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import workloads as wl                                             # noqa: E402
from defl_amd import capi                                          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--distinct", type=int, default=512)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--sizes", type=str, default="", help="comma separated stream sizes")
    ap.add_argument("--variant", type=int, default=-1, help="1 = the serial passes alone, 2 = parallel pass first")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per stream in the serial passes (8, 16, 32, 64)")
    a = ap.parse_args()
    eng = capi.load()
    lib = eng.lib
    assert lib.infl_gpu_init(0) == 0
    if a.variant >= 0:
        assert lib.infl_gpu_set_table_variant(a.variant) == a.variant
    if a.lanes:
        assert lib.infl_gpu_set_lanes_per_stream(a.lanes) == a.lanes
    out = {}
    sizes = [int(x) for x in a.sizes.split(",")] if a.sizes else [a.size] if a.size else [1024, 4096, 16384]
    for size in sizes:
        raws = [wl.text_stream(3000 + i, size) for i in range(a.distinct)]
        comps = [wl.deflate(r) for r in raws]
        n = a.streams
        src, so, do = wl.pack([comps[i % a.distinct] for i in range(n)], [size] * n)
        b = eng.device_batch(src, so, do)
        b.run(1, detail=False)
        dst, status, outlen, _ = b.results()
        assert (status == 0).all() and (outlen == size).all()
        for i in (0, 1, n // 2, n - 1):
            assert dst[i * size:(i + 1) * size].tobytes() == raws[i % a.distinct]
        assert lib.infl_gpu_sync() == 0
        t = time.perf_counter()
        for _ in range(a.reps):
            b.run(1, detail=False)
        assert lib.infl_gpu_sync() == 0
        dt = (time.perf_counter() - t) / a.reps
        out[str(size)] = {"ms": round(dt * 1e3, 3), "GBps": round(n * size / dt / 1e9, 1), "us_per_stream_slot": round(dt * 1e6 * 4096 / n, 1)}
        b.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
