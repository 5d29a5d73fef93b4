#!/usr/bin/env python3
"""Time of one infl_init / infl_include / infl / infl_destroy sequence (through ctypes) for single streams of a few sizes,
on a GPU box: `python tools/oneshot_latency.py`.  Prints decompressed size, compressed size and microseconds per call."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import workloads as wl
from defl_amd import capi
eng = capi.load()
assert eng.lib.infl_gpu_init(0) == 0
for size in (100, 300, 600, 2000, 65536):
    raw = wl.text_stream(77, size); c = wl.deflate(raw)
    for _ in range(20): rc, out = eng.infl_buf(c, size, 1)[:2]
    assert rc == 0
    t = time.perf_counter()
    for _ in range(300): eng.infl_buf(c, size, 1)
    print(size, len(c), "%.1f us per infl_buf" % ((time.perf_counter() - t) / 300 * 1e6))
