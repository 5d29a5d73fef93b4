#!/usr/bin/env python3
"""Diagnostic timing of the headline batch WITHOUT a parity gate, for builds that are wrong on purpose (a phase
switched off to see what it costs).  Never a measurement to report: bench.py is.

    DEFL_AMD_LIB=<lib.so> python tools/time_headline.py [--pool 512] [--steps 10] [--workload text|png|fixed|noise]
"""
import argparse
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import workloads as wl  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", type=int, default=512)
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--workload", default="text")
    ap.add_argument("--check", action="store_true", help="also report how many streams came out with status 0 and the right length")
    args = ap.parse_args()
    import torch
    from defl_amd import capi
    eng = capi.load()
    assert eng.lib.infl_gpu_init(0) == 0
    dev = torch.device("cuda", 0)
    th = bench.host_threads()
    if args.workload == "text":
        sizes = [65536] * args.pool
        comps = bench.build_pool(0, sizes, th)
    elif args.workload == "fixed":
        sizes = [65536] * args.pool
        comps = bench.pool_map(lambda i: wl.deflate(wl.text_stream(i), 6, True, zlib.Z_FIXED), range(args.pool), th)
    elif args.workload == "png":
        raws = [wl.png_like(i) for i in range(args.pool)]
        sizes = [len(r) for r in raws]
        comps = bench.pool_map(wl.deflate, raws, th)
    elif args.workload == "noise":
        rng = np.random.default_rng(8)
        raws = [bytes(rng.integers(32, 96, 65536, dtype=np.uint8)) for _ in range(args.pool)]
        sizes = [65536] * args.pool
        comps = bench.pool_map(wl.deflate, raws, th)
    else:
        raise SystemExit("unknown workload")
    b = bench.Batch(torch, dev, comps, sizes, 0, args.streams)
    step = bench.make_step(eng.lib, torch, b, 1, False)
    _, ms = bench.time_steps(torch, step, args.steps, 2, lambda: None, False)
    line = "%s %s: %.3f ms per launch, %.1f GB/s decompressed" % (os.environ.get("DEFL_AMD_LIB", "default"), args.workload, ms,
                                                                  b.out_bytes / ms / 1e6)
    if args.check:
        good = int(((b.d_status[: b.n] == 0) & (b.d_outlen[: b.n] == b.want_len)).sum())
        line += ", %d of %d streams with status 0 and the right length" % (good, b.n)
    print(line, flush=True)


if __name__ == "__main__":
    main()
