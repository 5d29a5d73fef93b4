#!/usr/bin/env python3
"""Cycle accounting of the wavefront-parallel block decoder (diagnostic build: make -C defl_amd/csrc prof).

    DEFL_AMD_LIB=defl_amd/libdefl_amd_prof.so python tools/par_profile.py [--streams N]

Prints the share of wave cycles each phase of huffman_block_parallel takes on the headline
workload, plus rounds, fix-up rescans, pointer-jumping sweeps and lanes used per round.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
import workloads as wl  # noqa: E402

PHASES = ["stage-in", "first scan", "fix-up", "place", "emit", "expand", "jump", "fetch+write", "?", "drain stores"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=8192)
    ap.add_argument("--pool", type=int, default=256)
    ap.add_argument("--workload", default="text", choices=["text", "noise", "fixed", "png", "mixed", "files"])
    ap.add_argument("--size", type=int, default=0, help="text: decompressed bytes per stream (default 64 KiB)")
    args = ap.parse_args()
    from defl_amd import capi
    eng = capi.load()
    assert eng.lib.infl_gpu_init(0) == 0
    fn = eng.lib.infl_debug_par_profile
    fn.argtypes = [C.POINTER(C.c_ulonglong)]
    eng.lib.infl_gpu_set_table_variant(2)
    if args.workload == "text":
        comps = bench.build_pool(0, [args.size or bench.STREAM_BYTES] * args.pool, 4)
    elif args.workload == "fixed":
        import zlib
        comps = [wl.deflate(wl.text_stream(i), 6, True, zlib.Z_FIXED) for i in range(args.pool)]
    elif args.workload == "png":
        caps = [len(wl.png_like(i)) for i in range(args.pool)]
        comps = [wl.deflate(wl.png_like(i)) for i in range(args.pool)]
    elif args.workload == "files":
        import glob
        paths = [p for pat in ("/usr/bin/python3*", "/usr/lib/x86_64-linux-gnu/libc.so*",
                               "/usr/lib/x86_64-linux-gnu/libstdc++.so*", "/usr/lib/python3.10/*.py", "/usr/bin/gcc*")
                 for p in sorted(glob.glob(pat)) if os.path.isfile(p)]
        raws = []
        for pth in paths:
            data = open(pth, "rb").read(1 << 21)
            raws += [data[o:o + 65536] for o in range(0, len(data) - 65535, 65536)]
        raws = raws[: args.pool]
        args.pool = len(raws)
        comps = [wl.deflate(x) for x in raws]
        caps = [65536] * len(raws)
    elif args.workload == "mixed":
        rng = np.random.default_rng(5)
        caps = [int(x) for x in np.exp(rng.uniform(np.log(1024), np.log(262144), args.pool))]
        comps = [wl.deflate(wl.text_stream(1000 + i, sz)) for i, sz in enumerate(caps)]
    else:
        rng = np.random.default_rng(8)
        comps = [wl.deflate(bytes(rng.integers(32, 96, 65536, dtype=np.uint8))) for i in range(args.pool)]
    if args.workload not in ("png", "mixed", "files"):
        caps = [(args.size if args.workload == "text" and args.size else bench.STREAM_BYTES)] * len(comps)
    reps = (args.streams + args.pool - 1) // args.pool
    comps = (comps * reps)[: args.streams]
    caps = (caps * reps)[: args.streams]
    src, src_off, dst_off = wl.pack(comps, caps)
    b = eng.device_batch(src, src_off, dst_off)
    buf = (C.c_ulonglong * 32)()
    b.run(1, detail=False)
    assert fn(buf) == 0                      # warm-up, cleared
    b.run(1, detail=False)
    assert fn(buf) == 0
    v = np.array(list(buf), dtype=np.float64)
    _, status, outlen, _ = b.results(want_dst=False)
    assert (status == 0).all()
    # how many streams the parallel pass hands over to the serial passes (diagnostic variant 3:
    # the parallel pass alone, handed-over streams keep their mark)
    if eng.lib.infl_gpu_set_table_variant(3) == 3:
        b.run(1, detail=False)
        assert fn(buf) == 0
        _, st3, _, _ = b.results(want_dst=False)
        print("handed over to the serial passes: %d of %d streams" % (int((st3 != 0).sum()), len(st3)))
        print("  blocks handed over because: tables too large for the compact format %d, malformed %d; "
              "serial stretches inside the parallel pass %d" % (buf[15], buf[29], buf[28]))
        eng.lib.infl_gpu_set_table_variant(2)
    # (streams handed to the serial passes do not flush their counters: the shares describe the others)
    tot = v[:10].sum()
    n = len(comps)
    print("streams %d; wave cycles per stream %.0f" % (n, tot / n))
    for k, name in enumerate(PHASES):
        print("  %-11s %6.2f %%  %9.0f cycles/stream" % (name, 100 * v[k] / tot, v[k] / n))
    print("  %-11s %6.2f %%  %9.0f cycles/stream (dynamic block headers and table builds, outside the phases above)"
          % ("block setup", 100 * v[14] / tot, v[14] / n))
    rounds = v[10]
    print("rounds/stream %.2f; rescans/round %.2f; sweeps/round %.2f; lanes/round %.1f"
          % (rounds / n, v[11] / rounds, v[12] / rounds, v[13] / rounds))
    if v[30]:
        print("steps %d per stream, %.1f %% of them with a second-level lookup by some lane"
              % (v[30] / n, 100 * v[31] / v[30]))
    if os.environ.get("DEFL_PROFILE_SETUP"):      # a build with -DDEFL_PAR_PROFILE_SETUP: the slots hold the set-up split
        names = ["header + code-length lengths", "code-length table", "code-length loop", "literal/length table", "distance table"]
        print("block setup split (cycles per stream): " + "; ".join("%s %.0f" % (nm, v[16 + k] / n) for k, nm in enumerate(names)))
        return
    print("full scans per round (share of rounds): " +
          "  ".join("%d:%.2f%%" % (k, 100 * v[16 + k] / rounds) for k in range(12)))


if __name__ == "__main__":
    main()
