#!/usr/bin/env python3
"""Which streams does the wavefront-parallel pass hand over to the serial passes?  Diagnostic build only
(make -C defl_amd/csrc prof; DEFL_AMD_LIB=defl_amd/libdefl_amd_prof.so): runs the fixture files of
tests/golden/defl_testdata, or synthetic streams, through the parallel pass alone and lists the streams
that keep the hand-over mark, with their sizes."""
import argparse
import ctypes as C
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads as wl  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="fixtures", choices=["fixtures", "embedded-runs", "files"])
    args = ap.parse_args()
    from defl_amd import capi
    eng = capi.load()
    assert eng.lib.infl_gpu_init(0) == 0
    assert eng.lib.infl_gpu_set_table_variant(3) == 3, "needs the diagnostic build"
    fn = eng.lib.infl_debug_par_profile
    fn.argtypes = [C.POINTER(C.c_ulonglong)]
    buf = (C.c_ulonglong * 32)()
    flags = 0
    if args.what == "fixtures":
        d = os.path.join(ROOT, "tests", "golden", "defl_testdata")
        names = sorted(n for n in os.listdir(os.path.join(d, "compressed"))
                       if os.path.exists(os.path.join(d, "raw", n)))
        comps = [open(os.path.join(d, "compressed", n), "rb").read() for n in names]
        caps = [os.path.getsize(os.path.join(d, "raw", n)) + 1000 for n in names]
    elif args.what == "files":
        # 64 KiB slices of whatever binaries, libraries and scripts this machine holds, zlib levels 1, 6, 9
        import glob
        paths = [p for pat in ("/usr/bin/python3*", "/usr/lib/x86_64-linux-gnu/libc.so*", "/usr/lib/x86_64-linux-gnu/libstdc++.so*",
                               "/usr/lib/python3/dist-packages/*.py", "/usr/lib/python3.10/*.py", "/usr/share/dict/*",
                               "/opt/rocm/lib/librocm_smi64.so*", "/usr/bin/gcc*", "/etc/*.conf")
                 for p in sorted(glob.glob(pat)) if os.path.isfile(p)]
        names, comps, caps = [], [], []
        for pth in paths:
            data = open(pth, "rb").read(1 << 22)
            for off in range(0, min(len(data), 16 * 65536), 65536):
                piece = data[off:off + 65536]
                if len(piece) < 2000:
                    continue
                for lvl in (1, 6, 9):
                    names.append("%s+%d L%d" % (os.path.basename(pth), off, lvl))
                    comps.append(wl.deflate(piece, lvl))
                    caps.append(len(piece))
        flags = 1
    else:
        # text with runs and repeats spliced in: what real files look like (flat image areas, padding)
        rng = np.random.default_rng(11)
        names, comps, caps = [], [], []
        for i in range(256):
            t = bytearray(wl.text_stream(i))
            for _ in range(int(rng.integers(0, 4))):
                at, n = int(rng.integers(0, 60000)), int(rng.integers(300, 5000))
                t[at:at + n] = bytes([int(rng.integers(0, 256))]) * n
            names.append("text+runs %d" % i)
            comps.append(wl.deflate(bytes(t)))
            caps.append(len(t))
        flags = 1
    src, src_off, dst_off = wl.pack(comps, caps)
    b = eng.device_batch(src, src_off, dst_off)
    fn(buf)
    b.run(flags, detail=False)
    assert fn(buf) == 0
    _, status, outlen, _ = b.results(want_dst=False)
    bad = [i for i in range(len(names)) if status[i] != 0]
    print("%d of %d streams handed over (blocks: tables too large for the compact format %d, malformed %d; "
          "serial stretches inside the parallel pass %d)" % (len(bad), len(names), buf[15], buf[29], buf[28]))
    for i in bad[:60]:
        print("  %-40s compressed %7d  raw capacity %7d" % (names[i], len(comps[i]), caps[i]))


if __name__ == "__main__":
    main()
