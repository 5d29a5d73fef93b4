#!/usr/bin/env python3
"""What failing streams cost when the caller asks for detail codes (infl_batch_dev_detail): a batch of 64 KiB text
streams of which one in `--every` is cut off at a random place; ms per launch, and the results against the
plain call's verdicts."""
import argparse
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import workloads as wl                                             # noqa: E402
from defl_amd import capi                                          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=16384)
    ap.add_argument("--every", type=int, default=64)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    eng = capi.load()
    lib = eng.lib
    assert lib.infl_gpu_init(0) == 0
    rng = random.Random(3)
    distinct = 128
    comps = [wl.deflate(wl.text_stream(4000 + i, 65536)) for i in range(distinct)]
    streams = []
    for i in range(a.streams):
        c = comps[i % distinct]
        if i % a.every == 7:
            c = c[: rng.randrange(len(c) // 4, len(c) - 8)]
        streams.append(c)
    src, so, do = wl.pack(streams, [65536] * a.streams)
    b = eng.device_batch(src, so, do)
    out = {"streams": a.streams, "failing": sum(1 for i in range(a.streams) if i % a.every == 7)}
    for name, detail in (("plain", False), ("detail", True)):
        b.run(1, detail=detail)
        _, status, outlen, det = b.results(want_dst=False)
        t = time.perf_counter()
        for _ in range(a.reps):
            b.run(1, detail=detail)
        assert lib.infl_gpu_sync() == 0
        out[name + "_ms"] = round((time.perf_counter() - t) / a.reps * 1e3, 2)
        bad = status != 0
        assert int(bad.sum()) == out["failing"] and (outlen[bad] == 0).all()
        if detail:
            assert (det[bad] == capi.D_NEED_INPUT).all(), set(det[bad].tolist())
    print(json.dumps(out))
    b.free()


if __name__ == "__main__":
    main()
