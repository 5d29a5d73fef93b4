#!/usr/bin/env python3
"""Differential fuzz of the GPU engine against the CPU oracle, larger than what the test suite runs
(diagnostic: `python tests/fuzz_campaign.py [first seed] [seeds] [seconds]` on a GPU box).  Streams that
stress the parallel decoder's special paths (rounds cut short, serial stretches, failing streams
reported by the parallel pass): text with runs, near fixed-length codes, fixed-Huffman blocks, flush
points; mutated, truncated, with too little output space.  Both entry points are compared, bit for bit.
With DEFL_FUZZ_HEADERS=1 the streams are small and the damage goes into their first 150 bytes: the block headers
(code-length code, repeat codes, table acceptance) get most of it.  DEFL_FUZZ_LANES = lanes per stream in the serial
passes (default 16)."""
import os
import random
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads as wl  # noqa: E402


def make_raw(rng, nrng):
    size = rng.choice([rng.randint(1, 3000), rng.randint(3000, 70000), rng.randint(70000, 300000)])
    kind = rng.randrange(6)
    if kind == 0:
        raw = bytearray(wl.text_stream(rng.randrange(1 << 20), size))
    elif kind == 1:
        raw = bytearray(nrng.integers(32, min(256, 32 + (1 << rng.randint(1, 8))), size, dtype=np.uint8).tobytes())
    elif kind == 2:
        raw = bytearray(b"".join(bytes([rng.randrange(256)]) * rng.randint(1, 6000) for _ in range(size // 2000 + 1))[:size])
    elif kind == 3:
        rec = bytes(rng.randrange(256) for _ in range(rng.randint(2, 600)))
        raw = bytearray((rec * (size // len(rec) + 1))[:size])
    else:
        raw = bytearray(wl.text_stream(rng.randrange(1 << 20), size))
        for _ in range(rng.randint(1, 6)):
            at, n = rng.randrange(0, size), rng.randint(100, 20000)
            raw[at:at + n] = (bytes([rng.randrange(256)]) * n)[: size - at]
    return bytes(raw)


def compress(raw, rng):
    strat = rng.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
    level = rng.choice([1, 6, 6, 9])
    if rng.random() < 0.25:
        c = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strat)
        out, i = [], 0
        while i < len(raw):
            n = rng.randint(1, 40000)
            out.append(c.compress(raw[i:i + n]))
            out.append(c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_BLOCK])))
            i += n
        return b"".join(out) + c.flush()
    return wl.deflate(raw, level, True, strat)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
    from defl_amd import capi
    from tests import oracle_lib
    from tests.test_gpu_parity import _check_batch
    eng, oracle = capi.load(), oracle_lib.load()
    assert eng.lib.infl_gpu_init(0) == 0
    t0, total = time.time(), 0
    for seed in range(first, first + seeds):
        rng, nrng = random.Random(seed), np.random.default_rng(seed)
        headers = os.environ.get("DEFL_FUZZ_HEADERS") == "1"
        if headers:
            raws = [make_raw(rng, nrng)[: rng.randint(200, 6000)] for _ in range(40)]
            bases = [(r, compress(r, rng)) for r in raws]
        else:
            bases = [(r, compress(r, rng)) for r in (make_raw(rng, nrng) for _ in range(10))]
        streams, caps = [], []
        for _ in range(2000 if headers else 200):
            raw, comp = rng.choice(bases)
            m, what = bytearray(comp), rng.random()
            if headers and len(m) > 4:
                for _ in range(rng.randint(1, 3)):
                    m[rng.randrange(2, min(150, len(m)))] ^= 1 << rng.randrange(8)
                if what < 0.15:
                    m = m[: rng.randrange(2, min(200, len(m)) + 1)]
            elif what < 0.3:
                for _ in range(rng.randint(1, 4)):
                    m[rng.randrange(min(2, len(m) - 1), len(m))] ^= 1 << rng.randrange(8)
            elif what < 0.5:
                m = m[: rng.randrange(0, len(m) + 1)]
            elif what < 0.55:
                i = rng.randrange(0, len(m))
                m[i:i + rng.randint(1, 40)] = bytes(rng.randrange(256) for _ in range(rng.randint(1, 40)))
            streams.append(bytes(m))
            caps.append(rng.choice([len(raw), len(raw), len(raw) + 100, max(0, len(raw) - rng.randint(1, 5000)),
                                    rng.randint(0, len(raw))]))
        print("seed %d: %d streams, %d KB in" % (seed, len(streams), sum(map(len, streams)) >> 10), flush=True)
        _check_batch(eng, oracle, streams, caps, 1, int(os.environ.get("DEFL_FUZZ_LANES", "16")), "campaign seed %d" % seed, 2)
        total += len(streams)
        if time.time() - t0 > budget:
            break
    print("campaign ok: %d streams in %.0f s" % (total, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
