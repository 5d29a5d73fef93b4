"""Parity of the wavefront-parallel block decoder (table variant 2) on the shapes that stress its
own machinery: block ends at every position of a round, several blocks per stream, chains that hit
the per-lane byte cap, codes that never re-synchronise, output that fills up inside a round,
input that ends inside a round.  Everything is compared with the CPU oracle, bit for bit."""
import random
import zlib

import numpy as np
import pytest

import workloads as wl
from tests.test_gpu_parity import _check_batch, _corpus

pytestmark = pytest.mark.gpu


def _deflate_flushed(raw, rng, level=6):
    """zlib stream with flush points: empty stored blocks and fresh dynamic blocks in between."""
    c = zlib.compressobj(level)
    out, i = [], 0
    while i < len(raw):
        n = rng.randint(1, 9000)
        out.append(c.compress(raw[i:i + n]))
        out.append(c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_BLOCK])))
        i += n
    out.append(c.flush())
    return b"".join(out)


def test_every_block_length(engine, oracle):
    rng = random.Random(11)
    text = _corpus(rng, 60000)
    raws = [text[:n] for n in list(range(0, 700, 7)) + list(range(700, 60000, 1499))]
    streams = [wl.deflate(r) for r in raws]
    _check_batch(engine, oracle, streams, [len(r) for r in raws], 1, 16, "lengths", 2)


def test_many_blocks_per_stream(engine, oracle):
    rng = random.Random(12)
    raws = [_corpus(rng, rng.randint(1, 150000)) for _ in range(40)]
    streams = [_deflate_flushed(r, rng, rng.choice([1, 6, 9])) for r in raws]
    _check_batch(engine, oracle, streams, [len(r) + 5 for r in raws], 1, 16, "flushed", 2)


def test_streams_of_many_short_blocks(engine, oracle):
    """Flush points every few bytes (interactive protocols): hundreds of tiny blocks per stream,
    decoded symbol by symbol after the first one; long blocks in between go back to rounds."""
    rng = random.Random(19)
    streams, caps = [], []
    for k in range(24):
        raw = _corpus(rng, rng.randint(200, 30000))
        c = zlib.compressobj(rng.choice([1, 6, 9]), zlib.DEFLATED, 15, 9,
                             rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED]))
        out, i = [], 0
        while i < len(raw):
            n = rng.randint(1, 120) if rng.random() < 0.9 else rng.randint(2000, 9000)
            out.append(c.compress(raw[i:i + n]))
            out.append(c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_BLOCK, zlib.Z_PARTIAL_FLUSH])))
            i += n
        out.append(c.flush())
        comp = b"".join(out)
        streams += [comp, comp[: len(comp) * 2 // 3], comp]
        caps += [len(raw), len(raw), len(raw) // 2]
    _check_batch(engine, oracle, streams, caps, 1, 16, "short blocks", 2)


def test_long_matches_hit_the_byte_cap(engine, oracle):
    rng = random.Random(13)
    raws = []
    for _ in range(24):
        parts = []
        while sum(map(len, parts)) < 90000:
            kind = rng.random()
            if kind < 0.5:
                parts.append(bytes([rng.randrange(256)]) * rng.randint(1, 5000))
            elif kind < 0.8:
                parts.append(_corpus(rng, rng.randint(1, 3000)))
            else:
                parts.append(bytes(rng.randrange(256) for _ in range(rng.randint(1, 40))) * rng.randint(1, 300))
        raws.append(b"".join(parts))
    streams = [wl.deflate(r, rng.choice([1, 6, 9])) for r in raws]
    _check_batch(engine, oracle, streams, [len(r) for r in raws], 1, 16, "runs", 2)


def _text_with_runs(rng, size, runs):
    t = bytearray(_corpus(rng, size))
    for _ in range(runs):
        at, n = rng.randrange(0, max(1, size - 10)), rng.randint(300, 9000)
        t[at:at + n] = bytes([rng.randrange(256)]) * min(n, size - at)
    return bytes(t)


def test_serial_stretches_inside_text(engine, oracle):
    """Very long matches inside ordinary text: the parallel decoder has those stretches decoded
    symbol by symbol and goes on behind them.  Stretches at the start, in the middle, at the
    end of a block, several in a row (a run far longer than one stretch), output capacity and
    input ending inside a stretch."""
    rng = random.Random(18)
    raws = [_text_with_runs(rng, 70000, k) for k in (1, 2, 3, 5, 8)]
    raws += [bytes(5000) + _corpus(rng, 30000), _corpus(rng, 30000) + bytes(40000),
             _corpus(rng, 4000) + bytes(300000) + _corpus(rng, 4000), bytes(200000)]
    streams = [wl.deflate(r, lvl) for r in raws for lvl in (6, 9)]
    caps = [len(r) for r in raws for _ in (6, 9)]
    _check_batch(engine, oracle, streams, caps, 1, 16, "stretches", 2)
    comp = wl.deflate(raws[7])
    out_caps = [0, 1, 3999, 4000, 4001, 4258, 5000, 70000, 304000 - 1, 307999, 308000]
    _check_batch(engine, oracle, [comp] * len(out_caps), out_caps, 1, 16, "stretch caps", 2)
    cuts = sorted(set(range(1, len(comp), max(1, len(comp) // 97))))
    _check_batch(engine, oracle, [comp[:k] for k in cuts], [len(raws[7])] * len(cuts), 1, 16, "stretch cuts", 2)


def test_codes_that_do_not_resynchronise(engine, oracle):
    rng = np.random.default_rng(14)
    raws = [bytes(rng.integers(32, 32 + (1 << k), 30000 + 1000 * k, dtype=np.uint8)) for k in (1, 2, 3, 4, 5, 6, 7)]
    raws += [bytes(rng.integers(0, 256, 20000, dtype=np.uint8))]
    streams = [wl.deflate(r, 6, True, zlib.Z_HUFFMAN_ONLY) for r in raws] + [wl.deflate(r) for r in raws]
    streams += [wl.deflate(r, 6, True, zlib.Z_FIXED) for r in raws]
    _check_batch(engine, oracle, streams, [len(r) for r in raws] * 3, 1, 16, "no-resync", 2)


def test_output_fills_up_inside_a_round(engine, oracle):
    rng = random.Random(15)
    text = _corpus(rng, 50000)
    comp = wl.deflate(text)
    caps = [0, 1, 63, 64, 65, 1000, 2047, 2048, 2049, 6143, 6144, 6145, 20000, 49999, 50000, 50001]
    _check_batch(engine, oracle, [comp] * len(caps), caps, 1, 16, "caps", 2)


def test_input_ends_inside_a_round(engine, oracle):
    rng = random.Random(16)
    text = _corpus(rng, 50000)
    comp = wl.deflate(text)
    cuts = sorted(set(list(range(0, 64)) + list(range(64, len(comp), 397)) + [len(comp) - k for k in range(1, 12)]))
    streams = [comp[:k] for k in cuts]
    _check_batch(engine, oracle, streams, [50000] * len(cuts), 1, 16, "cuts", 2)


def test_bit_flips_in_long_streams(engine, oracle):
    rng = random.Random(17)
    text = _corpus(rng, 80000)
    comp = bytearray(wl.deflate(text))
    streams = []
    for _ in range(96):
        m = bytearray(comp)
        for _ in range(rng.randint(1, 3)):
            i = rng.randrange(2, len(m))
            m[i] ^= 1 << rng.randrange(8)
        streams.append(bytes(m))
    _check_batch(engine, oracle, streams, [80000] * len(streams), 1, 16, "flips", 2)


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_differential_fuzz_many_rounds(engine, oracle, seed):
    """Mutated and truncated multi-round streams (several blocks, levels, strategies), random
    capacities: status, length and bytes must match the oracle for every one of them."""
    rng = random.Random(seed)
    nrng = np.random.default_rng(seed)
    bases = []
    for k in range(12):
        size = rng.randint(2000, 120000)
        kind = k % 4
        if kind == 0:
            raw = _corpus(rng, size)
        elif kind == 1:
            raw = bytes(nrng.integers(97, 97 + rng.choice([2, 4, 16, 26]), size, dtype=np.uint8))
        elif kind == 2:
            raw = b"".join(bytes([rng.randrange(256)]) * rng.randint(1, 700) for _ in range(size // 300 + 1))[:size]
        else:
            raw = _corpus(rng, size // 2) + bytes(nrng.integers(0, 256, size - size // 2, dtype=np.uint8))
        comp = _deflate_flushed(raw, rng, rng.choice([1, 6, 9])) if k % 3 == 0 else \
            wl.deflate(raw, rng.choice([1, 6, 9]), True, rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE]))
        bases.append((raw, comp))
    streams, caps = [], []
    for _ in range(600):
        raw, comp = rng.choice(bases)
        m = bytearray(comp)
        what = rng.random()
        if what < 0.45:
            for _ in range(rng.randint(1, 4)):
                m[rng.randrange(2, len(m))] ^= 1 << rng.randrange(8)
        elif what < 0.7:
            m = m[:rng.randrange(0, len(m))]
        elif what < 0.8:
            i = rng.randrange(2, len(m))
            m[i:i + rng.randint(1, 40)] = bytes(rng.randrange(256) for _ in range(rng.randint(1, 40)))
        streams.append(bytes(m))
        caps.append(rng.choice([len(raw), len(raw), len(raw) + 100, max(0, len(raw) - rng.randint(1, 5000))]))
    _check_batch(engine, oracle, streams, caps, 1, 16, "fuzz-%d" % seed, 2)


@pytest.mark.parametrize("n", [1, 255, 257, 5000, 70001])
def test_sorted_entry_point_gives_the_same_results(engine, oracle, n):
    """infl_batch_dev_sorted (the batch put into size classes by a counting sort on the device before the decode) against
    the oracle and against infl_batch_dev: streams of 0 bytes to 100 KiB in no order, damaged and truncated ones, stored
    blocks and streams whose output does not fit among them; batch sizes around the sort kernels' block size."""
    lib = engine.lib
    rng = random.Random(4000 + n)
    distinct = min(n, 600)
    raws, comps, caps = [], [], []
    for k in range(distinct):
        size = int(2 ** rng.uniform(0, 16.6)) if k % 9 else 0
        raw = wl.text_stream(7000 + k, size) if k % 5 else wl.stored_stream(k, size)[0]
        comp = wl.deflate(raw) if k % 5 else wl.stored_stream(k, size)[1]
        if k % 37 == 11 and len(comp) > 20:
            comp = comp[: len(comp) // 2]
        if k % 41 == 13 and len(comp) > 30:
            bad = bytearray(comp); bad[len(bad) // 2] ^= 0x24; comp = bytes(bad)
        raws.append(raw); comps.append(comp); caps.append(max(0, len(raw) - (3 if k % 53 == 17 else 0)))
    pick = [rng.randrange(distinct) for _ in range(n)]
    streams, cap = [comps[j] for j in pick], [caps[j] for j in pick]
    src, src_off, dst_off = wl.pack(streams, cap)
    want_dst, want_status, want_outlen = oracle.infl_batch(src, src_off, dst_off, 1)
    b = engine.device_batch(src, src_off, dst_off)
    work = lib.infl_gpu_malloc(1024 + 4 * n + 16)
    try:
        assert work
        b.run(1, detail=False)
        dst0, status0, outlen0, _ = b.results()
        fill = np.full(max(1, b.dst_bytes), 0xA5, dtype=np.uint8)
        assert lib.infl_gpu_upload(b.d_dst, fill.ctypes.data, b.dst_bytes) == 0
        assert lib.infl_batch_dev_sorted(b.d_src, b.d_soff, b.d_dst, b.d_doff, 1, n, b.d_status, b.d_outlen, work, None) == 0
        dst, status, outlen, _ = b.results()
    finally:
        lib.infl_gpu_free(work)
        b.free()
    assert np.array_equal(status, want_status) and np.array_equal(outlen, want_outlen)
    assert np.array_equal(status0, want_status) and np.array_equal(outlen0, want_outlen)
    for i in range(n):
        a, m = int(dst_off[i]), int(outlen[i])
        assert np.array_equal(dst[a:a + m], want_dst[a:a + m]), i
    assert (status != 0).sum() > 0 or n == 1
