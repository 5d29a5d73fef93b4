"""Runs the HIP kernel SOURCE on the CPU (tests/emu: one host thread per lane, barriers where
the wavefront runs in lockstep, AddressSanitizer on) and compares it with the oracle.  This
checks the kernel's own indexing, bounds and lane hand-offs without a GPU; the GPU parity
tests proper are in test_gpu_parity.py."""
import os
import random
import zlib

import numpy as np
import pytest

import workloads as wl
from tests import emu_lib, oracle_lib
from tests.deflate_builder import header_cases

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "defl_testdata")
ALL = sorted(os.listdir(os.path.join(DATA, "compressed")))


def _read(kind, name):
    with open(os.path.join(DATA, kind, name), "rb") as f:
        return f.read()


def _compare(oracle, streams, caps, flags, lanes, variant=0, plain=False):
    """plain = without detail codes and input positions asked for (the parallel pass then reports failures itself)"""
    src, src_off, dst_off = wl.pack(streams, caps)
    want_dst, want_status, want_outlen = oracle.infl_batch(src, src_off, dst_off, flags)
    dst, status, outlen, detail, inpos = emu_lib.run(src, src_off, dst_off, flags, lanes, variant, plain=plain)
    for i in range(len(streams)):
        assert status[i] == want_status[i] and outlen[i] == want_outlen[i], \
            (i, int(status[i]), int(want_status[i]), int(outlen[i]), int(want_outlen[i]), int(detail[i]))
        a, n = int(dst_off[i]), int(outlen[i])
        assert np.array_equal(dst[a:a + n], want_dst[a:a + n]), i
        if plain:
            continue
        # detail codes agree with the oracle's own view of why a stream failed
        rc, pos, out, det = oracle.infl(streams[i], caps[i], flags)
        assert det == detail[i], (i, det, int(detail[i]))
        assert inpos[i] <= len(streams[i])
    return status, outlen, detail


@pytest.mark.parametrize("lanes,variant", [(64, 0), (8, 0), (16, 1), (32, 1), (64, 2), (16, 2), (16, 5), (8, 5)])
def test_emulated_kernel_on_fixtures(oracle, lanes, variant):
    names = list(ALL)
    streams = [_read("compressed", n) for n in names]
    caps = [(len(_read("raw", n)) + 1000) if os.path.exists(os.path.join(DATA, "raw", n)) else 70000
            for n in names]
    status, outlen, _ = _compare(oracle, streams, caps, 0, lanes, variant)
    for n, s in zip(names, status):
        if os.path.exists(os.path.join(DATA, "raw", n)):
            assert s == 0, n


@pytest.mark.parametrize("lanes,variant", [(64, 0), (16, 0), (8, 1), (64, 2), (16, 5)])
def test_emulated_kernel_on_header_cases(oracle, lanes, variant):
    streams = header_cases()
    _compare(oracle, streams, [4096] * len(streams), 0, lanes, variant)


def test_emulated_kernel_zlib_and_fuzz(oracle):
    rng = random.Random(77)
    raws = [wl.text_stream(i, 3000 + 500 * i) for i in range(6)] + [bytes(2000), rng.randbytes(1500)]
    good = [wl.deflate(r, lvl, True, strat) for r in raws
            for lvl, strat in ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY))]
    caps = [len(r) for r in raws for _ in range(3)]
    _compare(oracle, good, caps, 1, 32)
    _compare(oracle, good, caps, 1, 16, 1)
    _compare(oracle, good, caps, 1, 16, 5)
    _compare(oracle, good, caps, 1, 8, 5)
    _compare(oracle, good, caps, 1, 64, 2)
    _compare(oracle, good, [max(0, c - 1) for c in caps], 1, 64, 2)
    _compare(oracle, good, [max(0, c - 1) for c in caps], 1, 64)
    bad = []
    for t in range(60):
        b = bytearray(rng.choice(good))
        if t % 3 == 0:
            b = b[:rng.randint(0, len(b))]
        elif t % 3 == 1:
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        else:
            b = bytearray(rng.randbytes(rng.randint(1, 64)))
        bad.append(bytes(b))
    _compare(oracle, bad, [rng.choice([0, 100, 8000]) for _ in bad], 1, 32)
    _compare(oracle, bad, [8000] * len(bad), 0, 8)
    _compare(oracle, bad, [rng.choice([0, 100, 8000]) for _ in bad], 1, 8, 1)
    _compare(oracle, bad, [rng.choice([0, 100, 8000]) for _ in bad], 1, 16, 5)
    _compare(oracle, bad, [rng.choice([0, 100, 8000]) for _ in bad], 1, 64, 2)


def test_emulated_parallel_decoder_corner_shapes(oracle):
    """Wavefront-parallel block decoder (variant 2, 64 lanes): several blocks per stream, chains
    that hit the per-lane byte cap, codes that never re-synchronise, output and input ending
    inside a round."""
    rng = random.Random(78)
    nrng = np.random.default_rng(79)
    text = wl.text_stream(5, 24000)
    c = zlib.compressobj(6)
    flushed = b"".join(c.compress(text[i:i + 3500]) + c.flush(zlib.Z_FULL_FLUSH if i % 7000 else zlib.Z_SYNC_FLUSH)
                       for i in range(0, len(text), 3500)) + c.flush()
    runs = b"".join(bytes([rng.randrange(256)]) * rng.randint(1, 3000) for _ in range(12)) + text[:2000]
    noise = bytes(nrng.integers(32, 96, 9000, dtype=np.uint8))
    streams = [flushed, wl.deflate(runs), wl.deflate(noise), wl.deflate(noise, 6, True, zlib.Z_HUFFMAN_ONLY),
               wl.deflate(text, 6, True, zlib.Z_FIXED)]
    caps = [len(text), len(runs), len(noise), len(noise), len(text)]
    _compare(oracle, streams, caps, 1, 64, 2)
    comp = wl.deflate(text)
    out_caps = [0, 1, 63, 64, 65, 2047, 2048, 2049, 6145, len(text) - 1]
    _compare(oracle, [comp] * len(out_caps), out_caps, 1, 64, 2)
    cuts = [1, 2, 7, 8, 9, 40, 300, 301, len(comp) // 2, len(comp) - 5, len(comp) - 1]
    _compare(oracle, [comp[:k] for k in cuts], [len(text)] * len(cuts), 1, 64, 2)


def test_emulated_parallel_decoder_serial_stretches(oracle):
    """Very long matches inside text (variant 2): stretches decoded symbol by symbol inside the
    parallel decoder, the rounds going on behind them; capacity and input ending inside one."""
    rng = random.Random(80)
    text = bytearray(wl.text_stream(9, 20000))
    for at, n in ((0, 2500), (7000, 6000), (16000, 4000)):
        text[at:at + n] = bytes([rng.randrange(256)]) * n
    text = bytes(text)
    long_run = wl.text_stream(10, 1500) + bytes(120000) + wl.text_stream(11, 1500)
    streams = [wl.deflate(text), wl.deflate(text, 9), wl.deflate(long_run), wl.deflate(bytes(30000))]
    caps = [len(text), len(text), len(long_run), 30000]
    _compare(oracle, streams, caps, 1, 64, 2)
    comp = wl.deflate(long_run)
    out_caps = [0, 1499, 1500, 1501, 1758, 60000, len(long_run) - 1]
    _compare(oracle, [comp] * len(out_caps), out_caps, 1, 64, 2)
    cuts = list(range(1, len(comp), max(1, len(comp) // 12)))
    _compare(oracle, [comp[:k] for k in cuts], [len(long_run)] * len(cuts), 1, 64, 2)


def test_emulated_parallel_pass_reports_certain_failures_itself(oracle):
    """Without detail codes and input positions asked for, the parallel pass writes the result of
    a stream that certainly fails (malformed, truncated, out of space) itself: same status and
    output position as the oracle, nothing left marked for the serial passes."""
    rng = random.Random(81)
    text = wl.text_stream(12, 9000)
    comp = wl.deflate(text)
    runs = wl.deflate(wl.text_stream(13, 1000) + bytes(12000) + wl.text_stream(14, 1000))
    streams, caps = [], []
    for k in (5, 40, 700, len(comp) // 2, len(comp) - 3):
        streams.append(comp[:k]); caps.append(len(text))
    for c in (0, 1, 100, 5000, len(text) - 1):
        streams.append(comp); caps.append(c)
    for _ in range(12):
        b = bytearray(comp)
        b[rng.randrange(2, len(b))] ^= 1 << rng.randrange(8)
        streams.append(bytes(b)); caps.append(len(text))
    for k in (30, len(runs) // 2, len(runs) - 2):
        streams.append(runs[:k]); caps.append(14000)
    for c in (999, 1000, 1300, 9000, 13999):
        streams.append(runs); caps.append(c)
    streams += [comp, runs]; caps += [len(text), 14000]
    src, src_off, dst_off = wl.pack(streams, caps)
    want_dst, want_status, want_outlen = oracle.infl_batch(src, src_off, dst_off, 1)
    dst, status, outlen, _, _ = emu_lib.run(src, src_off, dst_off, 1, 64, 2, plain=True)
    assert (status == want_status).all() and (outlen == want_outlen).all()
    assert (want_status != 0).sum() > 12 and (want_status == 0).sum() >= 2
    for i in range(len(streams)):
        a, n = int(dst_off[i]), int(outlen[i])
        assert np.array_equal(dst[a:a + n], want_dst[a:a + n]), i
    # variant 3 = the parallel pass alone: only what it cannot judge keeps the hand-over mark
    _, st3, _, _, _ = emu_lib.run(src, src_off, dst_off, 1, 64, 3, plain=True)
    assert ((st3 == want_status) | (st3 == 0x40000000)).all()
    short = sum(1 for c in streams if 0 < len(c) < 256)
    assert (st3 == 0x40000000).sum() <= short, st3     # (short inputs are left to the serial passes)


def test_emulated_parallel_decoder_many_short_blocks(oracle):
    """Flush points every few bytes: after a short block the next one is decoded symbol by
    symbol inside the parallel decoder; a long block in between goes back to rounds."""
    rng = random.Random(82)
    text = wl.text_stream(15, 9000)
    streams, caps = [], []
    for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED):
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, strat)
        out, i = [], 0
        while i < len(text):
            n = rng.randint(1, 90) if i < 2500 or i > 7000 else 4000
            out.append(c.compress(text[i:i + n]))
            out.append(c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_BLOCK])))
            i += n
        comp = b"".join(out) + c.flush()
        streams += [comp, comp[: len(comp) // 2], comp]
        caps += [len(text), len(text), 3000]
    _compare(oracle, streams, caps, 1, 64, 2)


def test_emulated_incremental_kernel(oracle):
    """The one-stream resume kernel behind infl_stream(): streams fed in pieces of 1 and 61 bytes
    end where the one-shot decode ends (same bytes, same verdict), whatever the piece size; stops
    inside dynamic headers, inside matches, at stored blocks and at the output's end included."""
    rng = random.Random(83)
    text = wl.text_stream(21, 360)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    multi = c.compress(text[:150]) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(text[150:]) + c.flush()
    streams = [wl.deflate(text, 6, False), wl.deflate(text, 6, False, zlib.Z_FIXED), multi,
               wl.deflate(bytes(rng.randrange(256) for _ in range(150)), 0, False),
               bytes.fromhex("000300fcff6162634b494d0300"),            # stored "abc", then fixed "def"
               wl.deflate(b"a" * 600 + text[:100], 6, False), b"", wl.deflate(text, 6, False)[:-3]]
    caps = [len(text), len(text), len(text), 150, 10, 700, 10, len(text)]
    src, src_off, dst_off = wl.pack(streams, caps)
    want_dst, want_status, want_outlen = oracle.infl_batch(src, src_off, dst_off, 0)
    for chunk in (1, 61):
        dst, status, outlen = emu_lib.run_resume(src, src_off, dst_off, 0, chunk)
        for i in range(len(streams)):
            if want_status[i] == 0 and len(streams[i]):
                assert status[i] == 0 and outlen[i] == want_outlen[i], (chunk, i, int(status[i]), int(outlen[i]))
                a = int(dst_off[i])
                assert np.array_equal(dst[a:a + int(outlen[i])], want_dst[a:a + int(outlen[i])]), (chunk, i)
            elif len(streams[i]):
                assert status[i] == 2, (chunk, i, int(status[i]))       # truncated: still waiting for input
    # zlib wrapper fed byte by byte, and an output buffer that is too small
    z = wl.deflate(text)
    src, src_off, dst_off = wl.pack([z, z, bytes.fromhex("78000300")], [len(text), 100, 10])
    dst, status, outlen = emu_lib.run_resume(src, src_off, dst_off, 1, 1)
    assert status[0] == 0 and dst[:len(text)].tobytes() == text
    assert status[1] == 3 and 0 < outlen[1] <= 100 and dst[len(text):len(text) + int(outlen[1])].tobytes() == text[:int(outlen[1])]
    assert status[2] == 1


def _gzip_member(raw, name=b"", comment=b"", extra=b"", hcrc=False, level=6):
    """A gzip member (RFC 1952) with the optional header fields asked for."""
    flg = (4 if extra else 0) | (8 if name else 0) | (16 if comment else 0) | (2 if hcrc else 0)
    hdr = bytes([0x1f, 0x8b, 8, flg, 1, 2, 3, 4, 0, 3])
    if extra:
        hdr += len(extra).to_bytes(2, "little") + extra
    if name:
        hdr += name + b"\0"
    if comment:
        hdr += comment + b"\0"
    if hcrc:
        hdr += (zlib.crc32(hdr) & 0xffff).to_bytes(2, "little")
    body = wl.deflate(raw, level, False)
    return hdr + body + (zlib.crc32(raw) & 0xffffffff).to_bytes(4, "little") + (len(raw) & 0xffffffff).to_bytes(4, "little")


def test_emulated_integrity_passes():
    """The opt-in checks behind infl_batch_dev_checked on the CPU: Adler-32 of zlib streams and gzip
    framing (header fields skipped, CRC-32 and ISIZE) against Python's zlib; damaged and missing
    trailers are told apart (detail 5 = checksum mismatch, 2 = input ended)."""
    rng = random.Random(85)
    raws = [b"", b"a", b"abc" * 7, wl.text_stream(41, 1023), wl.text_stream(42, 1024), wl.text_stream(43, 1025),
            wl.text_stream(44, 5000), bytes(rng.randrange(256) for _ in range(3333)), bytes(2500)]
    # zlib + Adler-32
    good = [wl.deflate(r) for r in raws]
    assert all(int.from_bytes(c[-4:], "big") == zlib.adler32(r) for c, r in zip(good, raws))
    bad = []
    for c in good[3:6]:
        b = bytearray(c); b[-2] ^= 0x40; bad.append(bytes(b))           # wrong trailer
    short = [good[6][:-1], good[6][:-4]]                                   # trailer cut
    body = bytearray(good[7]); body[len(body) // 2] ^= 1                   # damaged data: whatever it decodes to, never OK with this trailer
    streams = good + bad + short + [bytes(body)]
    caps = [len(r) for r in raws] + [len(r) for r in raws[3:6]] + [5000, 5000, 3333 + 64]
    src, so, do = wl.pack(streams, caps)
    dst, status, outlen, detail, _ = emu_lib.run(src, so, do, 1, 64, 2, check=1)
    for i, r in enumerate(raws):
        assert status[i] == 0 and outlen[i] == len(r) and detail[i] == 0, (i, int(status[i]), int(detail[i]))
        assert dst[int(do[i]):int(do[i]) + len(r)].tobytes() == r
    k = len(raws)
    assert list(status[k:k + 3]) == [-1] * 3 and list(detail[k:k + 3]) == [5] * 3
    assert list(status[k + 3:k + 5]) == [-1, -1] and list(detail[k + 3:k + 5]) == [2, 2]
    assert status[k + 5] == -1
    # gzip
    members = [_gzip_member(raws[0]), _gzip_member(raws[2], name=b"file.txt"), _gzip_member(raws[3], comment=b"c" * 40, hcrc=True),
               _gzip_member(raws[4], extra=b"\x01\x02" * 9, name=b"n"), _gzip_member(raws[5], level=1), _gzip_member(raws[6]),
               _gzip_member(raws[7], name=b"x", comment=b"y", extra=b"zzzz", hcrc=True)]
    mraws = [raws[0], raws[2], raws[3], raws[4], raws[5], raws[6], raws[7]]
    b1 = bytearray(members[5]); b1[-6] ^= 2                                # CRC-32 wrong
    b2 = bytearray(members[5]); b2[-1] ^= 1                                # ISIZE wrong
    extra = [bytes(b1), bytes(b2), members[5][:-3], members[3][:14], b"\x1f\x8b\x07" + members[1][3:], good[3], b""]
    streams = members + extra
    caps = [len(r) for r in mraws] + [5000, 5000, 5000, 1024, 21, 1023, 10]
    src, so, do = wl.pack(streams, caps)
    dst, status, outlen, detail, _ = emu_lib.run(src, so, do, 0, 64, 2, check=2)
    for i, r in enumerate(mraws):
        assert status[i] == 0 and outlen[i] == len(r), (i, int(status[i]), int(detail[i]))
        assert dst[int(do[i]):int(do[i]) + len(r)].tobytes() == r
    k = len(members)
    assert list(status[k:]) == [-1] * 7
    assert list(detail[k:k + 2]) == [5, 5] and detail[k + 2] == 2 and detail[k + 3] == 2     # mismatches; trailer / header cut
    assert detail[k + 4] == 1 and detail[k + 5] == 1 and detail[k + 6] == 2                   # not gzip (twice); nothing there


def test_emulated_png_unfilter():
    """The PNG un-filter kernel under AddressSanitizer with exact-size buffers per image: several bands, rows
    shorter and longer than the 16-byte pieces and the 256-byte windows, every bpp, empty images, failures."""
    oracle = oracle_lib.load()
    rng = np.random.default_rng(5)
    filtered, geom, want = [], [], []
    for i, (w, h, bpp) in enumerate([(20, 10, 4), (1, 1, 1), (3, 1, 8), (5, 2, 3), (15, 66, 1), (16, 64, 1), (17, 3, 2),
                                     (70, 5, 4), (300, 2, 1), (129, 70, 2), (9, 130, 6), (40, 65, 8), (272, 3, 1)]):
        pixels, f, _ = wl.png_filtered(i, w, h, bpp)
        filtered.append(f); geom.append((h, w * bpp, bpp)); want.append(pixels)
    for bpp in (5, 7):
        rows, rowbytes = 9, 5 * bpp
        d = rng.integers(0, 256, size=(rows, rowbytes + 1), dtype=np.uint8)
        d[:, 0] = rng.integers(0, 5, size=rows)
        filtered.append(d.tobytes()); geom.append((rows, rowbytes, bpp)); want.append(oracle.png_unfilter(d.tobytes(), rows, rowbytes, bpp)[1])
    filtered.append(b""); geom.append((0, 64, 4)); want.append(b"")
    filtered.append(bytes([0, 1, 2, 3, 4])); geom.append((5, 0, 1)); want.append(b"")
    for f, g, w_ in zip(filtered, geom, want):
        assert oracle.png_unfilter(f, *g) == (0, w_)
    out, st = emu_lib.run_png(filtered, geom)
    for i, w_ in enumerate(want):
        assert st[i] == 0 and out[i] == w_, (i, geom[i])
    # failures: a filter type above 4 (in the second band), too little input, a failed stream, bpp out of range
    pixels, good, _ = wl.png_filtered(3, 8, 70, 4)
    late = bytearray(good); late[69 * 33] = 9
    out, st = emu_lib.run_png([good, bytes(late), good, good, good, good], [(70, 32, 4)] * 4 + [(70, 32, 0), (70, 32, 9)],
                              in_len=[len(good), len(good), len(good) - 1, len(good), len(good), len(good)],
                              in_status=[0, 0, 0, -1, 0, 0])
    assert list(st) == [0, -1, -1, -1, -1, -1] and out[0] == pixels


def test_emulated_png_unfilter_geometry_sweep():
    """Row lengths around the kernel's 16-byte pieces, its 256-byte windows and the 16-byte fast-path limit,
    scanline counts around its bands of 64, random pixel sizes and filter types, random bytes."""
    oracle = oracle_lib.load()
    rng = np.random.default_rng(77)
    filtered, geom, want = [], [], []
    for rb in (1, 2, 3, 4, 15, 16, 17, 31, 32, 33, 63, 64, 65, 191, 192, 193, 255, 256, 257, 271, 272, 273, 511, 512, 513, 700):
        for rows in (1, 2, 63, 64, 65, 128, 129):
            if rb * rows > 60000:
                continue
            bpp = int(rng.choice([1, 2, 3, 4, 6, 8]))
            d = rng.integers(0, 256, size=(rows, rb + 1), dtype=np.uint8)
            d[:, 0] = rng.integers(0, 5, size=rows)
            rc, w_ = oracle.png_unfilter(d.tobytes(), rows, rb, bpp)
            assert rc == 0
            filtered.append(d.tobytes()); geom.append((rows, rb, bpp)); want.append(w_)
    out, st = emu_lib.run_png(filtered, geom)
    bad = [g for g, a, b, s in zip(geom, out, want, st) if a != b or s != 0]
    assert not bad, bad[:10]


def test_emulated_full_size_streams_and_wide_fuzz(oracle):
    """What the fiber emulator makes affordable: streams of the benchmark's own size (64 KiB text at three levels,
    PNG-like scanlines, noise, runs, binary slices of this interpreter) through every pass under AddressSanitizer,
    and a differential fuzz of a few hundred damaged streams through the default three-pass pipeline, with detail
    codes and input positions compared as well."""
    rng = random.Random(91)
    nrng = np.random.default_rng(92)
    exe = open(os.path.realpath("/usr/bin/python3"), "rb").read(1 << 20)
    raws = [wl.text_stream(40, 65536), wl.text_stream(41, 65536), wl.png_like(2), bytes(nrng.integers(32, 96, 40000, dtype=np.uint8)),
            b"".join(bytes([rng.randrange(256)]) * rng.randint(1, 5000) for _ in range(30)), exe[200000:265536], exe[600000:640000]]
    good = [wl.deflate(raws[0], 6), wl.deflate(raws[0], 1), wl.deflate(raws[1], 9), wl.deflate(raws[1], 6, True, zlib.Z_FIXED)] + \
           [wl.deflate(r, 6) for r in raws[2:]]
    caps = [len(raws[0])] * 2 + [len(raws[1])] * 2 + [len(r) for r in raws[2:]]
    _compare(oracle, good, caps, 1, 64, 2)
    _compare(oracle, good, caps, 1, 16, 1)
    _compare(oracle, good[:3], caps[:3], 1, 32, 0)
    bad, bcaps = [], []
    for t in range(300):
        k = rng.randrange(len(good))
        b, cap = bytearray(good[k]), caps[k]
        what = rng.randrange(5)
        if what == 0:
            b = b[: rng.randrange(2, len(b))]
        elif what == 1:
            for _ in range(rng.randint(1, 3)):
                b[rng.randrange(2, len(b))] ^= 1 << rng.randrange(8)
        elif what == 2:
            cap = rng.randrange(0, cap)
        elif what == 3:
            at = rng.randrange(2, len(b))
            b[at:at + rng.randint(1, 30)] = rng.randbytes(rng.randint(1, 30))
        else:
            b = b[: rng.randrange(2, len(b))]
            cap = rng.randrange(cap // 2, cap + 1)
        bad.append(bytes(b)); bcaps.append(cap)
    _compare(oracle, bad, bcaps, 1, 16, 2)
    _compare(oracle, bad[:100], bcaps[:100], 1, 16, 2, plain=True)


def test_emulated_lane_order_does_not_matter(oracle, monkeypatch):
    """Between two barriers the emulator runs the lanes one after the other, lane 0 first; here from lane 63 down.
    A kernel whose lanes exchange data without a fence in between gives different results in the two orders."""
    monkeypatch.setenv("EMU_ORDER", "reverse")
    test_emulated_kernel_on_fixtures(oracle, 64, 2)
    test_emulated_kernel_on_fixtures(oracle, 16, 1)
    test_emulated_kernel_on_header_cases(oracle, 64, 2)
    test_emulated_kernel_zlib_and_fuzz(oracle)
    test_emulated_parallel_decoder_corner_shapes(oracle)
    test_emulated_parallel_decoder_serial_stretches(oracle)
    test_emulated_incremental_kernel(oracle)
    test_emulated_integrity_passes()
    test_emulated_png_unfilter()


@pytest.mark.parametrize("lanes", [16, 8])
def test_emulated_short_streams_share_wavefronts(oracle, lanes):
    """Streams of under 256 input bytes are left to the serial pass by the parallel one; with 8 or 16 lanes per
    stream that pass looks at windows of 64 streams and shares the marked ones out among its groups.  Short and long
    streams mixed in every pattern of a window (none, one, all, the last window incomplete), failing ones among them."""
    rng = random.Random(400 + lanes)
    streams, caps = [], []

    def short(i):
        raw = wl.text_stream(7000 + i, rng.choice([0, 1, 5, 60, 200, 400, 600]))
        c = wl.deflate(raw, rng.choice([1, 6, 9]), True, rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED]))
        what = rng.random()
        if what < 0.15 and len(c) > 4:
            b = bytearray(c); b[rng.randrange(2, len(b))] ^= 1 << rng.randrange(8); c = bytes(b)
        elif what < 0.25:
            c = c[: rng.randrange(0, len(c))]
        return c, (len(raw) if what < 0.9 else rng.randrange(0, len(raw) + 1))

    def long_(i):
        raw = wl.text_stream(7100 + i, rng.choice([1500, 3000]))
        return wl.deflate(raw), len(raw)

    patterns = [[1] * 64, [0] * 64, [0] * 63 + [1], [1] + [0] * 63, [1, 0] * 32, [0, 0, 0, 1] * 16, [0] * 20 + [1] * 5 + [0] * 39,
                [rng.randrange(2) for _ in range(64)], [1] * 5]                # (the last window has 5 streams)
    for w, pat in enumerate(patterns):
        for k, is_short in enumerate(pat):
            c, cap = short(64 * w + k) if is_short else long_(64 * w + k)
            streams.append(c); caps.append(cap)
    assert sum(1 for c in streams if len(c) >= 256) > 100 and sum(1 for c in streams if 0 < len(c) < 256) > 100
    status, _, _ = _compare(oracle, streams, caps, 1, lanes, 2, plain=True)
    assert (status == 0).sum() > 300 and (status != 0).sum() > 10
    _compare(oracle, streams, caps, 1, lanes, 2)
    # the two passes alone (variant 4, a diagnostic): every short stream that decodes was finished by the windowed pass
    # (its own mark), none was passed over and left to the last pass
    src, src_off, dst_off = wl.pack(streams, caps)
    _, st4, _, _, _ = emu_lib.run(src, src_off, dst_off, 1, lanes, 4, plain=True)
    for i, c in enumerate(streams):
        if 8 <= len(c) < 256 and status[i] == 0:
            assert st4[i] == 0x40000003, (i, hex(int(st4[i])))
        elif len(c) >= 256 and status[i] == 0:
            assert st4[i] == 0, (i, hex(int(st4[i])))


def _pieces_case(rng, n_streams, sizes):
    """Streams cut into pieces that lie scattered through a blob (a chunked container): -> blob, piece_off, piece_len,
    piece_dst, src_off, the streams."""
    streams = [bytes(rng.randrange(256) for _ in range(rng.choice(sizes))) for _ in range(n_streams)]
    blob, piece_off, piece_len, piece_dst, src_off = bytearray(), [], [], [], [0]
    at = 0
    for s in streams:
        k = 0
        while k < len(s) or (not s and k == 0):
            n = min(len(s) - k, rng.choice([0, 1, 2, 15, 16, 17, 63, 100, 1023, 1024, 1025, 3000]))
            blob += bytes(rng.randrange(256) for _ in range(rng.randrange(0, 13)))        # chunk headers, CRCs, other chunks
            piece_off.append(len(blob)); piece_len.append(n); piece_dst.append(at + k)
            blob += s[k:k + n]
            k += n
            if not s:
                break
        at += len(s)
        src_off.append(at)
    blob += bytes(rng.randrange(256) for _ in range(5))
    return bytes(blob), piece_off, piece_len, piece_dst, src_off, streams


def test_emulated_gather_of_pieces():
    """infl_gather_kernel: pieces of every length and alignment end up where infl_include() calls in the same order would
    put them (one behind the other per stream); nothing outside blob and arena is touched (exact-size buffers, ASan)."""
    rng = random.Random(77)
    blob, po, pl, pd, so, streams = _pieces_case(rng, 40, [0, 1, 5, 16, 31, 100, 1024, 4097, 9000])
    out = emu_lib.run_gather(blob, po, pl, pd, so[-1])
    assert out.tobytes() == b"".join(streams)
    assert emu_lib.run_gather(b"", [], [], [], 0).size == 0
    # pieces that leave gaps in the arena: the gaps stay as they were
    out = emu_lib.run_gather(bytes(range(200)), [3, 50], [20, 17], [1, 40], 64)
    want = bytearray(b"\xee" * 64); want[1:21] = bytes(range(3, 23)); want[40:57] = bytes(range(50, 67))
    assert out.tobytes() == bytes(want)
    # pieces whose descriptors reach outside the blob or the arena (the kernel is told both sizes) are refused whole:
    # source past the end, source offset far outside, target past the end, offset + length overflowing 64 bits
    out, st = emu_lib.run_gather(bytes(range(200)), [3, 190, 1 << 40, 10, 20, (1 << 64) - 4], [20, 11, 4, 30, 17, 8],
                                 [1, 21, 30, 40, 48, 0], 64, want_status=True)
    assert st.tolist() == [0, -1, -1, -1, -1, -1]
    want = bytearray(b"\xee" * 64); want[1:21] = bytes(range(3, 23))
    assert out.tobytes() == bytes(want)


def test_emulated_serial_passes_alone_match_shapes(oracle):
    """Variant 5 = the serial passes alone, the first one through the windowed kernel: matches at distances around 2 KiB,
    matches that repeat their source (distance < length), runs, stored blocks between Huffman blocks."""
    rng = random.Random(55)
    raws = []
    raws.append(wl.text_stream(1, 9000))                                          
    raws.append((b"abcdefghij" * 30 + bytes(rng.randrange(256) for _ in range(40))) * 12)   
    raws.append(bytes(rng.randrange(256) for _ in range(1700)) * 3)               
    raws.append(bytes(rng.randrange(256) for _ in range(1791)) * 3)               
    raws.append(bytes(rng.randrange(256) for _ in range(2049)) * 3)               
    raws.append(b"xy" * 3000 + b"q" * 2500 + b"xyz" * 900)                        # repeating sources, runs
    raws.append(bytes(5000))
    raws.append(wl.text_stream(2, 300) + bytes(rng.randrange(256) for _ in range(700)) + wl.text_stream(2, 300) * 4)
    comps = [wl.deflate(r, lvl) for r in raws for lvl in (1, 6, 9)]
    caps = [len(r) for r in raws for _ in range(3)]
    # stored blocks between compressed ones: flush with incompressible bytes in between
    c = zlib.compressobj(6)
    parts, mixed = [], b""
    for k in range(6):
        piece = wl.text_stream(10 + k, 700) if k % 2 == 0 else bytes(rng.randrange(256) for _ in range(900))
        mixed += piece
        parts.append(c.compress(piece) + c.flush(zlib.Z_FULL_FLUSH))
    comps.append(b"".join(parts) + c.flush()); caps.append(len(mixed))
    for lanes in (16, 8):
        _compare(oracle, comps, caps, 1, lanes, 5)
    _compare(oracle, comps, [max(0, c_ - 5) for c_ in caps], 1, 16, 5)
