"""GPU parity tests: the HIP engine, called through its C ABI, against the CPU oracle and the
committed golden fixtures.  Integer/byte work: the bar is bit-exact (return code, output
position, output bytes)."""
import json
import os
import random
import zlib

import numpy as np
import pytest

import workloads as wl
from defl_amd import capi
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "defl_testdata")
ALL = sorted(os.listdir(os.path.join(DATA, "compressed")))
PAIRS = [f for f in ALL if os.path.exists(os.path.join(DATA, "raw", f))]
DEFAULT_VARIANT = 2
VARIANTS = [(64, 0), (32, 0), (16, 0), (8, 0), (64, 1), (32, 1), (16, 1), (8, 1), (16, 2), (64, 2), (16, 5), (8, 5)]


def _read(kind, name):
    with open(os.path.join(DATA, kind, name), "rb") as f:
        return f.read()


def _check_batch(engine, oracle, streams, caps, flags, lanes=64, what="", variant=0):
    """Runs one batch on the GPU (device-resident path) and on the oracle; compares everything."""
    src, src_off, dst_off = wl.pack(streams, caps)
    want_dst, want_status, want_outlen = oracle.infl_batch(src, src_off, dst_off, flags)
    assert engine.lib.infl_gpu_set_lanes_per_stream(lanes) == lanes
    assert engine.lib.infl_gpu_set_table_variant(variant) == variant
    b = engine.device_batch(src, src_off, dst_off)
    try:
        # first the plain entry point (no detail codes asked for: the parallel pass may report a
        # failing stream itself), then the one with detail codes (the serial passes judge it)
        b.run(flags, detail=False)
        dst0, status0, outlen0, _ = b.results()
        b.run(flags)
        dst, status, outlen, detail = b.results()
    finally:
        b.free()
        engine.lib.infl_gpu_set_lanes_per_stream(16)
        engine.lib.infl_gpu_set_table_variant(DEFAULT_VARIANT)
    bad = np.nonzero((status != want_status) | (outlen != want_outlen))[0]
    assert bad.size == 0, (what, "stream", int(bad[0]), int(status[bad[0]]), int(want_status[bad[0]]),
                           int(outlen[bad[0]]), int(want_outlen[bad[0]]), int(detail[bad[0]]))
    bad = np.nonzero((status0 != want_status) | (outlen0 != want_outlen))[0]
    assert bad.size == 0, (what, "plain entry point, stream", int(bad[0]), int(status0[bad[0]]),
                           int(want_status[bad[0]]), int(outlen0[bad[0]]), int(want_outlen[bad[0]]))
    for i in range(len(streams)):
        a, n = int(dst_off[i]), int(outlen[i])
        for got in (dst, dst0):
            if not np.array_equal(got[a:a + n], want_dst[a:a + n]):
                k = int(np.nonzero(got[a:a + n] != want_dst[a:a + n])[0][0])
                raise AssertionError((what, "stream", i, "first bad byte", k, "of", n))
    return status, outlen, detail


def test_device_present(engine):
    assert engine.lib.infl_gpu_device_count() >= 1
    assert engine.lib.infl_gpu_init(0) == 0


@pytest.mark.parametrize("name", PAIRS)
def test_file_pair_infl_buf(engine, name):
    """The reference's own file test (test_files.c:213-235) through the drop-in API."""
    raw, comp = _read("raw", name), _read("compressed", name)
    rc, pos, out = engine.infl_buf(comp, len(raw) + 1000, 0)
    assert rc == capi.UNZ_OK
    assert pos == len(raw) and out == raw


@pytest.mark.parametrize("lanes,variant", VARIANTS)
def test_all_fixtures_one_batch(engine, oracle, lanes, variant):
    streams = [_read("compressed", n) for n in ALL]
    caps = [len(_read("raw", n)) + 1000 if n in PAIRS else 70000 for n in ALL]
    status, outlen, _ = _check_batch(engine, oracle, streams, caps, 0, lanes, "fixtures", variant)
    for n, s, o in zip(ALL, status, outlen):
        if n in PAIRS:
            assert s == 0 and o == len(_read("raw", n)), n


def test_kats(engine):
    with open(os.path.join(HERE, "golden", "kats.json")) as f:
        kats = json.load(f)["vectors"]
    for k in kats:
        rc, pos, out = engine.infl_buf(bytes.fromhex(k["src"]), k["dstlen"], k["flags"])
        if k["expect"] == "err":
            assert rc != capi.UNZ_OK, k["name"]
        else:
            assert rc == capi.UNZ_OK and out == bytes.fromhex(k["out"]), k["name"]


def test_empty_input(engine):
    assert engine.infl_buf(b"", 16, 0)[:2] == (capi.UNZ_OK, 0)
    assert engine.infl_buf(b"", 16, 1)[:2] == (capi.UNZ_OK, 0)


def _corpus(rng, n):
    words = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(2, 10)))
             for _ in range(300)]
    out = bytearray()
    while len(out) < n:
        out += rng.choice(words) + rng.choice([b" ", b" ", b", ", b".\n"])
    return bytes(out[:n])


def _zlib_cases(seed):
    rng = random.Random(seed)
    for level in (0, 1, 6, 9):
        for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
            for size in (0, 1, 2, 100, 4097, 70000, 200000):
                for kind in ("text", "random", "zeros", "mixed"):
                    if kind == "text":
                        raw = _corpus(rng, size)
                    elif kind == "random":
                        raw = rng.randbytes(size)
                    elif kind == "zeros":
                        raw = bytes(size)
                    else:
                        raw = _corpus(rng, size // 2) + rng.randbytes(size - size // 2)
                    yield raw, level, strategy


@pytest.mark.parametrize("lanes,variant", VARIANTS)
@pytest.mark.parametrize("wrap", [0, 1])
def test_zlib_streams_batch(engine, oracle, wrap, lanes, variant):
    """Valid streams of every block type; exact capacity, one byte short, generous."""
    streams, caps, raws = [], [], []
    for raw, level, strategy in _zlib_cases(7 + wrap):
        comp = wl.deflate(raw, level, bool(wrap), strategy)
        for cap in (len(raw), max(0, len(raw) - 1), len(raw) + 300):
            streams.append(comp)
            caps.append(cap)
            raws.append(raw)
    status, outlen, detail = _check_batch(engine, oracle, streams, caps, wrap, lanes, "zlib", variant)
    for i, raw in enumerate(raws):
        if caps[i] >= len(raw):
            assert status[i] == 0 and outlen[i] == len(raw)
        else:
            assert status[i] == -1 and detail[i] == capi.D_FULL


@pytest.mark.parametrize("lanes,variant", [(64, 0), (16, 0), (64, 1), (16, 1), (8, 1), (16, 2), (16, 5), (8, 5)])
def test_fuzz_mutations_and_truncations(engine, oracle, lanes, variant):
    """Differential fuzz against the oracle on malformed input: flipped bits, truncation,
    random bytes.  Return code, output position and bytes must match stream by stream."""
    rng = random.Random(4242)
    base = []
    for i in range(12):
        raw = _corpus(rng, rng.randint(50, 30000))
        base.append(wl.deflate(raw, rng.choice([1, 6, 9]), False,
                               rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE])))
    base.append(wl.stored_stream(1, 5000, False)[1])
    base.append(wl.stored_stream(2, 70000, False)[1] )
    streams = []
    for trial in range(6000):
        b = bytearray(rng.choice(base))
        mode = trial % 4
        if mode == 0:
            b = b[:rng.randint(0, len(b))]
        elif mode == 1:
            for _ in range(rng.randint(1, 4)):
                if b:
                    b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        elif mode == 2:
            b = b[:rng.randint(1, min(len(b), 400))]
            for _ in range(rng.randint(1, 3)):
                b[rng.randrange(len(b))] = rng.randrange(256)
        else:
            b = bytearray(rng.randbytes(rng.randint(1, 300)))
        streams.append(bytes(b))
    caps = [rng.choice([0, 10, 1000, 40000, 80000]) for _ in streams]
    for flags in (0, 1):
        _check_batch(engine, oracle, streams, caps, flags, lanes, "fuzz flags=%d" % flags, variant)


def test_dynamic_header_edge_cases(engine, oracle):
    """Hand-built dynamic headers: over-subscribed and incomplete code-length sets, repeat
    codes at position 0 and past the end, symbols without an entry, table-capacity overflow."""
    from tests.deflate_builder import header_cases
    streams = header_cases()
    caps = [4096] * len(streams)
    for lanes, variant in VARIANTS:
        _check_batch(engine, oracle, streams, caps, 0, lanes, "headers", variant)


def test_text_batch_roundtrip_full_size(engine, oracle):
    """Config-2 shaped batch (64 KiB text streams, zlib level 6): every stream against the raw text it was
    made from, and a sample (status, output position, bytes) against the oracle."""
    n = 512
    raws = [wl.text_stream(i) for i in range(n)]
    streams = [wl.deflate(r) for r in raws]
    src, src_off, dst_off = wl.pack(streams, [65536] * n)
    b = engine.device_batch(src, src_off, dst_off)
    try:
        b.run(1)
        dst, status, outlen, _ = b.results()
    finally:
        b.free()
    assert (status == 0).all() and (outlen == 65536).all()
    assert dst[:n * 65536].tobytes() == b"".join(raws)
    for i in (0, 1, 77, 255, n - 1):
        rc, pos, out, _ = oracle.infl(streams[i], 65536, 1)
        assert (rc, pos) == (int(status[i]), int(outlen[i])) and out[:pos] == dst[i * 65536:i * 65536 + pos].tobytes()


def test_host_batch_and_many(engine, oracle):
    """infl_batch() on host memory and infl_many() over streams fed piecewise with
    infl_include() (PNG-IDAT style) give the same bytes as the oracle on the joined input."""
    import ctypes as C
    n = 24
    raws = [wl.png_like(i) for i in range(n)]
    comps = [wl.deflate(r) for r in raws]
    src, src_off, dst_off = wl.pack(comps, [len(r) for r in raws])
    dst, status, outlen = engine.batch_host(src, src_off, dst_off, 1)
    assert (status == 0).all()
    assert dst[:int(dst_off[-1])].tobytes() == b"".join(raws)

    lib = engine.lib
    outs = [(C.c_uint8 * len(r))() for r in raws]
    keep = []
    sts = (C.c_void_p * n)()
    for i in range(n):
        sts[i] = lib.infl_init(outs[i], len(raws[i]), 1)
        for piece in wl.idat_pieces(i, comps[i]):
            buf = (C.c_uint8 * len(piece)).from_buffer_copy(piece)
            keep.append(buf)
            lib.infl_include(sts[i], buf, len(piece))
    st_arr = (C.c_int32 * n)()
    assert lib.infl_many(sts, n, st_arr) == 0
    for i in range(n):
        assert st_arr[i] == 0
        assert lib.infl_output_pos(sts[i]) == len(raws[i])
        assert bytes(outs[i]) == raws[i]
        lib.infl_destroy(sts[i])


def test_host_batch_sliced_large(engine, oracle):
    """A host-memory batch large enough for the sliced path of infl_batch() (slices of whole
    streams uploaded, decoded and downloaded in turn): ragged stream sizes and capacities,
    damaged and truncated streams in between, and the bytes behind every stream's output left
    as the caller had them."""
    rng = random.Random(77)
    base_raw = [wl.text_stream(i, rng.choice([700, 9000, 65536, 150000])) for i in range(48)]
    base = [wl.deflate(r) for r in base_raw]
    streams, caps = [], []
    for i in range(3000):
        k = rng.randrange(len(base))
        s = base[k]
        cap = len(base_raw[k]) + rng.choice([0, 0, 1, 300])
        kind = rng.randrange(20)
        if kind == 0:
            s = s[: rng.randrange(len(s))]
        elif kind == 1:
            b = bytearray(s)
            b[rng.randrange(2, len(b))] ^= 1 << rng.randrange(8)
            s = bytes(b)
        elif kind == 2:
            cap = rng.randrange(len(base_raw[k]))
        streams.append(s)
        caps.append(cap)
    src, src_off, dst_off = wl.pack(streams, caps)
    assert int(dst_off[-1]) > (96 << 20)
    want_dst, want_status, want_outlen = oracle.infl_batch(src, src_off, dst_off, 1, 8)
    n = len(streams)
    dst = np.full(int(dst_off[-1]) + 64, 0xA5, dtype=np.uint8)
    status = np.full(n, -77, dtype=np.int32)
    outlen = np.zeros(n, dtype=np.uint32)
    so = np.ascontiguousarray(src_off, dtype=np.uint64)
    do = np.ascontiguousarray(dst_off, dtype=np.uint64)
    for rep in range(2):                                 # the second call reuses the workspace
        rc = engine.lib.infl_batch(src.ctypes.data, so.ctypes.data, dst.ctypes.data, do.ctypes.data, 1, n,
                                   status.ctypes.data, outlen.ctypes.data)
        assert rc == 0
        assert np.array_equal(status, want_status) and np.array_equal(outlen, want_outlen)
        ok = want_status == 0
        assert 0.8 < ok.mean() < 1.0
        for i in range(n):
            a, m, e = int(dst_off[i]), int(outlen[i]), int(dst_off[i + 1])
            if ok[i]:
                assert np.array_equal(dst[a:a + m], want_dst[a:a + m]), i
                assert (dst[a + m:e] == 0xA5).all(), i
        assert (dst[int(dst_off[-1]):] == 0xA5).all()


def test_chunked_include_like_reference(engine):
    """test_file_chunked of the reference (test_files.c:260-331): 1/2/4/8-byte pieces."""
    for name in ("hello", "large_text_64k", "multiple_blocks", "png_simulation", "uncompressed_multi"):
        import ctypes as C
        raw, comp = _read("raw", name), _read("compressed", name)
        out = (C.c_uint8 * (len(raw) + 1000))()
        st = engine.lib.infl_init(out, len(raw) + 1000, 0)
        pos, keep = 0, []
        while pos < len(comp):
            n = min(1 << (pos % 4), len(comp) - pos)
            buf = (C.c_uint8 * n).from_buffer_copy(comp[pos:pos + n])
            keep.append(buf)
            engine.lib.infl_include(st, buf, n)
            pos += n
        assert engine.lib.infl(st) == capi.UNZ_OK
        assert engine.lib.infl_output_pos(st) == len(raw)
        assert bytes(out[:len(raw)]) == raw
        engine.lib.infl_destroy(st)


def test_streaming_api(engine):
    """infl_stream() fed in 64..1024-byte pieces (test_files.c:688-790) and the zlib-header /
    output-full cases: UNZ_UNFINISHED until complete, UNZ_EFULL resumable after a resize."""
    import ctypes as C
    lib = engine.lib
    for name, step in (("hello", 64), ("multiple_blocks", 128), ("large_text_64k", 256), ("log_pattern", 512)):
        raw, comp = _read("raw", name), _read("compressed", name)
        out = (C.c_uint8 * (len(raw) + 16))()
        st = lib.infl_init(out, len(raw) + 16, 0)
        rc = capi.UNZ_UNFINISHED
        keep = []
        for pos in range(0, len(comp), step):
            piece = comp[pos:pos + step]
            buf = (C.c_uint8 * len(piece)).from_buffer_copy(piece)
            keep.append(buf)
            rc = lib.infl_stream(st, buf, len(piece))
            assert rc in (capi.UNZ_OK, capi.UNZ_UNFINISHED)
            if pos + step < len(comp) and rc == capi.UNZ_UNFINISHED:
                assert lib.infl_output_pos(st) <= len(raw)
        assert rc == capi.UNZ_OK
        assert bytes(out[:len(raw)]) == raw and lib.infl_output_pos(st) == len(raw)
        lib.infl_destroy(st)

    raw = wl.text_stream(5, 20000)
    comp = wl.deflate(raw)
    small = (C.c_uint8 * 5000)()
    st = lib.infl_init(small, 5000, 1)
    src = (C.c_uint8 * len(comp)).from_buffer_copy(comp)
    assert lib.infl_stream(st, src, 1) == capi.UNZ_UNFINISHED      # waits for the 2-byte header
    assert lib.infl_stream(st, C.byref(src, 1), len(comp) - 1) == capi.UNZ_EFULL
    got = lib.infl_output_pos(st)
    assert 0 < got <= 5000 and bytes(small[:got]) == raw[:got]
    big = (C.c_uint8 * 20000)()
    C.memmove(big, small, got)
    assert lib.infl_resize_output(st, big, 20000) == capi.UNZ_OK
    assert lib.infl_stream(st, None, 0) == capi.UNZ_OK
    assert bytes(big) == raw
    lib.infl_destroy(st)


def test_streaming_bytewise_png_simulation(engine):
    """The reference's byte-wise streaming test (test/test_files.c:792-869, :1120-1129): png_simulation
    (30 628 compressed bytes) fed to infl_stream() one byte per call, then empty calls until it is done.
    Every call decodes on from where the last one stopped; bound: a minute for the 30 628 calls."""
    import ctypes as C
    import time
    lib = engine.lib
    raw, comp = _read("raw", "png_simulation"), _read("compressed", "png_simulation")
    out = (C.c_uint8 * (len(raw) + 1000))()
    st = lib.infl_init(out, len(raw) + 1000, 0)
    src = (C.c_uint8 * len(comp)).from_buffer_copy(comp)
    t0 = time.time()
    rc, pos, last_out = capi.UNZ_UNFINISHED, 0, 0
    while pos < len(comp) and rc == capi.UNZ_UNFINISHED:
        rc = lib.infl_stream(st, C.byref(src, pos), 1)
        pos += 1
        now = lib.infl_output_pos(st)
        assert now >= last_out                       # the output only ever grows
        last_out = now
    tries = 0
    while rc == capi.UNZ_UNFINISHED and tries < 32:
        rc = lib.infl_stream(st, None, 0)
        tries += 1
    took = time.time() - t0
    assert rc == capi.UNZ_OK and pos == len(comp)
    assert bytes(out[:len(raw)]) == raw and lib.infl_output_pos(st) == len(raw)
    assert took < 60.0, took
    lib.infl_destroy(st)


def test_streaming_edge_vectors(engine):
    """The three streaming edge inputs of the reference's test program (test/test_files.c:871-1029)."""
    import ctypes as C
    import json
    lib = engine.lib
    codes = {"ok": capi.UNZ_OK, "unfinished": capi.UNZ_UNFINISHED}
    with open(os.path.join(HERE, "golden", "kats.json")) as f:
        vectors = json.load(f)["streaming"]
    assert len(vectors) == 3
    for v in vectors:
        data = bytes.fromhex(v["src"])
        out = (C.c_uint8 * v["dstlen"])()
        st = lib.infl_init(out, v["dstlen"], v["flags"])
        src = (C.c_uint8 * len(data)).from_buffer_copy(data)
        rc, fed = capi.UNZ_UNFINISHED, 0
        while fed < len(data) and rc == capi.UNZ_UNFINISHED:
            n = min(v["feed"], len(data) - fed)
            rc = lib.infl_stream(st, C.byref(src, fed), n)
            fed += n
        assert rc in [codes[c] for c in v["expect_rc"]], (v["name"], rc)
        want = bytes.fromhex(v["out_prefix"])
        assert bytes(out[:len(want)]) == want, v["name"]
        lib.infl_destroy(st)


def test_streaming_resumes_inside_blocks(engine, oracle):
    """infl_stream() against the one-shot decode on streams of several kinds, fed in pieces of 1 to 4096
    bytes: same bytes, UNZ_OK exactly when the final block is complete, UNZ_UNFINISHED before; a
    truncated stream stays UNZ_UNFINISHED, a corrupted one turns UNZ_ERR; infl_reset() starts over."""
    import ctypes as C
    lib = engine.lib
    rng = random.Random(91)
    text = wl.text_stream(31, 50000)
    c = zlib.compressobj(6)
    flushed = b"".join(c.compress(text[i:i + 3000]) + c.flush(zlib.Z_SYNC_FLUSH) for i in range(0, 30000, 3000)) + c.flush()
    cases = [(wl.deflate(text), text, 1), (wl.deflate(text, 6, False, zlib.Z_FIXED), text, 0), (flushed, text[:30000], 1),
             (wl.stored_stream(3, 70000)[1], wl.stored_stream(3, 70000)[0], 1),
             (wl.deflate(bytes(40000) + text[:5000], 9), bytes(40000) + text[:5000], 1)]
    st = None
    for comp, raw, flags in cases:
        for step in (1 if len(comp) < 3000 else 7, 333, 4096):
            out = (C.c_uint8 * len(raw))()
            if st is None:
                st = lib.infl_init(out, len(raw), flags)
            else:
                lib.infl_reset(st, out, len(raw), flags)
            src = (C.c_uint8 * len(comp)).from_buffer_copy(comp)
            rc, fed = capi.UNZ_UNFINISHED, 0
            while fed < len(comp) and rc == capi.UNZ_UNFINISHED:
                n = min(step, len(comp) - fed)
                rc = lib.infl_stream(st, C.byref(src, fed), n)
                fed += n
            # done exactly when the final block is in: the zlib trailer (4 bytes) is never read
            assert rc == capi.UNZ_OK and fed + step > len(comp) - (4 if flags else 0), (len(comp), step, fed, rc)
            assert lib.infl_output_pos(st) == len(raw)
            assert bytes(out) == raw
    # truncated: never more than UNZ_UNFINISHED; corrupted: UNZ_ERR
    comp = wl.deflate(text)
    out = (C.c_uint8 * len(text))()
    lib.infl_reset(st, out, len(text), 1)
    src = (C.c_uint8 * len(comp)).from_buffer_copy(comp)
    assert lib.infl_stream(st, src, len(comp) - 9) == capi.UNZ_UNFINISHED
    assert lib.infl_stream(st, None, 0) == capi.UNZ_UNFINISHED
    got = lib.infl_output_pos(st)
    assert 0 < got < len(text) and bytes(out[:got]) == text[:got]
    assert lib.infl_stream(st, C.byref(src, len(comp) - 9), 9) == capi.UNZ_OK and bytes(out) == text
    bad = bytearray(comp)
    bad[len(bad) // 2] ^= 0x55
    want_rc, _, _, _ = oracle.infl(bytes(bad), len(text), 1)
    lib.infl_reset(st, out, len(text), 1)
    srcb = (C.c_uint8 * len(bad)).from_buffer_copy(bytes(bad))
    rc = capi.UNZ_UNFINISHED
    for fed in range(0, len(bad), 1000):
        rc = lib.infl_stream(st, C.byref(srcb, fed), min(1000, len(bad) - fed))
        if rc != capi.UNZ_UNFINISHED:
            break
    assert (rc == capi.UNZ_OK) == (want_rc == 0)
    lib.infl_destroy(st)


def test_failing_streams_taken_up_where_they_fail(engine, oracle):
    """Large streams that fail somewhere inside (cut off, damaged, too little room): with detail codes asked for,
    the parallel pass notes where and the last pass goes on from there; status, output position and detail code
    must be what the oracle gives for a decode from the start, at every cut."""
    rng = random.Random(21)
    streams, caps = [], []
    bases = [(wl.text_stream(600 + k, 65536), None) for k in range(4)]
    bases.append((wl.png_like(3), None))
    bases.append((bytes(rng.randrange(32, 40) for _ in range(50000)), None))
    bases = [(r, wl.deflate(r, 6 if k % 2 else 1)) for k, (r, _) in enumerate(bases)]
    # several blocks per stream: the note names the block the failure lies in
    multi = zlib.compressobj(6)
    parts = [multi.compress(bases[0][0][:20000]), multi.flush(zlib.Z_FULL_FLUSH), multi.compress(bases[1][0][:30000]),
             multi.flush(zlib.Z_SYNC_FLUSH), multi.compress(bases[2][0][:15536]), multi.flush()]
    bases.append((bases[0][0][:20000] + bases[1][0][:30000] + bases[2][0][:15536], b"".join(parts)))
    for raw, comp in bases:
        for _ in range(12):
            kind = rng.randrange(4)
            c, cap = bytearray(comp), len(raw)
            if kind == 0:
                c = c[: rng.randrange(8, len(c) - 1)]
            elif kind == 1:
                at = rng.randrange(40, len(c) - 4)
                c[at] ^= 1 << rng.randrange(8)
            elif kind == 2:
                cap = rng.randrange(0, len(raw))
            else:
                c = c[: rng.randrange(8, len(c) - 1)]
                cap = rng.randrange(len(raw) // 2, len(raw) + 1)
            streams.append(bytes(c)); caps.append(cap)
    status, outlen, detail = _check_batch(engine, oracle, streams, caps, 1, 16, "failing streams", 2)
    assert (status != 0).sum() >= len(streams) * 3 // 4
    for i, (c, cap) in enumerate(zip(streams, caps)):
        rc, pos, _, det = oracle.infl(c, cap, 1)
        assert (int(status[i]), int(outlen[i]), int(detail[i])) == (rc, pos, det), (i, len(c), cap)


@pytest.mark.parametrize("lanes", [16, 8])
def test_short_streams_among_long_ones(engine, oracle, lanes):
    """The parallel pass leaves streams of under 256 input bytes to the serial pass behind it, whose wavefronts share
    the marked streams of a window of 64 out by rank.  20 000 streams: runs of short ones, runs of long ones, single
    ones among the others, damaged, cut short and with too little room; the count is not a multiple of 64."""
    rng = random.Random(320 + lanes)
    base = []
    for i in range(300):
        raw = wl.text_stream(9000 + i, rng.choice([0, 1, 7, 40, 130, 260, 400, 520, 700, 1100, 2500, 9000]))
        base.append((raw, wl.deflate(raw, rng.choice([1, 6, 9]), True, rng.choice([zlib.Z_DEFAULT_STRATEGY] * 3 + [zlib.Z_FIXED]))))
    short = [b for b in base if len(b[1]) < 256]
    long_ = [b for b in base if len(b[1]) >= 256]
    assert len(short) > 100 and len(long_) > 80
    streams, caps = [], []
    while len(streams) < 20003:
        kind, run = rng.randrange(4), rng.choice([1, 1, 2, 5, 17, 64, 150])
        for _ in range(run):
            raw, c = rng.choice(short if kind == 0 else long_ if kind == 1 else base)
            what = rng.random()
            if what < 0.05 and len(c) > 4:
                b = bytearray(c); b[rng.randrange(2, len(b))] ^= 1 << rng.randrange(8); c = bytes(b)
            elif what < 0.08:
                c = c[: rng.randrange(0, len(c) + 1)]
            streams.append(c)
            caps.append(len(raw) if what < 0.95 else rng.randrange(0, len(raw) + 1))
    streams, caps = streams[:20003], caps[:20003]
    status, _, _ = _check_batch(engine, oracle, streams, caps, 1, lanes, "short among long", 2)
    assert (status == 0).sum() > 15000 and (status != 0).sum() > 500


@pytest.mark.parametrize("n", [6143, 6144])
def test_short_streams_either_side_of_the_batch_size_that_routes_them(engine, oracle, n):
    """Batches of under 6144 streams keep short streams in the parallel pass (its other instantiation), larger ones
    leave them to the serial pass: same results either side."""
    rng = random.Random(n)
    base = [(r, wl.deflate(r)) for r in (wl.text_stream(9500 + i, rng.choice([0, 3, 90, 200, 350, 500, 900])) for i in range(64))]
    streams, caps = [], []
    for i in range(n):
        raw, c = base[rng.randrange(64)]
        if i % 97 == 5 and len(c) > 6:
            c = c[: len(c) // 2]
        streams.append(c); caps.append(len(raw) if i % 89 else max(0, len(raw) - 7))
    status, _, _ = _check_batch(engine, oracle, streams, caps, 1, 16, "batch of %d" % n, 2)
    assert (status == 0).sum() > n * 0.9 and (status != 0).sum() > 20
