"""Device-side gather of input pieces (infl_gather_dev): what a run of infl_include() calls does to a stream's logical
input (reference src/infl/mem.c:114-216), for pieces that already lie in device memory.  The expected result is the
pieces one behind the other (the reference's own chunked tests check exactly that through the decode,
test/test_files.c:1091-1103); the decode behind it is compared with the CPU oracle."""
import struct
import zlib

import numpy as np
import pytest
import random

import workloads as wl
from tests import oracle_lib
from tests.test_kernel_emulation import _pieces_case

pytestmark = pytest.mark.gpu


def test_pieces_of_every_length_and_alignment(engine):
    assert engine.lib.infl_gpu_init(0) == 0
    rng = random.Random(78)
    blob, po, pl, pd, so, streams = _pieces_case(rng, 300, [0, 1, 5, 16, 31, 100, 1024, 4097, 9000, 70000])
    n = len(streams)
    src0 = np.full(so[-1], 0xEE, dtype=np.uint8)                       # the arena before the gather
    b = engine.device_batch(src0, np.array(so, dtype=np.uint64), np.zeros(n + 1, dtype=np.uint64))
    try:
        engine.gather_into(b, blob, po, pl, pd)
        got = np.zeros(max(1, so[-1]), dtype=np.uint8)
        assert engine.lib.infl_gpu_download(got.ctypes.data, b.d_src, so[-1]) == 0
    finally:
        b.free()
    assert got[:so[-1]].tobytes() == b"".join(streams)
    assert engine.lib.infl_gather_dev(None, None, None, None, 0, None, None) == 0      # nothing to do
    assert engine.lib.infl_gather_dev(None, None, None, None, 5, None, None) == -1


def test_more_pieces_than_workgroups_and_one_large_piece(engine):
    """100 000 pieces of 0..40 bytes (the launch has at most 65 536 workgroups, which take the pieces in turn) and one
    piece of 3 MiB (a wavefront goes through it 1 KiB at a time)."""
    assert engine.lib.infl_gpu_init(0) == 0
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 41, size=100000).astype(np.uint32)
    lens[777] = 3 << 20
    gaps = rng.integers(0, 9, size=lens.size).astype(np.uint64)
    piece_off = np.cumsum(gaps + np.concatenate([[0], lens[:-1]]).astype(np.uint64)).astype(np.uint64)
    blob = rng.integers(0, 256, size=int(piece_off[-1] + lens[-1] + 3), dtype=np.uint8)
    piece_dst = np.concatenate([[0], np.cumsum(lens.astype(np.uint64))[:-1]]).astype(np.uint64)
    total = int(lens.sum())
    b = engine.device_batch(np.zeros(total, dtype=np.uint8), np.array([0, total], dtype=np.uint64), np.zeros(2, dtype=np.uint64))
    try:
        engine.gather_into(b, blob.tobytes(), piece_off, lens, piece_dst)
        got = np.zeros(total, dtype=np.uint8)
        assert engine.lib.infl_gpu_download(got.ctypes.data, b.d_src, total) == 0
    finally:
        b.free()
    want = np.concatenate([blob[int(o):int(o) + int(n)] for o, n in zip(piece_off, lens)])
    assert np.array_equal(got, want)


def _png_file(i, width, height, bpp):
    """A PNG-shaped container (signature, IHDR, IDAT chunks of seeded sizes, IEND; CRCs as the format has them) around
    one image: -> file bytes, (offset, length) of every IDAT payload, the pixels, the geometry."""
    pixels, filtered, _ = wl.png_filtered(i, width, height, bpp)
    comp = wl.deflate(filtered)
    out = bytearray(b"\x89PNG\r\n\x1a\n")

    def chunk(kind, body):
        at = len(out) + 8
        out.extend(struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body)))
        return at
    chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 6, 0, 0, 0))
    spans = [(chunk(b"IDAT", p), len(p)) for p in wl.idat_pieces(i, comp)]
    chunk(b"IEND", b"")
    return bytes(out), spans, pixels, filtered, (height, width * bpp, bpp)


def test_png_files_uploaded_whole_gather_inflate_unfilter(engine):
    """The README's PNG loop (reference README.md:120-155) with everything after the upload on the device: the files
    go up as they are, the IDAT payloads are gathered into one arena, decoded in one batch and un-filtered."""
    assert engine.lib.infl_gpu_init(0) == 0
    oracle = oracle_lib.load()
    blob, po, pl, pd, so, caps, geom, want_px, want_f = bytearray(), [], [], [], [0], [], [], [], []
    for i in range(96):
        w, h, bpp = [(128, 128, 4), (33, 17, 3), (200, 5, 1), (64, 64, 8)][i % 4]
        f, spans, pixels, filtered, g = _png_file(900 + i, w, h, bpp)
        at = so[-1]
        for off, ln in spans:
            po.append(len(blob) + off); pl.append(ln); pd.append(at); at += ln
        blob += f
        so.append(at); caps.append(len(filtered)); geom.append(g); want_px.append(pixels); want_f.append(filtered)
    n = len(caps)
    assert len(po) > 2 * n                                              # several pieces per image
    do = np.concatenate([[0], np.cumsum(caps)]).astype(np.uint64)
    b = engine.device_batch(np.zeros(so[-1], dtype=np.uint8), np.array(so, dtype=np.uint64), do)
    try:
        engine.gather_into(b, blob, po, pl, pd)
        b.run(flags=1)
        dst, status, outlen, _ = b.results()
        out, st = engine.png_unfilter(b, geom)
    finally:
        b.free()
    assert (status == 0).all() and (outlen == np.array(caps)).all()
    for i in range(n):
        a = int(do[i])
        assert dst[a:a + caps[i]].tobytes() == want_f[i], i
        assert oracle.png_unfilter(want_f[i], *geom[i]) == (0, want_px[i])
        assert st[i] == 0 and out[i] == want_px[i], i


def test_checked_calls_refuse_descriptors_outside_their_buffers(engine):
    """infl_gather_dev_checked / infl_png_unfilter_dev_checked: a piece or an image whose descriptor reaches outside the
    buffers whose sizes the caller gave is refused (status -1) and nothing of it is read or written; the others are done.
    (The same descriptors were first run through the kernel source under AddressSanitizer: test_emulated_gather_of_pieces.)"""
    lib = engine.lib
    assert lib.infl_gpu_init(0) == 0
    blob = np.arange(200, dtype=np.uint8)
    po = np.array([3, 190, 1 << 40, 10, 20, (1 << 64) - 4, 100], dtype=np.uint64)
    pl = np.array([20, 11, 4, 30, 17, 8, 0], dtype=np.uint32)
    pd = np.array([1, 21, 30, 40, 48, 0, 64], dtype=np.uint64)
    n = len(po)
    bufs = [lib.infl_gpu_malloc(4096) for _ in range(6)]          # blob, off, len, dst, arena, piece status
    try:
        assert all(bufs)
        arena = np.full(64, 0xEE, dtype=np.uint8)
        for b, a in zip(bufs[:5], (blob, po, pl, pd, arena)):
            assert lib.infl_gpu_upload(b, a.ctypes.data, a.nbytes) == 0
        assert lib.infl_gather_dev_checked(bufs[0], blob.nbytes, bufs[1], bufs[2], bufs[3], n, bufs[4], 64, bufs[5], None) == 0
        assert lib.infl_gpu_sync() == 0
        st = np.zeros(n, dtype=np.int32)
        assert lib.infl_gpu_download(st.ctypes.data, bufs[5], st.nbytes) == 0
        assert lib.infl_gpu_download(arena.ctypes.data, bufs[4], 64) == 0
        assert st.tolist() == [0, -1, -1, -1, -1, -1, 0]          # (an empty piece at the very end of both buffers is fine)
        want = np.full(64, 0xEE, dtype=np.uint8)
        want[1:21] = blob[3:23]
        assert np.array_equal(arena, want)
        # without a status array the refused pieces are skipped all the same
        assert lib.infl_gather_dev_checked(bufs[0], blob.nbytes, bufs[1], bufs[2], bufs[3], n, bufs[4], 64, None, None) == 0
        assert lib.infl_gpu_sync() == 0
        assert lib.infl_gpu_download(arena.ctypes.data, bufs[4], 64) == 0 and np.array_equal(arena, want)
    finally:
        for b in bufs:
            if b:
                lib.infl_gpu_free(b)

    # un-filter: image 1's input range and image 2's output range reach beyond the arenas
    pixels, filtered, _ = wl.png_filtered(5, 16, 8, 4)
    g = np.zeros((3, 4), dtype=np.uint32)
    g[:, :3] = (8, 64, 4)
    fl = len(filtered)
    in_off = np.array([0, fl, 2 * fl + 5, 3 * fl + 5], dtype=np.uint64)      # image 1's range ends 5 bytes past its data ...
    in_bytes = 3 * fl                                                       # ... and image 2's past the arena
    out_off = np.array([0, 512, 1024, 1536], dtype=np.uint64)
    out_bytes = 1500                                                        # image 2's output does not fit
    data = np.frombuffer(filtered * 3, dtype=np.uint8)
    bufs = [lib.infl_gpu_malloc(8192) for _ in range(6)]                    # in, in_off, out, out_off, geom, status
    try:
        assert all(bufs)
        for b, a in ((bufs[0], data), (bufs[1], in_off), (bufs[3], out_off), (bufs[4], g)):
            assert lib.infl_gpu_upload(b, a.ctypes.data, a.nbytes) == 0
        assert lib.infl_png_unfilter_dev_checked(bufs[0], in_bytes, bufs[1], None, None, bufs[2], out_bytes, bufs[3],
                                                 bufs[4], 3, bufs[5], None) == 0
        assert lib.infl_gpu_sync() == 0
        st = np.zeros(3, dtype=np.int32)
        out = np.zeros(512, dtype=np.uint8)
        assert lib.infl_gpu_download(st.ctypes.data, bufs[5], 12) == 0
        assert lib.infl_gpu_download(out.ctypes.data, bufs[2], 512) == 0
        assert st.tolist() == [0, 0, -1] and out.tobytes() == pixels
        assert lib.infl_png_unfilter_dev_checked(bufs[0], 2 * fl + 4, bufs[1], None, None, bufs[2], 4096, bufs[3],
                                                 bufs[4], 3, bufs[5], None) == 0
        assert lib.infl_gpu_sync() == 0
        assert lib.infl_gpu_download(st.ctypes.data, bufs[5], 12) == 0
        assert st.tolist() == [0, -1, -1]
    finally:
        for b in bufs:
            if b:
                lib.infl_gpu_free(b)
