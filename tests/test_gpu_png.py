"""PNG un-filter on the GPU (infl_png_unfilter_dev) behind the batch decode, against the CPU checker
(oracle/png_unfilter.c).  The reference holds neither code nor fixtures for this step (README.md:150-152 names
it as the caller's job): parity unpinned by the reference; tests/test_png_unfilter.py pins the checker."""
import numpy as np
import pytest

import workloads as wl
from defl_amd import capi
from tests import oracle_lib

pytestmark = pytest.mark.gpu


def _decode_and_unfilter(engine, filtered, geom, level=6, chained=True):
    comps = [wl.deflate(f, level) for f in filtered]
    src, so, do = wl.pack(comps, [max(1, len(f)) for f in filtered])
    b = engine.device_batch(src, so, do)
    b.run(flags=1)
    _, status, outlen, _ = b.results(want_dst=False)
    out, st = engine.png_unfilter(b, geom, chained=chained)
    b.free()
    return out, st, status, outlen


def test_every_filter_type_every_pixel_size(engine):
    """Images of one to 300 scanlines (more than 64 = several bands of the kernel), rows shorter and longer than
    its 16-byte chunks, every bpp a PNG can have, filter types shuffled per scanline; plus random bytes."""
    assert engine.lib.infl_gpu_init(0) == 0
    oracle = oracle_lib.load()
    rng = np.random.default_rng(11)
    filtered, geom, want = [], [], []
    shapes = [(128, 128, 4), (1, 1, 1), (3, 1, 8), (5, 2, 3), (15, 65, 1), (16, 64, 1), (17, 63, 2), (300, 7, 3),
              (7, 300, 4), (129, 129, 6), (64, 128, 8), (31, 200, 2), (1000, 130, 4), (2, 64, 4), (4, 65, 4)]
    for i, (w, h, bpp) in enumerate(shapes):
        pixels, f, _ = wl.png_filtered(i, w, h, bpp)
        filtered.append(f); geom.append((h, w * bpp, bpp)); want.append(pixels)
    for t in range(5):                                         # one filter type for a whole image
        pixels, f, _ = wl.png_filtered(50 + t, 100, 70, 4, types=[t] * 70)
        filtered.append(f); geom.append((70, 400, 4)); want.append(pixels)
    for bpp in (1, 2, 3, 4, 5, 6, 7, 8):                       # random bytes: no picture behind them
        rows, rowbytes = 67 + bpp, 11 * bpp
        d = rng.integers(0, 256, size=(rows, rowbytes + 1), dtype=np.uint8)
        d[:, 0] = rng.integers(0, 5, size=rows)
        rc, w_ = oracle.png_unfilter(d.tobytes(), rows, rowbytes, bpp)
        assert rc == 0
        filtered.append(d.tobytes()); geom.append((rows, rowbytes, bpp)); want.append(w_)
    filtered.append(b""); geom.append((0, 64, 4)); want.append(b"")                 # no scanlines
    filtered.append(bytes([0, 1, 2, 3, 4])); geom.append((5, 0, 1)); want.append(b"")   # empty scanlines
    for f, g, w_ in zip(filtered, geom, want):
        assert oracle.png_unfilter(f, *g) == (0, w_)
    out, st, status, _ = _decode_and_unfilter(engine, filtered, geom)
    assert (status == 0).all()
    for i, w_ in enumerate(want):
        assert st[i] == 0, (i, geom[i])
        assert out[i] == w_, (i, geom[i])
    out, st, _, _ = _decode_and_unfilter(engine, filtered, geom, level=0, chained=False)
    assert (st == 0).all() and out == want


def test_failures_are_reported_per_image(engine):
    assert engine.lib.infl_gpu_init(0) == 0
    pixels, good, _ = wl.png_filtered(1, 32, 40, 4)
    bad_type = bytearray(good); bad_type[17 * 129] = 5
    late_bad = bytearray(wl.png_filtered(2, 32, 100, 4)[1]); late_bad[99 * 129] = 200
    filtered = [good, bytes(bad_type), good, good, bytes(late_bad), good, good]
    geom = [(40, 128, 4), (40, 128, 4), (41, 128, 4), (40, 128, 9), (100, 128, 4), (40, 128, 0), (40, 128, 4)]
    comps = [wl.deflate(f) for f in filtered]
    comps[6] = comps[6][: len(comps[6]) // 2]                 # a stream that ends early
    src, so, do = wl.pack(comps, [len(f) for f in filtered])
    b = engine.device_batch(src, so, do)
    b.run(flags=1)
    _, status, _, _ = b.results(want_dst=False)
    assert list(status != 0) == [False] * 6 + [True]
    out, st = engine.png_unfilter(b, geom)
    b.free()
    assert list(st) == [0, -1, -1, -1, -1, -1, -1]
    assert out[0] == pixels


def test_png_config_decode_then_unfilter(engine):
    """The PNG-shaped configuration: 2048 zlib streams of 128 x 128 RGBA scanlines, decoded and un-filtered on
    the device; every image against the pixels it was made from (and a sample against the checker)."""
    assert engine.lib.infl_gpu_init(0) == 0
    oracle = oracle_lib.load()
    n, distinct = 2048, 64
    base = [wl.png_filtered(1000 + i) for i in range(distinct)]
    comp = [wl.deflate(f) for _, f, _ in base]
    for i in (0, 31, 63):
        assert oracle.png_unfilter(base[i][1], 128, 512, 4) == (0, base[i][0])
    src, so, do = wl.pack([comp[i % distinct] for i in range(n)], [128 * 513] * n)
    b = engine.device_batch(src, so, do)
    b.run(flags=1)
    _, status, outlen, _ = b.results(want_dst=False)
    assert (status == 0).all() and (outlen == 128 * 513).all()
    out, st = engine.png_unfilter(b, [(128, 512, 4)] * n)
    b.free()
    assert (st == 0).all()
    for i in range(n):
        assert out[i] == base[i % distinct][0], i
