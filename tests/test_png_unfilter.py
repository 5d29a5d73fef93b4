"""The CPU checker for the PNG un-filter kernel (oracle/png_unfilter.c).  The reference holds no code and no
fixture for this step (README.md:150-152 only names it), so parity is unpinned by the reference: the checker is
pinned here against a second, independent restatement of the PNG specification's section 9 (plain Python,
written from the per-byte recurrence) and against round trips through the encoder-side filter."""
import numpy as np
import pytest

import workloads as wl
from tests import oracle_lib


def py_unfilter(data: bytes, rows: int, rowbytes: int, bpp: int) -> bytes:
    out = bytearray(rows * rowbytes)
    for r in range(rows):
        ft = data[r * (rowbytes + 1)]
        for x in range(rowbytes):
            left = out[r * rowbytes + x - bpp] if x >= bpp else 0
            above = out[(r - 1) * rowbytes + x] if r else 0
            corner = out[(r - 1) * rowbytes + x - bpp] if r and x >= bpp else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = left
            elif ft == 2:
                pred = above
            elif ft == 3:
                pred = (left + above) // 2
            else:
                est = left + above - corner
                dl, da, dc = abs(est - left), abs(est - above), abs(est - corner)
                pred = left if dl <= da and dl <= dc else above if da <= dc else corner
            out[r * rowbytes + x] = (data[r * (rowbytes + 1) + 1 + x] + pred) & 0xFF
    return bytes(out)


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


@pytest.mark.parametrize("bpp", [1, 2, 3, 4, 6, 8])
def test_checker_matches_python_restatement_on_random_bytes(oracle, bpp):
    rng = np.random.default_rng(bpp)
    for rows, width in ((1, 1), (2, 5), (7, 3), (70, 9), (5, 40)):
        rowbytes = width * bpp
        data = rng.integers(0, 256, size=(rows, rowbytes + 1), dtype=np.uint8)
        data[:, 0] = rng.integers(0, 5, size=rows)
        rc, got = oracle.png_unfilter(data.tobytes(), rows, rowbytes, bpp)
        assert rc == 0 and got == py_unfilter(data.tobytes(), rows, rowbytes, bpp)


def test_checker_inverts_the_encoder_filter(oracle):
    for i, (w, h, bpp) in enumerate(((128, 128, 4), (33, 70, 3), (200, 3, 1), (17, 129, 8), (64, 64, 2), (5, 5, 6))):
        pixels, filtered, types = wl.png_filtered(i, w, h, bpp)
        assert set(types) <= {0, 1, 2, 3, 4}
        rc, got = oracle.png_unfilter(filtered, h, w * bpp, bpp)
        assert rc == 0 and got == pixels
    for t in range(5):                                         # one type for the whole image
        pixels, filtered, _ = wl.png_filtered(9, 40, 30, 4, types=[t] * 30)
        assert oracle.png_unfilter(filtered, 30, 160, 4) == (0, pixels)


def test_checker_edges(oracle):
    assert oracle.png_unfilter(b"", 0, 16, 4) == (0, b"")
    assert oracle.png_unfilter(bytes([4, 3, 2]), 3, 0, 1) == (0, b"")
    assert oracle.png_unfilter(bytes([5, 1, 2, 3]), 1, 3, 1)[0] == -1
    # Paeth picks the byte above where it is the closest to left + above - corner
    assert oracle.png_unfilter(bytes([0, 10, 20, 4, 0, 0]), 2, 2, 1)[1] == bytes([10, 20, 10, 20])
