"""infl_batch_noalloc(): the allocation-free variant of infl_batch() (its device staging is a fixed
array inside the library) gives the results of infl_batch(), call after call on the same array."""
import numpy as np
import pytest

import workloads as wl
from defl_amd import capi

pytestmark = pytest.mark.gpu


def _noalloc(eng, src, src_off, dst_off, flags):
    n = len(src_off) - 1
    src = np.ascontiguousarray(src, dtype=np.uint8)
    src_off = np.ascontiguousarray(src_off, dtype=np.uint64)
    dst_off = np.ascontiguousarray(dst_off, dtype=np.uint64)
    dst = np.full(max(1, int(dst_off[-1])), 0xA5, dtype=np.uint8)
    status = np.full(max(1, n), -99, dtype=np.int32)
    outlen = np.full(max(1, n), 12345, dtype=np.uint32)
    rc = eng.lib.infl_batch_noalloc(src.ctypes.data, src_off.ctypes.data, dst.ctypes.data, dst_off.ctypes.data,
                                    flags, n, status.ctypes.data, outlen.ctypes.data)
    return rc, dst, status[:n], outlen[:n]


def _same(a, b, dst_off):
    (dst_a, st_a, ol_a), (dst_b, st_b, ol_b) = a, b
    assert np.array_equal(st_a, st_b) and np.array_equal(ol_a, ol_b)
    for i in range(len(st_a)):
        lo, n = int(dst_off[i]), int(ol_a[i])
        assert np.array_equal(dst_a[lo:lo + n], dst_b[lo:lo + n]), i


def test_noalloc_matches_infl_batch_twice_in_a_row():
    eng = capi.load()
    assert eng.lib.infl_gpu_init(0) == 0
    # first call: 200 streams, text, stored, truncated and corrupted ones among them
    raws = [wl.text_stream(i, 3000 + 97 * i) for i in range(180)] + [wl.stored_stream(i, 40000)[0] for i in range(20)]
    comps = [wl.deflate(r) for r in raws[:180]] + [wl.stored_stream(i, 40000)[1] for i in range(20)]
    comps[5] = comps[5][: len(comps[5]) // 2]
    bad = bytearray(comps[9]); bad[40] ^= 0x10; comps[9] = bytes(bad)
    caps = [len(r) for r in raws]
    caps[11] = caps[11] - 7                         # output space runs out
    src, src_off, dst_off = wl.pack(comps, caps)
    want = eng.batch_host(src, src_off, dst_off, 1)
    rc, dst, status, outlen = _noalloc(eng, src, src_off, dst_off, 1)
    assert rc == 0
    _same(want, (dst, status, outlen), dst_off)
    assert (status == 0).sum() >= 195 and (status != 0).sum() >= 2
    tails1 = [dst[int(dst_off[i]) + int(outlen[i]):int(dst_off[i + 1])] for i in range(len(caps))]
    assert all((t == 0).all() for t in tails1) and sum(t.size for t in tails1) > 1000   # failed streams: all of their share

    # second call right behind it, smaller, other data: nothing of the first call shows through
    raws2 = [wl.text_stream(1000 + i, 900 + 11 * i) for i in range(37)]
    comps2 = [wl.deflate(r, 6, False) for r in raws2]
    src2, so2, do2 = wl.pack(comps2, [len(r) for r in raws2])
    want2 = eng.batch_host(src2, so2, do2, 0)
    rc2, dst2, st2, ol2 = _noalloc(eng, src2, so2, do2, 0)
    assert rc2 == 0 and (st2 == 0).all()
    _same(want2, (dst2, st2, ol2), do2)
    assert dst2[: int(do2[-1])].tobytes() == b"".join(raws2)

    # third call: streams that fail where the first call's plain text lay: their share comes back as zeros
    comps3 = [c[: len(c) // 3] for c in comps2]
    src3, so3, do3 = wl.pack(comps3, [len(r) + 100 for r in raws2])
    rc3, dst3, st3, ol3 = _noalloc(eng, src3, so3, do3, 0)
    assert rc3 == 0 and (st3 != 0).all() and (ol3 == 0).all()
    assert (dst3[: int(do3[-1])] == 0).all()


def test_noalloc_refuses_offsets_that_decrease():
    eng = capi.load()
    assert eng.lib.infl_gpu_init(0) == 0
    comp = wl.deflate(b"abc" * 10)
    src, so, do = wl.pack([comp] * 4, [64] * 4)
    bad = so.copy()
    bad[2] = bad[1] - 1
    rc, _, status, _ = _noalloc(eng, src, bad, do, 1)
    assert rc == capi.UNZ_ERR and (status == -99).all()
    bad = do.copy()
    bad[3], bad[2] = bad[2], bad[3]
    rc, _, status, _ = _noalloc(eng, src, so, bad, 1)
    assert rc == capi.UNZ_ERR and (status == -99).all()
    rc, _, status, outlen = _noalloc(eng, src, so, do, 1)
    assert rc == 0 and (status == 0).all() and (outlen == 30).all()


def test_noalloc_rejects_what_exceeds_the_fixed_workspace():
    eng = capi.load()
    assert eng.lib.infl_gpu_init(0) == 0
    comp = wl.deflate(b"abc" * 10)
    # too many streams
    n = 4097
    src, so, do = wl.pack([comp] * n, [30] * n)
    rc, _, status, _ = _noalloc(eng, src, so, do, 1)
    assert rc == capi.UNZ_EFULL and (status == -99).all()
    # too much output capacity
    src, so, do = wl.pack([comp] * 3, [6 << 20] * 3)
    rc, _, status, _ = _noalloc(eng, src, so, do, 1)
    assert rc == capi.UNZ_EFULL and (status == -99).all()
    # exactly at the limits is fine
    src, so, do = wl.pack([comp] * 4, [4 << 20] * 4)
    rc, dst, status, outlen = _noalloc(eng, src, so, do, 1)
    assert rc == 0 and (status == 0).all() and (outlen == 30).all()
