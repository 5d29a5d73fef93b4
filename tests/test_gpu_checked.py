"""Opt-in integrity and framing on the GPU (infl_batch_checked / infl_batch_dev_checked): Adler-32 of
zlib streams, gzip members with CRC-32 and ISIZE.  The reference holds no code or fixture for either
(it never reads what follows the final block), so the checker here is Python's zlib / gzip."""
import gzip
import random
import zlib

import numpy as np
import pytest

import workloads as wl
from defl_amd import capi
from tests.test_kernel_emulation import _gzip_member

pytestmark = pytest.mark.gpu


def test_adler32_good_corrupted_truncated(engine):
    assert engine.lib.infl_gpu_init(0) == 0
    rng = random.Random(7)
    raws = [wl.text_stream(100 + i, 1 + 977 * i) for i in range(40)] + [wl.text_stream(7, 65536), bytes(200000),
            bytes(rng.randrange(256) for _ in range(70001)), b""]
    good = [wl.deflate(r, 6 if i % 3 else 1) for i, r in enumerate(raws)]
    streams, caps, expect = [], [], []
    for c, r in zip(good, raws):
        streams.append(c); caps.append(len(r)); expect.append((0, capi.D_OK))
    for c, r in list(zip(good, raws))[5:15]:
        b = bytearray(c); b[-1 - rng.randrange(4)] ^= 1 << rng.randrange(8)
        streams.append(bytes(b)); caps.append(len(r)); expect.append((-1, capi.D_CHECKSUM))
    for c, r in list(zip(good, raws))[20:26]:
        streams.append(c[:-1 - rng.randrange(4)]); caps.append(len(r)); expect.append((-1, capi.D_NEED_INPUT))
    streams.append(good[40][: len(good[40]) // 2]); caps.append(65536); expect.append((-1, capi.D_NEED_INPUT))
    streams.append(good[40]); caps.append(60000); expect.append((-1, capi.D_FULL))
    src, so, do = wl.pack(streams, caps)
    dst, status, outlen, detail = engine.batch_checked(src, so, do, capi.CHECK_ADLER32)
    for i, (st, dt) in enumerate(expect):
        assert (status[i], detail[i]) == (st, dt), (i, int(status[i]), int(detail[i]))
    for i, r in enumerate(raws):
        assert outlen[i] == len(r) and dst[int(do[i]):int(do[i]) + len(r)].tobytes() == r
    # the plain calls still ignore the trailer, like the reference
    _, st_plain, _ = engine.batch_host(src, so, do, 1)
    k = len(raws)
    assert (st_plain[k:k + 10] == 0).all()


def test_gzip_members(engine):
    assert engine.lib.infl_gpu_init(0) == 0
    rng = random.Random(8)
    raws = [wl.text_stream(300 + i, 50 + 1500 * i) for i in range(30)] + [b"", bytes(100000)]
    members = []
    for i, r in enumerate(raws):
        if i % 4 == 0:
            members.append(gzip.compress(r, 6))                       # Python's own writer
        else:
            members.append(_gzip_member(r, name=b"n%d" % i if i % 2 else b"", comment=b"c" * (i % 5), extra=b"e" * (i % 7),
                                        hcrc=i % 3 == 0, level=1 + i % 9))
    streams, caps, expect = list(members), [len(r) for r in raws], [(0, capi.D_OK)] * len(raws)
    for j in (3, 8, 12):
        b = bytearray(members[j]); b[-5 - rng.randrange(4)] ^= 0x10           # CRC-32
        streams.append(bytes(b)); caps.append(len(raws[j])); expect.append((-1, capi.D_CHECKSUM))
        b = bytearray(members[j]); b[-1 - rng.randrange(4)] ^= 0x01           # ISIZE
        streams.append(bytes(b)); caps.append(len(raws[j])); expect.append((-1, capi.D_CHECKSUM))
        streams.append(members[j][:-1 - rng.randrange(8)]); caps.append(len(raws[j])); expect.append((-1, capi.D_NEED_INPUT))
    streams.append(members[5][:9]); caps.append(100); expect.append((-1, capi.D_NEED_INPUT))       # inside the header
    streams.append(b"\x1f\x8c" + members[5][2:]); caps.append(len(raws[5])); expect.append((-1, capi.D_INVALID))
    streams.append(wl.deflate(raws[5])); caps.append(len(raws[5])); expect.append((-1, capi.D_INVALID))   # a zlib stream
    src, so, do = wl.pack(streams, caps)
    dst, status, outlen, detail = engine.batch_checked(src, so, do, capi.CHECK_GZIP)
    for i, (st, dt) in enumerate(expect):
        assert (status[i], detail[i]) == (st, dt), (i, int(status[i]), int(detail[i]))
    for i, r in enumerate(raws):
        assert outlen[i] == len(r) and dst[int(do[i]):int(do[i]) + len(r)].tobytes() == r


def test_checked_device_batch_of_64k_streams(engine):
    """Device-resident arenas through infl_batch_dev_checked: 2048 zlib streams of 64 KiB, one in 16 with a
    damaged trailer; verdicts against zlib.adler32 of what the oracle-checked decode produced."""
    lib = engine.lib
    assert lib.infl_gpu_init(0) == 0
    n = 2048
    raws = [wl.text_stream(500 + i % 64, 65536) for i in range(n)]
    base = {i: wl.deflate(raws[i]) for i in range(64)}
    comps = []
    for i in range(n):
        c = base[i % 64]
        if i % 16 == 5:
            b = bytearray(c); b[-1] ^= 0x80; c = bytes(b)
        comps.append(c)
    src, so, do = wl.pack(comps, [65536] * n)
    b = engine.device_batch(src, so, do)
    work = lib.infl_gpu_malloc(8 * n)
    assert work
    assert lib.infl_batch_dev_checked(b.d_src, b.d_soff, b.d_dst, b.d_doff, capi.CHECK_ADLER32, n, b.d_status, b.d_outlen,
                                      b.d_detail, work, None) == 0
    dst, status, outlen, detail = b.results()
    lib.infl_gpu_free(work)
    b.free()
    want_bad = np.array([i % 16 == 5 for i in range(n)])
    assert ((status != 0) == want_bad).all() and (detail[want_bad] == capi.D_CHECKSUM).all() and (outlen == 65536).all()
    for i in (0, 5, 777, n - 1):
        assert dst[i * 65536:(i + 1) * 65536].tobytes() == raws[i]
        assert zlib.adler32(raws[i]) == int.from_bytes(base[i % 64][-4:], "big")
