"""CPU-side checks of the C-ABI library: it loads, exports every symbol the public headers
declare, and the host logic that needs no device behaves like the reference's."""
import ctypes as C
import os
import re

from defl_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", "defl", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(infl[a-z_]*)\s*\(", text)) - {"infl_buf"}


def test_library_exports_every_declared_symbol(engine):
    declared = _declared("infl.h") | _declared("infl_batch.h")
    assert declared == set(capi.EXPORTS)
    for name in declared:
        assert hasattr(engine.lib, name), name


def test_infl_without_input_is_noop(engine):
    """Nothing included -> UNZ_NOOP (reference src/infl/infl.c:1544-1545), no device needed."""
    buf = (C.c_uint8 * 16)()
    st = engine.lib.infl_init(buf, 16, 0)
    assert st
    assert engine.lib.infl(st) == capi.UNZ_NOOP
    assert engine.lib.infl_output_pos(st) == 0
    assert engine.lib.infl_input_pos(st) == 0
    engine.lib.infl_destroy(st)
    engine.lib.infl_destroy(None)          # NULL-safe like the reference (mem.c:337)


def test_resize_output_rules(engine):
    """infl_resize_output rejects NULL and a capacity below the output position (mem.c:282-283)."""
    buf = (C.c_uint8 * 16)()
    st = engine.lib.infl_init(buf, 16, 0)
    assert engine.lib.infl_resize_output(st, None, 32) == capi.UNZ_ERR
    assert engine.lib.infl_resize_output(None, buf, 32) == capi.UNZ_ERR
    assert engine.lib.infl_resize_output(st, buf, 8) == capi.UNZ_OK
    engine.lib.infl_destroy(st)


def test_include_accepts_many_pieces_and_reset(engine):
    """Piece bookkeeping: small pieces coalesce, large ones are borrowed, reset reuses pages."""
    buf = (C.c_uint8 * 16)()
    st = engine.lib.infl_init(buf, 16, 0)
    small = (C.c_uint8 * 100)()
    large = (C.c_uint8 * 20000)()
    for _ in range(1000):
        engine.lib.infl_include(st, small, 100)
    engine.lib.infl_include(st, large, 20000)
    engine.lib.infl_include(st, small, 0)
    engine.lib.infl_reset(st, buf, 16, 1)
    assert engine.lib.infl(st) == capi.UNZ_NOOP
    engine.lib.infl_reset_pool(st)
    engine.lib.infl_destroy(st)


def test_stream_with_no_data_is_ok(engine):
    """infl_stream(NULL, 0) on a fresh stream: 'empty data' -> UNZ_OK (stream.c:489-493)."""
    buf = (C.c_uint8 * 16)()
    st = engine.lib.infl_init(buf, 16, 0)
    assert engine.lib.infl_stream(st, None, 0) == capi.UNZ_OK
    engine.lib.infl_destroy(st)


def test_stream_with_data_but_no_decode_yet_is_noop(engine):
    """Data given with infl_include(), then infl_stream(NULL, 0) before any decode has begun: UNZ_NOOP
    (stream.c:505-507); the bytes are still there for a later call.  Returns before any device is needed."""
    buf = (C.c_uint8 * 16)()
    st = engine.lib.infl_init(buf, 16, 0)
    piece = (C.c_uint8 * 6).from_buffer_copy(bytes.fromhex("010100feff41"))
    engine.lib.infl_include(st, piece, 6)
    assert engine.lib.infl_stream(st, None, 0) == capi.UNZ_NOOP
    assert engine.lib.infl_stream(st, None, 0) == capi.UNZ_NOOP
    assert engine.lib.infl_output_pos(st) == 0
    engine.lib.infl_reset_pool(st)
    assert engine.lib.infl_stream(st, None, 0) == capi.UNZ_OK      # nothing there any more: "empty data"
    engine.lib.infl_destroy(st)


def test_page_pool_recycles_pages_across_streams(engine):
    """Pages of destroyed streams go to later streams (the library's page pool): many generations of streams with
    small pieces, each generation destroyed before the next; contents never mix (host bookkeeping only)."""
    buf = (C.c_uint8 * 16)()
    piece = (C.c_uint8 * 5000)()
    for gen in range(6):
        sts = [engine.lib.infl_init(buf, 16, 0) for _ in range(64)]
        for k, st in enumerate(sts):
            for j in range(10):
                C.memset(piece, (gen * 7 + k + j) & 0xff, 5000)
                engine.lib.infl_include(st, piece, 5000)
        for st in sts[::2]:
            engine.lib.infl_destroy(st)
        for st in sts[1::2]:
            engine.lib.infl_reset_pool(st)
            engine.lib.infl_include(st, piece, 100)
            engine.lib.infl_destroy(st)


def test_no_device_fails_loudly_not_silently(engine):
    """Without a GPU the decode entry points report an error; nothing decodes on the CPU."""
    if engine.lib.infl_gpu_device_count() > 0:
        return
    rc, pos, out = engine.infl_buf(bytes.fromhex("010100feff41"), 10, 0)
    assert rc == capi.UNZ_ERR and pos == 0 and out == b""
