"""The reference's own named test lists, with its own feed sizes, through the drop-in API on the GPU
(test/test_files.c:1037-1048 names; :260-331 chunked flow; :688-790 streaming flow), every golden pair
through infl_stream() in pieces of 1, 7, 64 and 4096 bytes, and infl_include() / infl_stream() mixed on
one stream object (the reference's infl_stream is infl_include + decode over the one chunk list,
src/infl/stream.c:481-507)."""
import ctypes as C
import os

import pytest

from defl_amd import capi

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "defl_testdata")
PAIRS = sorted(f for f in os.listdir(os.path.join(DATA, "compressed")) if os.path.exists(os.path.join(DATA, "raw", f)))

# test_files.c:1037-1041
CHUNKED = ["hello", "hello_world", "json", "html", "text_repeated", "zeros_1k", "repeated_a_258", "ascii",
           "huffman_single_a", "distance_test_1", "length_test_3", "bit_align_7"]
# test_files.c:1043-1048 (xml, binary and zlib_1 do not exist in the reference's data directory either:
# its main() skips names it does not find, :1106-1118)
STREAMING = ["hello", "hello_world", "json", "xml", "binary", "zeros_1k", "huffman_single_a", "multi_block_1",
             "dynamic_huffman_1", "distance_test_1", "length_test_3", "bit_align_7", "zlib_1"]


def _read(kind, name):
    with open(os.path.join(DATA, kind, name), "rb") as f:
        return f.read()


def test_the_lists_name_what_the_reference_runs():
    assert all(n in PAIRS for n in CHUNKED)
    assert sorted(n for n in STREAMING if n not in PAIRS) == ["binary", "xml", "zlib_1"]
    assert len([n for n in STREAMING if n in PAIRS]) == 10


@pytest.mark.parametrize("name", CHUNKED)
def test_reference_chunked_list(engine, name):
    """test_file_chunked (test_files.c:260-331): pieces of 1 << (pos % 4) bytes, then one infl()."""
    lib = engine.lib
    raw, comp = _read("raw", name), _read("compressed", name)
    out = (C.c_uint8 * (len(raw) + 1000))()
    src = (C.c_uint8 * max(1, len(comp))).from_buffer_copy(comp or b"\0")
    st = lib.infl_init(out, len(raw) + 1000, 0)
    pos = 0
    while pos < len(comp):
        n = min(1 << (pos % 4), len(comp) - pos)
        lib.infl_include(st, C.byref(src, pos), n)
        pos += n
    assert lib.infl(st) == capi.UNZ_OK
    assert bytes(out[:len(raw)]) == raw
    lib.infl_destroy(st)


@pytest.mark.parametrize("name", [n for n in STREAMING if n in PAIRS])
def test_reference_streaming_list(engine, name):
    """test_file_streaming (test_files.c:688-790): feeds of 64 << (count % 5) bytes, at most five empty calls
    behind the last one."""
    lib = engine.lib
    raw, comp = _read("raw", name), _read("compressed", name)
    out = (C.c_uint8 * (len(raw) + 1000))()
    src = (C.c_uint8 * max(1, len(comp))).from_buffer_copy(comp or b"\0")
    st = lib.infl_init(out, len(raw) + 1000, 0)
    pos, count, rc = 0, 0, capi.UNZ_UNFINISHED
    while pos < len(comp):
        n = min(64 << (count % 5), len(comp) - pos)
        rc = lib.infl_stream(st, C.byref(src, pos), n)
        pos += n
        count += 1
        if rc != capi.UNZ_UNFINISHED:
            break
    tries = 0
    while rc == capi.UNZ_UNFINISHED and pos >= len(comp) and tries < 5:
        rc = lib.infl_stream(st, None, 0)
        tries += 1
    assert rc == capi.UNZ_OK, (name, rc)
    assert bytes(out[:len(raw)]) == raw
    lib.infl_destroy(st)


@pytest.mark.parametrize("step", [1, 7, 64, 4096])
def test_every_golden_pair_streamed_in_pieces(engine, step):
    """Every raw/compressed pair fed to infl_stream() in pieces of `step` bytes gives the one-shot bytes
    (byte-wise feeding is kept to the pairs of at most 40 000 compressed bytes: a call is a kernel launch)."""
    lib = engine.lib
    st = None
    done = 0
    for name in PAIRS:
        raw, comp = _read("raw", name), _read("compressed", name)
        if step == 1 and len(comp) > 40000:
            continue
        out = (C.c_uint8 * (len(raw) + 16))()
        if st is None:
            st = lib.infl_init(out, len(raw) + 16, 0)
        else:
            lib.infl_reset(st, out, len(raw) + 16, 0)
        src = (C.c_uint8 * max(1, len(comp))).from_buffer_copy(comp or b"\0")
        rc, pos = capi.UNZ_UNFINISHED, 0
        if not comp:
            rc = lib.infl_stream(st, None, 0)                   # "empty data": stream.c:489-493
        while pos < len(comp) and rc == capi.UNZ_UNFINISHED:
            n = min(step, len(comp) - pos)
            rc = lib.infl_stream(st, C.byref(src, pos), n)
            pos += n
        assert rc == capi.UNZ_OK, (name, step, rc, pos, len(comp))
        assert lib.infl_output_pos(st) == len(raw) and bytes(out[:len(raw)]) == raw, (name, step)
        done += 1
    lib.infl_destroy(st)
    assert done >= (130 if step == 1 else 139)


def test_include_and_stream_mixed_on_one_object(engine):
    """One input record behind both calls: bytes given with infl_include() between infl_stream() calls are decoded
    by the next infl_stream(); data included before any decode makes infl_stream(NULL, 0) a no-op (UNZ_NOOP,
    stream.c:505-507); a one-shot infl() on the object afterwards sees every byte that was streamed."""
    lib = engine.lib
    raw, comp = _read("raw", "png_simulation"), _read("compressed", "png_simulation")
    src = (C.c_uint8 * len(comp)).from_buffer_copy(comp)
    out = (C.c_uint8 * (len(raw) + 16))()
    st = lib.infl_init(out, len(raw) + 16, 0)
    third = len(comp) // 3
    assert third > 8192                                                   # (a piece of this size is borrowed, not copied)
    assert lib.infl_stream(st, None, 0) == capi.UNZ_OK                    # nothing there at all: "empty data"
    lib.infl_include(st, src, 100)
    assert lib.infl_stream(st, None, 0) == capi.UNZ_NOOP                  # data, but no decode has begun
    assert lib.infl_output_pos(st) == 0
    assert lib.infl_stream(st, C.byref(src, 100), third - 100) == capi.UNZ_UNFINISHED
    got = lib.infl_output_pos(st)
    assert 0 < got < len(raw) and bytes(out[:got]) == raw[:got]
    lib.infl_include(st, C.byref(src, third), third)                      # a large piece (borrowed) ...
    lib.infl_include(st, C.byref(src, 2 * third), 5)                      # ... and a small one (copied)
    assert lib.infl_stream(st, None, 0) == capi.UNZ_UNFINISHED            # both are seen by the next call
    got2 = lib.infl_output_pos(st)
    assert got2 > got and bytes(out[:got2]) == raw[:got2]
    assert lib.infl_stream(st, C.byref(src, 2 * third + 5), len(comp) - 2 * third - 5) == capi.UNZ_OK
    assert lib.infl_output_pos(st) == len(raw) and bytes(out[:len(raw)]) == raw
    assert lib.infl_stream(st, None, 0) == capi.UNZ_OK                    # done stays done
    # the one-shot call on the same object decodes the same input from its start
    C.memset(out, 0, len(raw))
    assert lib.infl(st) == capi.UNZ_OK
    assert lib.infl_output_pos(st) == len(raw) and bytes(out[:len(raw)]) == raw
    # infl_reset_pool() starts the input over
    lib.infl_reset(st, out, len(raw) + 16, 0)
    assert lib.infl_stream(st, None, 0) == capi.UNZ_OK
    assert lib.infl_stream(st, src, len(comp)) == capi.UNZ_OK and bytes(out[:len(raw)]) == raw
    lib.infl_destroy(st)
