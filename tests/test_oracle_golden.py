"""Pins the CPU oracle against every fixture the reference's own tests hold for the
one-shot path (test/test_files.c:174-257 file pairs, :460-544 error vectors, :618-686
regression vectors) and against the system zlib on valid streams."""
import json
import os
import random
import zlib

import pytest

import workloads as wl
from tests import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "defl_testdata")
PAIRS = sorted(f for f in os.listdir(os.path.join(DATA, "compressed"))
               if os.path.exists(os.path.join(DATA, "raw", f)))
EDGE = sorted(f for f in os.listdir(os.path.join(DATA, "compressed"))
              if not os.path.exists(os.path.join(DATA, "raw", f)))


def _read(kind, name):
    with open(os.path.join(DATA, kind, name), "rb") as f:
        return f.read()


def test_fixture_census():
    assert len(PAIRS) == 139 and len(EDGE) == 5


@pytest.mark.parametrize("name", PAIRS)
def test_file_pair(oracle, name):
    """infl_init(out, size + 1000, 0) + one infl_include + infl (test_files.c:213-235)."""
    raw, comp = _read("raw", name), _read("compressed", name)
    rc, pos, out, _ = oracle.infl(comp, len(raw) + 1000, 0)
    assert rc == ol.OK
    assert pos == len(raw) and out == raw


@pytest.mark.parametrize("name", PAIRS)
def test_file_pair_vs_zlib(oracle, name):
    comp = _read("compressed", name)
    if not comp:
        return
    assert zlib.decompress(comp, wbits=-15) == _read("raw", name)


@pytest.mark.parametrize("name", EDGE)
def test_edge_streams_agree_with_zlib(oracle, name):
    """The five hand-built edge_* streams have no raw twin in the reference; where zlib
    accepts one, the oracle must produce the same bytes."""
    comp = _read("compressed", name)
    try:
        want = zlib.decompress(comp, wbits=-15)
    except zlib.error:
        want = None
    rc, pos, out, _ = oracle.infl(comp, 70000, 0)
    if want is not None and rc == ol.OK:
        assert out == want


def test_kats(oracle):
    with open(os.path.join(HERE, "golden", "kats.json")) as f:
        kats = json.load(f)["vectors"]
    assert len(kats) == 8
    for k in kats:
        rc, pos, out, _ = oracle.infl(bytes.fromhex(k["src"]), k["dstlen"], k["flags"])
        if k["expect"] == "err":
            assert rc != ol.OK, k["name"]
        else:
            assert rc == ol.OK and out == bytes.fromhex(k["out"]), k["name"]


def test_empty_input_is_ok(oracle):
    assert oracle.infl(b"", 16, 0)[:2] == (ol.OK, 0)
    assert oracle.infl(b"", 16, 1)[:2] == (ol.OK, 0)


def _corpus(rng, n):
    words = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(2, 10)))
             for _ in range(200)]
    out = bytearray()
    while len(out) < n:
        out += rng.choice(words) + rng.choice([b" ", b" ", b", ", b".\n"])
    return bytes(out[:n])


@pytest.mark.parametrize("level", [0, 1, 6, 9])
@pytest.mark.parametrize("strategy", [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE])
def test_zlib_roundtrip(oracle, level, strategy):
    rng = random.Random(1234 + level * 10 + strategy)
    for size in (0, 1, 2, 100, 4097, 70000, 300000):
        for kind in ("text", "random", "zeros", "mixed"):
            if kind == "text":
                raw = _corpus(rng, size)
            elif kind == "random":
                raw = rng.randbytes(size)
            elif kind == "zeros":
                raw = bytes(size)
            else:
                raw = (_corpus(rng, size // 2) + rng.randbytes(size - size // 2))
            for wbits, flags in ((-15, 0), (15, 1), (-9, 0)):
                c = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
                comp = c.compress(raw) + c.flush()
                rc, pos, out, _ = oracle.infl(comp, len(raw), flags)
                assert rc == ol.OK and out == raw, (size, kind, wbits)
                if size:
                    # one byte short of capacity: output-full is flattened to ERR (infl.c:1078-1087)
                    rc, pos, _, detail = oracle.infl(comp, len(raw) - 1, flags)
                    assert rc == ol.ERR and detail == ol.D_FULL


def test_truncation_never_ok_and_mutations_match_zlib(oracle):
    """Differential fuzz: wherever zlib accepts a mutated stream, the oracle (if it also
    accepts) must produce the same bytes; truncating a valid stream inside its last
    block must not be accepted with the full output."""
    rng = random.Random(99)
    raw = _corpus(rng, 20000)
    comp = zlib.compress(raw, 6)[2:-4]
    for cut in range(1, len(comp), 97):
        rc, pos, out, _ = oracle.infl(comp[:cut], len(raw), 0)
        assert rc == ol.ERR
    agree = 0
    for trial in range(3000):
        b = bytearray(comp[:rng.randint(1, 600)] if trial % 2 else comp)
        for _ in range(rng.randint(1, 3)):
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        d = zlib.decompressobj(wbits=-15)
        try:
            want = d.decompress(bytes(b), 200000)
            ok = d.eof
        except zlib.error:
            ok = False
        rc, pos, out, _ = oracle.infl(bytes(b), 200000, 0)
        if ok and rc == ol.OK:
            assert out == want
            agree += 1
        if ok:
            # a stream zlib accepts in full is a valid stream: the reference accepts it too
            assert rc == ol.OK
    assert agree > 0


def test_text_generator_gathered_equals_joined():
    """The benchmark's text generator builds a stream with array operations; the definition is the plain join of
    words and separators.  Same random draws, same bytes, at sizes from nothing to 256 KiB."""
    gen = wl.TextGen()
    for i, size in [(0, 65536), (1, 65536), (77, 1024), (5, 3), (9, 0), (1000, 262144), (31, 50000), (5000, 128), (12, 1)]:
        assert gen.make(i, size) == gen.make_joined(i, size), (i, size)
    assert wl.text_stream(3, 4096) == gen.make_joined(3, 4096)
