"""infl_batch_dev() and infl_batch_dev_sorted() recorded into a HIP graph and replayed: the calls are kernel
launches on the caller's stream and nothing else (no allocation, no copy, no wait, no memset), so a caller can
capture them once and replay them on new data in the same buffers.  Every replay is checked against the CPU
oracle, with eager work (uploads, fills) and synchronisations between the replays."""
import ctypes as C
import random

import numpy as np
import pytest

import workloads as wl

pytestmark = pytest.mark.gpu

HIP_STREAM_CAPTURE_MODE_RELAXED = 2


def _hip():
    """The HIP runtime the engine is using in this process (the one of /opt/rocm, or the one that came with torch
    when a test before this one imported torch first: both have the same soname, whichever is loaded first serves)."""
    paths = {line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line}
    assert len(paths) == 1, paths
    hip = C.CDLL(paths.pop())
    for name, args in (("hipStreamCreate", [C.POINTER(C.c_void_p)]), ("hipStreamDestroy", [C.c_void_p]),
                       ("hipStreamSynchronize", [C.c_void_p]), ("hipStreamBeginCapture", [C.c_void_p, C.c_int]),
                       ("hipStreamEndCapture", [C.c_void_p, C.POINTER(C.c_void_p)]),
                       ("hipGraphInstantiate", [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]),
                       ("hipGraphLaunch", [C.c_void_p, C.c_void_p]), ("hipGraphExecDestroy", [C.c_void_p]),
                       ("hipGraphDestroy", [C.c_void_p]), ("hipGraphGetNodes", [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)])):
        fn = getattr(hip, name)
        fn.argtypes, fn.restype = args, C.c_int
    return hip


def _inputs(seed, n):
    """Two sets of streams that fit the same slots: slot i holds set A's or set B's stream i, zero padded."""
    rng = random.Random(seed)
    sets = []
    for which in range(2):
        comps, caps = [], []
        for k in range(n):
            size = int(2 ** rng.uniform(3, 14.3))
            raw = wl.text_stream(100000 * which + 31 * seed + k, size)
            comp = wl.deflate(raw) if k % 7 else wl.stored_stream(k + which, size)[1]
            if k % 29 == 5 + which and len(comp) > 40:
                bad = bytearray(comp); bad[len(bad) // 2] ^= 0x11; comp = bytes(bad)      # damaged
            comps.append(comp)
            caps.append(size if k % 31 != 7 else max(0, size - 9))                          # some short of space
        sets.append((comps, caps))
    slot = [max(len(sets[0][0][i]), len(sets[1][0][i])) + (i % 3) for i in range(n)]
    cap = [max(sets[0][1][i], sets[1][1][i]) for i in range(n)]
    src_off = np.concatenate(([0], np.cumsum(slot))).astype(np.uint64)
    dst_off = np.concatenate(([0], np.cumsum(cap))).astype(np.uint64)
    arenas = []
    for comps, _ in sets:
        a = np.zeros(int(src_off[-1]) + 16, dtype=np.uint8)
        for i, c in enumerate(comps):
            a[int(src_off[i]): int(src_off[i]) + len(c)] = np.frombuffer(c, dtype=np.uint8)
        arenas.append(a)
    return arenas, src_off, dst_off


@pytest.mark.parametrize("sorted_call", [False, True])
def test_graph_replays_equal_the_oracle(engine, oracle, sorted_call):
    lib = engine.lib
    hip = _hip()
    n = 7000                                                # (short streams go to the serial passes at this size)
    arenas, src_off, dst_off = _inputs(77 + sorted_call, n)
    want = [oracle.infl_batch(a[: int(src_off[-1])], src_off, dst_off, 1) for a in arenas]
    assert any((w[1] != 0).any() for w in want) and not np.array_equal(want[0][2], want[1][2])
    b = engine.device_batch(arenas[0][: int(src_off[-1])], src_off, dst_off)
    work = lib.infl_gpu_malloc(1024 + 4 * n + 16)
    stream, graph, gexec = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert work and hip.hipStreamCreate(C.byref(stream)) == 0
    try:
        assert hip.hipStreamBeginCapture(stream, HIP_STREAM_CAPTURE_MODE_RELAXED) == 0
        if sorted_call:
            rc = lib.infl_batch_dev_sorted(b.d_src, b.d_soff, b.d_dst, b.d_doff, 1, n, b.d_status, b.d_outlen, work, stream)
        else:
            rc = lib.infl_batch_dev(b.d_src, b.d_soff, b.d_dst, b.d_doff, 1, n, b.d_status, b.d_outlen, stream)
        assert hip.hipStreamEndCapture(stream, C.byref(graph)) == 0 and rc == 0 and graph
        nodes = C.c_size_t(0)
        assert hip.hipGraphGetNodes(graph, None, C.byref(nodes)) == 0
        assert nodes.value == (7 if sorted_call else 3)     # kernel launches only: the decode's three passes (+ the sort's four)
        assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
        fill = np.full(max(1, b.dst_bytes), 0xA5, dtype=np.uint8)
        junk = np.full(n, 0x5a5a5a5a, dtype=np.uint32)
        for rep, which in enumerate([0, 1, 1, 0, 1]):
            # eager work between the replays: the other input set into the same buffers, results wiped
            assert lib.infl_gpu_upload(b.d_src, arenas[which].ctypes.data, b.src_bytes) == 0
            assert lib.infl_gpu_upload(b.d_dst, fill.ctypes.data, b.dst_bytes) == 0
            assert lib.infl_gpu_upload(b.d_status, junk.ctypes.data, 4 * n) == 0
            assert lib.infl_gpu_upload(b.d_outlen, junk.ctypes.data, 4 * n) == 0
            assert hip.hipGraphLaunch(gexec, stream) == 0
            assert hip.hipStreamSynchronize(stream) == 0
            dst, status, outlen, _ = b.results()
            want_dst, want_status, want_outlen = want[which]
            assert np.array_equal(status, want_status), (rep, which)
            assert np.array_equal(outlen, want_outlen), (rep, which)
            for i in range(n):
                a, m = int(dst_off[i]), int(outlen[i])
                assert np.array_equal(dst[a:a + m], want_dst[a:a + m]), (rep, which, i)
    finally:
        if gexec:
            hip.hipGraphExecDestroy(gexec)
        if graph:
            hip.hipGraphDestroy(graph)
        hip.hipStreamDestroy(stream)
        lib.infl_gpu_free(work)
        b.free()
