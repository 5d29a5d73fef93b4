#!/usr/bin/env python3
"""Test infrastructure (it checks every result against the CPU oracle, hence its place under tests/).
Secondary measurements on one GPU (not the headline): the parity-test configurations of the
build's measurement plan, each decoded from device-resident arenas and checked against the CPU
oracle before a number is printed.

  stored   65 536 x 64 KiB incompressible payload as stored blocks + zlib wrapper (copy ceiling)
  fixed    the text workload compressed with Z_FIXED (fixed-Huffman blocks only)
  png      4 096 PNG-like images (128x128 RGBA scanlines, zlib-6) fed as IDAT pieces with
           infl_include() and decoded with one infl_many() call (host path, PCIe inclusive)
  mixed    text streams with log-uniform sizes 1 KiB .. 256 KiB, sorted by size (config 5 shape)
  runs, textruns, truncated, flushed, small, tiny, files, noisy, pngdev, gather, host
           what real batches hold besides clean text: byte runs, text with runs spliced in, a
           damaged stream in 64, flush points every few bytes, streams of 1 / 4 / 16 KiB, slices of
           local binaries and scripts, near fixed-length codes, PNG-like scanlines kept on the
           device, host arenas through infl_batch()
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads as wl  # noqa: E402


def run_device(eng, oracle, comps, raws_len, flags, reps, tile, steps=5, failing=0):
    import torch
    dev = torch.device("cuda", 0)
    n = len(comps) * tile
    lens = np.tile(np.array([len(c) for c in comps], dtype=np.uint64), tile)
    caps = np.tile(np.array(raws_len, dtype=np.uint64), tile)
    src_off = np.zeros(n + 1, dtype=np.uint64)
    dst_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=src_off[1:])
    np.cumsum(caps, out=dst_off[1:])
    pool = torch.from_numpy(np.frombuffer(b"".join(comps), dtype=np.uint8).copy()).to(dev)
    d_src = torch.cat([pool.repeat(tile), torch.zeros(64, dtype=torch.uint8, device=dev)])
    d_dst = torch.empty(int(dst_off[-1]) + 64, dtype=torch.uint8, device=dev)
    d_so = torch.from_numpy(src_off.view(np.int64)).to(dev)
    d_do = torch.from_numpy(dst_off.view(np.int64)).to(dev)
    d_st = torch.full((n,), -99, dtype=torch.int32, device=dev)
    d_ol = torch.zeros(n, dtype=torch.int32, device=dev)

    def step():
        assert eng.lib.infl_batch_dev(d_src.data_ptr(), d_so.data_ptr(), d_dst.data_ptr(), d_do.data_ptr(), flags, n,
                                      d_st.data_ptr(), d_ol.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    step()
    torch.cuda.synchronize()
    k = min(len(comps), 128)
    s, so, do = wl.pack(comps[:k], raws_len[:k])
    want, wst, wol = oracle.infl_batch(s, so, do, flags, 8)
    assert np.array_equal(d_st[:k].cpu().numpy(), wst) and int((d_st != 0).sum()) == failing * tile
    got = d_dst[: int(do[-1])].cpu().numpy()
    for i in range(k):
        a, m = int(do[i]), int(wol[i])
        assert np.array_equal(got[a:a + m], want[a:a + m]), "differs from the oracle"
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record(); step(); b.record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    out_b, in_b = int(dst_off[-1]), int(src_off[-1])
    return {"streams": n, "out_bytes": out_b, "in_bytes": in_b, "kernel_ms": round(ms, 3),
            "decompressed_MB_per_s": round(out_b / ms / 1e3, 1),
            "algorithmic_GB_per_s": round((out_b + in_b) / ms / 1e6, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="stored,fixed,mixed,runs,textruns,truncated,flushed,small,tiny,files,noisy,pngdev,gather,host,png")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--variant", type=int, default=-1, help="table variant (default: the library's)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per stream in the serial passes (default: the library's)")
    args = ap.parse_args()
    import torch  # noqa: F401  (same HIP runtime for the engine)
    from defl_amd import capi
    from tests import oracle_lib
    eng, oracle = capi.load(), oracle_lib.load()
    assert eng.lib.infl_gpu_init(0) == 0
    if args.variant >= 0:
        eng.lib.infl_gpu_set_table_variant(args.variant)
    if args.lanes:
        eng.lib.infl_gpu_set_lanes_per_stream(args.lanes)
    which = args.which.split(",")
    if "stored" in which:
        pairs = [wl.stored_stream(i, 65536) for i in range(256)]
        r = run_device(eng, oracle, [p[1] for p in pairs], [65536] * 256, 1, 5, int(256 * args.scale))
        print(json.dumps({"config": "stored-only 64 KiB streams", **r}), flush=True)
    if "fixed" in which:
        raws = [wl.text_stream(i) for i in range(256)]
        comps = [wl.deflate(x, 6, True, zlib.Z_FIXED) for x in raws]
        r = run_device(eng, oracle, comps, [65536] * 256, 1, 5, int(256 * args.scale))
        print(json.dumps({"config": "fixed-Huffman-only 64 KiB text streams", **r}), flush=True)
    if "mixed" in which:
        rng = np.random.default_rng(5)
        sizes = np.sort(np.exp(rng.uniform(np.log(1024), np.log(262144), 512)).astype(int))[::-1]
        raws = [wl.text_stream(1000 + i, int(sz)) for i, sz in enumerate(sizes)]
        comps = [wl.deflate(x) for x in raws]
        r = run_device(eng, oracle, comps, [len(x) for x in raws], 1, 5, int(64 * args.scale))
        print(json.dumps({"config": "mixed sizes 1 KiB..256 KiB (log-uniform), sorted descending", **r}), flush=True)
    if "runs" in which:
        # very compressible: long runs and repeats (matches of length 258, distance 1 and far)
        rng = np.random.default_rng(7)
        raws = []
        for i in range(64):
            parts = []
            while sum(len(x) for x in parts) < 65536:
                parts.append(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 4000)))
            raws.append(b"".join(parts)[:65536])
        comps = [wl.deflate(x) for x in raws]
        r = run_device(eng, oracle, comps, [65536] * 64, 1, 5, int(1024 * args.scale))
        print(json.dumps({"config": "byte runs (ratio %.4f)" % (sum(map(len, comps)) / (64 * 65536.0)), **r}), flush=True)
    if "textruns" in which:
        # ordinary text with runs spliced in (flat image areas, padding, repeated records in real files)
        rng = np.random.default_rng(11)
        raws = []
        for i in range(256):
            t = bytearray(wl.text_stream(i))
            for _ in range(int(rng.integers(0, 4))):
                at, n = int(rng.integers(0, 60000)), int(rng.integers(300, 5000))
                t[at:at + n] = bytes([int(rng.integers(0, 256))]) * n
            raws.append(bytes(t))
        comps = [wl.deflate(x) for x in raws]
        r = run_device(eng, oracle, comps, [65536] * 256, 1, 5, int(256 * args.scale))
        print(json.dumps({"config": "text with runs spliced in (ratio %.3f)" % (sum(map(len, comps)) / (256 * 65536.0)), **r}), flush=True)
    if "truncated" in which:
        # the headline text batch with one stream in 64 cut off in its last quarter (a damaged file in
        # the batch): those streams fail, the batch must not wait for them
        raws = [wl.text_stream(i) for i in range(256)]
        comps = [wl.deflate(x) for x in raws]
        cut = [i for i in range(0, 256, 64)]
        for i in cut:
            comps[i] = comps[i][: len(comps[i]) * 7 // 8]
        r = run_device(eng, oracle, comps, [65536] * 256, 1, 5, int(256 * args.scale), failing=len(cut))
        r["decompressed_MB_per_s"] = round(r["decompressed_MB_per_s"] * (256 - len(cut)) / 256.0, 1)
        print(json.dumps({"config": "64 KiB text streams, 1 in 64 truncated (rate over the intact ones)", **r}), flush=True)
    if "flushed" in which:
        # the text workload with a flush point every 64..192 bytes (interactive protocols): some 500 tiny
        # blocks per stream
        import random
        rng = random.Random(21)
        raws = [wl.text_stream(i) for i in range(128)]
        comps = []
        for x in raws:
            c, out, i = zlib.compressobj(6), [], 0
            while i < len(x):
                n = rng.randint(64, 192)
                out.append(c.compress(x[i:i + n]) + c.flush(zlib.Z_SYNC_FLUSH))
                i += n
            comps.append(b"".join(out) + c.flush())
        r = run_device(eng, oracle, comps, [65536] * 128, 1, 5, int(64 * args.scale))
        print(json.dumps({"config": "64 KiB text streams flushed every 64..192 bytes (ratio %.3f)"
                          % (sum(map(len, comps)) / (128 * 65536.0)), **r}), flush=True)
    if "small" in which:
        # many small streams (HTTP payloads, small images): per-stream and per-block costs dominate
        for size in (1024, 4096, 16384):
            raws = [wl.text_stream(5000 + i, size) for i in range(512)]
            comps = [wl.deflate(x) for x in raws]
            r = run_device(eng, oracle, comps, [size] * 512, 1, 5, int((64 << 20) // size // 512 * 4 * args.scale))
            print(json.dumps({"config": "text streams of %d bytes" % size, **r}), flush=True)
    if "tiny" in which:
        # records of a few hundred bytes: under 256 input bytes the serial pass takes them, several to a wavefront
        for size in (128, 256, 512):
            raws = [wl.text_stream(5000 + i, size) for i in range(512)]
            comps = [wl.deflate(x) for x in raws]
            r = run_device(eng, oracle, comps, [size] * 512, 1, 5, int(128 * args.scale))
            print(json.dumps({"config": "text streams of %d bytes" % size, **r}), flush=True)
        # a corpus with everything from tiny records to 64 KiB (log-uniform sizes, in no order)
        rng = np.random.default_rng(11)
        sizes = np.exp(rng.uniform(np.log(64), np.log(65536), 2048)).astype(np.int64)
        raws = [wl.text_stream(6000 + i, int(sz)) for i, sz in enumerate(sizes)]
        comps = [wl.deflate(x) for x in raws]
        r = run_device(eng, oracle, comps, [len(x) for x in raws], 1, 5, int(16 * args.scale))
        print(json.dumps({"config": "wide mix 64 B..64 KiB (log-uniform, unsorted; %d%% of the streams under 256 input bytes)"
                                    % round(100.0 * sum(len(c) < 256 for c in comps) / len(comps)), **r}), flush=True)
    if "files" in which:
        # 64 KiB slices of binaries, libraries and scripts found on this machine, zlib level 6
        import glob
        paths = [p for pat in ("/usr/bin/python3*", "/usr/lib/x86_64-linux-gnu/libc.so*",
                               "/usr/lib/x86_64-linux-gnu/libstdc++.so*", "/usr/lib/python3.10/*.py", "/usr/bin/gcc*")
                 for p in sorted(glob.glob(pat)) if os.path.isfile(p)]
        raws = []
        for pth in paths:
            data = open(pth, "rb").read(1 << 21)
            raws += [data[o:o + 65536] for o in range(0, len(data) - 65535, 65536)]
        raws = raws[:512]
        comps = [wl.deflate(x) for x in raws]
        r = run_device(eng, oracle, comps, [65536] * len(raws), 1, 5, max(1, int(65536 // len(raws) * args.scale)))
        print(json.dumps({"config": "64 KiB slices of %d local binaries and scripts (ratio %.3f)"
                          % (len(paths), sum(map(len, comps)) / (65536.0 * len(raws))), **r}), flush=True)
    if "noisy" in which:
        # barely compressible: 6 random bits per byte, almost all literals with ~6-bit codes
        rng = np.random.default_rng(8)
        raws = [bytes(rng.integers(32, 96, 65536, dtype=np.uint8)) for i in range(64)]
        comps = [wl.deflate(x) for x in raws]
        r = run_device(eng, oracle, comps, [65536] * 64, 1, 5, int(1024 * args.scale))
        print(json.dumps({"config": "6-bit noise (ratio %.4f)" % (sum(map(len, comps)) / (64 * 65536.0)), **r}), flush=True)
    if "pngdev" in which:
        # the PNG-like scanline data of the IDAT configuration, device resident (kernel only)
        raws = [wl.png_like(i) for i in range(256)]
        comps = [wl.deflate(x) for x in raws]
        r = run_device(eng, oracle, comps, [len(x) for x in raws], 1, 5, int(128 * args.scale))
        print(json.dumps({"config": "PNG-like scanlines, device resident (ratio %.3f)"
                                    % (sum(map(len, comps)) / float(sum(map(len, raws)))), **r}), flush=True)
    if "gather" in which:
        # the IDAT configuration with the files already on the device: the IDAT payloads (11.5 pieces per image, 12
        # bytes of chunk header and CRC between them) are gathered into one arena, which is then decoded
        raws = [wl.png_like(i) for i in range(256)]
        comps = [wl.deflate(x) for x in raws]
        tile = int(16 * args.scale)
        blob, po, pl, pd, so, caps = bytearray(), [], [], [], [0], []
        for t in range(tile):
            for i, c in enumerate(comps):
                at = so[-1]
                for piece in wl.idat_pieces(i + 256 * t, c):
                    blob += b"\0" * 12
                    po.append(len(blob)); pl.append(len(piece)); pd.append(at)
                    blob += piece; at += len(piece)
                so.append(at); caps.append(len(raws[i]))
        n = len(caps)
        do = np.concatenate([[0], np.cumsum(caps)]).astype(np.uint64)
        b = eng.device_batch(np.zeros(so[-1], dtype=np.uint8), np.array(so, dtype=np.uint64), do)
        lib = eng.lib
        arrs = [np.frombuffer(bytes(blob), dtype=np.uint8), np.array(po, dtype=np.uint64), np.array(pl, dtype=np.uint32),
                np.array(pd, dtype=np.uint64)]
        dev = [lib.infl_gpu_malloc(a.nbytes + 16) for a in arrs]
        assert all(dev)
        for d, a in zip(dev, arrs):
            assert lib.infl_gpu_upload(d, a.ctypes.data, a.nbytes) == 0
        def gather():
            assert lib.infl_gather_dev(dev[0], dev[1], dev[2], dev[3], len(po), b.d_src, None) == 0
        gather(); b.run(1, detail=False)
        dst, status, outlen, _ = b.results()
        assert (status == 0).all()
        for i in (0, 1, n // 2, n - 1):
            assert dst[int(do[i]):int(do[i + 1])].tobytes() == raws[i % 256]
        times = {}
        for name, fn in (("gather", gather), ("gather + decode", lambda: (gather(), b.run(1, detail=False)))):
            assert lib.infl_gpu_sync() == 0
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            assert lib.infl_gpu_sync() == 0
            times[name] = (time.perf_counter() - t0) / 20
        for d in dev:
            lib.infl_gpu_free(d)
        b.free()
        print(json.dumps({"config": "PNG IDAT pieces gathered on the device (infl_gather_dev), then decoded", "streams": n,
                          "pieces": len(po), "gathered_bytes": int(so[-1]), "gather_ms": round(times["gather"] * 1e3, 3),
                          "gather_GB_per_s_read_plus_written": round(2 * so[-1] / times["gather"] / 1e9, 1),
                          "gather_plus_decode_ms": round(times["gather + decode"] * 1e3, 3),
                          "out_bytes": int(do[-1])}), flush=True)
    if "host" in which:
        # host arenas through infl_batch(): upload, three launches, download (PCIe inclusive)
        n = int(16384 * args.scale)
        base = [wl.deflate(wl.text_stream(i)) for i in range(256)]
        comps = [base[i % 256] for i in range(n)]
        src, src_off, dst_off = wl.pack(comps, [65536] * n)
        dst = np.zeros(int(dst_off[-1]), dtype=np.uint8)
        status = np.zeros(n, dtype=np.int32)
        outlen = np.zeros(n, dtype=np.uint32)
        so = np.ascontiguousarray(src_off, dtype=np.uint64)
        do = np.ascontiguousarray(dst_off, dtype=np.uint64)
        fn = eng.lib.infl_batch
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
        fn.restype = C.c_int
        call = lambda: fn(src.ctypes.data, so.ctypes.data, dst.ctypes.data, do.ctypes.data, 1, n,
                          status.ctypes.data, outlen.ctypes.data)
        assert call() == 0                      # warm-up (workspace allocation)
        t0 = time.perf_counter()
        assert call() == 0
        dt = time.perf_counter() - t0
        assert (status == 0).all() and (outlen == 65536).all()
        want, _, _ = oracle.infl_batch(src[: int(so[64])], so[:65], do[:65], 1, 8)
        assert np.array_equal(dst[: int(do[64])], want[: int(do[64])])
        print(json.dumps({"config": "host arenas through infl_batch (PCIe inclusive)", "streams": n,
                          "out_bytes": int(do[-1]), "in_bytes": int(so[-1]), "call_ms": round(dt * 1e3, 2),
                          "decompressed_MB_per_s": round(int(do[-1]) / dt / 1e6, 1)}), flush=True)
    if "png" in which:
        n = int(4096 * args.scale)
        raws = [wl.png_like(i % 256) for i in range(n)]
        comps = {i: wl.deflate(raws[i]) for i in range(min(n, 256))}
        lib = eng.lib
        outs = [(C.c_uint8 * len(raws[i]))() for i in range(n)]
        sts = (C.c_void_p * n)()
        keep, npieces = [], 0
        t0 = time.perf_counter()
        for i in range(n):
            sts[i] = lib.infl_init(outs[i], len(raws[i]), 1)
            for piece in wl.idat_pieces(i % 256, comps[i % 256]):
                buf = (C.c_uint8 * len(piece)).from_buffer_copy(piece)
                keep.append(buf)
                lib.infl_include(sts[i], buf, len(piece))
                npieces += 1
        t_inc = time.perf_counter() - t0
        status = (C.c_int32 * n)()
        lib.infl_many(sts, n, status)            # warm-up (workspace allocation)
        t0 = time.perf_counter()
        assert lib.infl_many(sts, n, status) == 0
        t_many = time.perf_counter() - t0
        assert all(s == 0 for s in status)
        for i in range(0, n, 97):
            assert bytes(outs[i]) == raws[i]
        out_b = sum(len(x) for x in raws)
        for i in range(n):
            lib.infl_destroy(sts[i])
        print(json.dumps({"config": "PNG IDAT: %d images, infl_include per piece + one infl_many (host memory, "
                                    "PCIe and gather included)" % n, "streams": n, "pieces_per_image": round(npieces / n, 1),
                          "out_bytes": out_b, "infl_many_ms": round(t_many * 1e3, 2),
                          "decompressed_MB_per_s": round(out_b / t_many / 1e6, 1),
                          "include_calls_ms": round(t_inc * 1e3, 1)}), flush=True)


if __name__ == "__main__":
    main()
