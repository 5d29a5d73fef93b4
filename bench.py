#!/usr/bin/env python3
"""Headline benchmark: decompressed MB/s of a batch of independent zlib streams on N x MI355X.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--workload text64k|mixed]

Workloads (synthetic Zipf text, zlib level 6, dynamic Huffman):
  text64k  65 536 streams of 64 KiB (the configuration the metric is quoted on);
  mixed    1 048 576 streams of 1 KiB .. 256 KiB, log-uniform, sizes in no order (the sharded mixed-size
           configuration), decoded with infl_batch_dev_sorted (size classes made on the device; the rate of the plain
           call is measured beside it).
Scaling:
  weak     every GPU decodes a batch of the full size (default: per-GPU work fixed as N grows);
  strong   ONE batch of the full size is cut into N contiguous ranges that balance the sum of
           compressed + decompressed bytes (defl_amd.shard.partition), one range per GPU.
One process per GPU, no data-path collective: streams are independent.  RCCL carries only the
result counters and the per-rank times.  Inputs and outputs stay in HBM; a step is one pass of the
hot path (one infl_batch_dev call) over the rank's streams.  Started without a launcher and with
--gpus N > 1, the script starts `torch.distributed.run` itself (before it touches any GPU) and
relays the result.  Prints ONE JSON line on rank 0.

With one GPU the line also carries `secondary`: the other configurations of the measurement plan, each run
behind the headline's timed region and parity gate with a parity gate of its own over every stream (stored-only
and fixed-Huffman-only batches, the PNG IDAT configuration through infl_include + infl_many and through the
device-side gather, mixed sizes in pool order and by falling size).
"""
import argparse
import ctypes as C
import json
from concurrent.futures import ThreadPoolExecutor
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import workloads as wl  # noqa: E402
from defl_amd import shard  # noqa: E402

STREAM_BYTES = 65536
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (spec), GB/s
FULL_STREAMS = {"text64k": 65536, "mixed": 1048576}


RESULT_OUT = sys.stdout


def note(msg):
    """Progress on stderr (stdout carries only the JSON line)."""
    if os.environ.get("RANK", "0") == "0":
        print("[bench %s] %s" % (time.strftime("%H:%M:%S"), msg), file=sys.stderr, flush=True)


def pool_sizes(workload, pool):
    """Decompressed size of every distinct stream of the pool."""
    if workload == "text64k":
        return [STREAM_BYTES] * pool
    rng = np.random.default_rng(5)
    return [int(x) for x in np.exp(rng.uniform(np.log(1024), np.log(262144), pool))]


def _make_stream(arg):
    i, size = arg
    return wl.deflate(wl.text_stream(i, size), 6, True)


def pool_map(fn, jobs, workers):
    """Threads, not processes: a fork after the HIP runtime is up (a profiler preloads it before
    main) can deadlock the children.  zlib and numpy release the GIL for most of the work."""
    if workers > 1:
        with ThreadPoolExecutor(workers) as ex:
            return list(ex.map(fn, jobs))
    return [fn(j) for j in jobs]


def build_pool(first, sizes, workers):
    return pool_map(_make_stream, [(first + k, s) for k, s in enumerate(sizes)], workers)


def host_threads():
    """Threads this process may use: the cores it is allowed on, capped by the share the machine gives the job
    (OMP_NUM_THREADS, where the operator sets it)."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    cap = os.environ.get("OMP_NUM_THREADS", "")
    if cap.isdigit() and int(cap) > 0:
        allowed = min(allowed, int(cap))
    return max(1, allowed)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(comps, sizes, oracle, threads, budget_s):
    """The CPU oracle (a port of the reference's infl_buf path; the reference itself cannot be
    built here) timed on the host cores over streams of the same workload: `threads` threads with
    static partitioning, passes repeated until about `budget_s` seconds of core time are spent;
    the value is the median pass.  The value is the reference's reuse pattern (one stream object
    reset between inputs: nothing is allocated per stream); the pattern of infl_buf() called as it
    stands (a stream object allocated, filled and freed per call), one thread and the system's zlib
    are timed beside it."""
    src, src_off, dst_off = wl.pack(comps, sizes)
    oracle.infl_batch(src, src_off, dst_off, 1, threads)            # warm-up (tables, pages)
    times, out_bytes = [], 0
    while (sum(times) * threads < budget_s or len(times) < 3) and len(times) < 64:
        t0 = time.perf_counter()
        _, status, outlen = oracle.infl_batch(src, src_off, dst_off, 1, threads)
        times.append(time.perf_counter() - t0)
        out_bytes = int(outlen.sum())
        assert (status == 0).all()
    as_is = []
    for _ in range(5):
        t0 = time.perf_counter()
        oracle.infl_batch(src, src_off, dst_off, 1, threads, as_is=True)
        as_is.append(time.perf_counter() - t0)
    sub = max(1, len(comps) // 16)
    s1, so1, do1 = wl.pack(comps[:sub], sizes[:sub])
    t1 = []
    for _ in range(3):
        t0 = time.perf_counter()
        _, _, ol1 = oracle.infl_batch(s1, so1, do1, 1, 1)
        t1.append(time.perf_counter() - t0)
    import zlib
    t0 = time.perf_counter()
    zbytes = sum(len(zlib.decompress(c)) for c in comps[:sub])
    tz = time.perf_counter() - t0
    med = float(np.median(times))
    return {
        "value": round(out_bytes / med / 1e6, 1),
        "unit": "MB/s",
        "cores": threads,
        "kind": "port",
        "pattern": "one stream object reset between inputs (infl_init once, then infl_reset + infl_include + infl per "
                   "stream): no allocation per stream",
        "as_is_MB_per_s": round(out_bytes / float(np.median(as_is)) / 1e6, 1),
        "as_is_pattern": "infl_buf() as it stands: a stream object allocated, filled and freed per call",
        "nproc": os.cpu_count(),
        "cpu_model": cpu_model(),
        "threads": threads,
        "passes": len(times),
        "spread_MB_per_s": [round(out_bytes / max(times) / 1e6, 1), round(out_bytes / min(times) / 1e6, 1)],
        "one_thread_MB_per_s": round(int(ol1.sum()) / float(np.median(t1)) / 1e6, 1),
        "system_zlib_one_thread_MB_per_s": round(zbytes / tz / 1e6, 1),
        "sample": "%d distinct streams of this workload, median of %d passes on %d threads (the cores the job "
                  "may use; the machine has %d)" % (len(comps), len(times), threads, os.cpu_count() or 0),
    }


def relaunch(args, argv):
    """--gpus N > 1 without a launcher: start one process per GPU through torch.distributed.run
    and pass its output on.  Runs before this process has touched a GPU and never touches one."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


class Batch:
    """This rank's streams.  The batch is a sequence of `full` streams drawn from a pool of distinct ones: in pool
    order (stream g of the batch is pool stream g % pool) or, with `perm`, grouped (stream g is pool stream
    perm[g // reps]: every distinct stream `reps` times in a row, in the order perm gives, e.g. by falling size).
    The rank owns the contiguous range [lo, hi).  Device layout: one source arena + u64 offsets, one destination
    arena + u64 offsets (capacity = the stream's decompressed size), status and length arrays."""

    def __init__(self, torch, dev, comps, sizes, lo, hi, perm=None, reps=1):
        self.torch, self.dev, self.lo, self.hi = torch, dev, lo, hi
        pool = len(comps)
        self.pool, self.n, self.perm, self.reps = pool, hi - lo, perm, reps
        clen = np.array([len(c) for c in comps], dtype=np.uint64)
        ulen = np.array(sizes, dtype=np.uint64)
        g = np.arange(lo, hi, dtype=np.int64)
        idx = g % pool if perm is None else np.asarray(perm, dtype=np.int64)[g // reps]
        self.src_off = np.zeros(self.n + 1, dtype=np.uint64)
        self.dst_off = np.zeros(self.n + 1, dtype=np.uint64)
        np.cumsum(clen[idx], out=self.src_off[1:])
        np.cumsum(ulen[idx], out=self.dst_off[1:])
        self.comp_bytes, self.out_bytes = int(self.src_off[-1]), int(self.dst_off[-1])
        self.cs = np.concatenate([[0], np.cumsum(clen)]).astype(np.int64)       # pool prefix sums
        self.us = np.concatenate([[0], np.cumsum(ulen)]).astype(np.int64)
        blk = torch.from_numpy(np.frombuffer(b"".join(comps), dtype=np.uint8).copy()).to(dev)
        pad = torch.zeros(64, dtype=torch.uint8, device=dev)
        if perm is None:
            # the pool block tiled on the device, cut to the rank's range
            b0, b1 = lo // pool, (hi + pool - 1) // pool                       # blocks touched
            tiled = blk.repeat(max(1, b1 - b0))
            start = int(self.cs[lo - b0 * pool])
            self.d_src = torch.cat([tiled[start:start + self.comp_bytes], pad])
            del tiled
        else:
            parts = [blk[int(self.cs[j]):int(self.cs[j + 1])].repeat(c) for j, c in self._groups()]
            self.d_src = torch.cat(parts + [pad])
            del parts
        del blk
        self.d_dst = torch.empty(self.out_bytes + 64, dtype=torch.uint8, device=dev)
        self.d_soff = torch.from_numpy(self.src_off.view(np.int64)).to(dev)
        self.d_doff = torch.from_numpy(self.dst_off.view(np.int64)).to(dev)
        self.d_status = torch.full((max(1, self.n),), -99, dtype=torch.int32, device=dev)
        self.d_outlen = torch.zeros(max(1, self.n), dtype=torch.int32, device=dev)
        self.want_len = torch.from_numpy(ulen[idx].astype(np.int32)).to(dev)

    def _groups(self):
        """(pool stream, how many times in a row) over the rank's range of a grouped batch."""
        out, g = [], self.lo
        while g < self.hi:
            k = g // self.reps
            c = min((k + 1) * self.reps, self.hi) - g
            out.append((int(self.perm[k]), c))
            g += c
        return out

    def verify(self, oracle, comps, sizes, threads):
        """Every output of the rank against the CPU oracle: the first occurrence of every distinct
        stream byte for byte on the host, every further occurrence byte for byte against that first
        one on the device.  Returns (ok, streams checked against the oracle, streams checked in all)."""
        torch, pool, lo, hi = self.torch, self.pool, self.lo, self.hi
        ok = bool((self.d_status[: self.n] == 0).all()) and bool((self.d_outlen[: self.n] == self.want_len).all())
        if not ok or self.n == 0:
            return ok, 0, 0
        blk_out = int(self.us[-1])
        # reference block: the oracle's output for pool streams 0 .. pool-1, on the device
        src, so, do = wl.pack(comps, sizes)
        want, wst, wol = oracle.infl_batch(src, so, do, 1, threads)
        ok = bool((wst == 0).all()) and bool((wol == np.array(sizes)).all())
        d_want = torch.from_numpy(want[:blk_out]).to(self.dev)
        at, checked = 0, 0
        if self.perm is None:
            # the rank's output is a run of (partial first block, whole blocks, partial last block)
            g = lo
            while g < hi and ok:
                j = g % pool
                cnt = min(pool - j, hi - g)                                     # streams up to the block's end
                a, b = int(self.us[j]), int(self.us[j + cnt])
                ok = torch.equal(self.d_dst[at:at + (b - a)], d_want[a:b])
                at += b - a
                g += cnt
                checked += cnt
        else:
            for j, c in self._groups():
                if not ok:
                    break
                a, b = int(self.us[j]), int(self.us[j + 1])
                if b > a:
                    ok = bool((self.d_dst[at:at + c * (b - a)].view(c, b - a) == d_want[a:b]).all())
                at += c * (b - a)
                checked += c
        return ok, min(pool, self.n), checked


class HostEvent:
    """Stands in for a device event where the decode runs on the host (the multi-process rehearsal on CPU tensors)."""

    def __init__(self):
        self.t = 0.0

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class OracleEngine:
    """Rehearsal only (DEFL_BENCH_STUB=1, set by the multi-process CPU test): the call bench.py makes, answered by the CPU
    oracle on CPU tensors, so that argument handling, process group, barriers, partition, parity gate and reductions of
    an N-rank run execute without a GPU.  Never used for a measurement: the result line says so."""

    def __init__(self, oracle):
        self.o = oracle.lib

    def infl_batch_dev(self, src, soff, dst, doff, flags, n, status, outlen, stream):
        return self.o.oracle_infl_batch(src, soff, dst, doff, flags, n, status, outlen, 1)


def time_steps(torch, step, steps, warmup, barrier, stub):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides.  -> (seconds for the K
    steps, mean milliseconds per step by events on the launch stream)."""
    sync = (lambda: None) if stub else torch.cuda.synchronize
    for _ in range(max(1, warmup)):
        step()
    sync()
    mk = HostEvent if stub else (lambda: torch.cuda.Event(enable_timing=True))
    ev = [(mk(), mk()) for _ in range(steps)]
    barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(steps):
        ev[k][0].record()
        step()
        ev[k][1].record()
    sync()
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, float(np.mean([x.elapsed_time(y) for x, y in ev]))


def make_step(eng, torch, b, flags, stub, work=None):
    """One pass of the hot path over the batch: infl_batch_dev, or with `work` (device memory for the call's own use)
    infl_batch_dev_sorted, which puts the batch into size classes first."""
    def step():
        stream = None if stub else torch.cuda.current_stream().cuda_stream
        if work is None:
            rc = eng.infl_batch_dev(b.d_src.data_ptr(), b.d_soff.data_ptr(), b.d_dst.data_ptr(), b.d_doff.data_ptr(),
                                    flags, b.n, b.d_status.data_ptr(), b.d_outlen.data_ptr(), stream)
        else:
            rc = eng.infl_batch_dev_sorted(b.d_src.data_ptr(), b.d_soff.data_ptr(), b.d_dst.data_ptr(), b.d_doff.data_ptr(),
                                           flags, b.n, b.d_status.data_ptr(), b.d_outlen.data_ptr(), work.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError("infl_batch_dev failed")
    return step


# ---------------------------------------------------------------------------------------------------------------
# Secondary configurations (one GPU): each decodes from device-resident arenas, is timed over `steps` launches with
# events on the launch stream and passes a parity gate over every stream before its numbers are reported.
# ---------------------------------------------------------------------------------------------------------------
def secondary_device(eng, torch, dev, oracle, threads, name, comps, sizes, full, steps, perm=None, reps=1, sort=False):
    b = Batch(torch, dev, comps, sizes, 0, full, perm, reps)
    work = torch.empty(1024 + 4 * full + 16, dtype=torch.uint8, device=dev) if sort else None
    step = make_step(eng.lib, torch, b, 1, False, work)
    _, ms = time_steps(torch, step, steps, 1, lambda: None, False)
    ok, n_oracle, n_checked = b.verify(oracle, comps, sizes, threads)
    if not ok or n_checked != full:
        raise SystemExit("parity gate failed: %s" % name)
    algo = (b.comp_bytes + b.out_bytes) / (ms * 1e-3) / 1e9
    res = {"config": name, "streams": full, "distinct_streams": len(comps), "out_bytes": b.out_bytes,
           "in_bytes": b.comp_bytes, "kernel_ms": round(ms, 3),
           "decompressed_MB_per_s": round(b.out_bytes / ms / 1e3, 1), "algorithmic_GB_per_s": round(algo, 1),
           "roofline_frac": round(algo / HBM_PEAK_GBS, 4),
           "parity_gate": "every stream: %d distinct against the CPU oracle, %d against those on the device" % (n_oracle, n_checked)}
    del b
    torch.cuda.empty_cache()
    return res


def png_workload(n, distinct=256):
    raws = [wl.png_like(i) for i in range(distinct)]
    comps = [wl.deflate(r) for r in raws]
    pieces = [wl.idat_pieces(i, comps[i % distinct]) for i in range(n)]
    return raws, comps, pieces


def secondary_png_host(eng, n, raws, comps, pieces):
    """The IDAT configuration through the reference's own calls: infl_init per image, infl_include per IDAT piece, one
    infl_many() for all images (host memory, gather and PCIe included).  Two generations of stream objects: the second
    is made after the first was destroyed (a caller that makes a stream per image, batch after batch)."""
    lib = eng.lib
    distinct = len(raws)
    blobs = [np.frombuffer(c, dtype=np.uint8).copy() for c in comps]
    addr = [b.ctypes.data for b in blobs]
    outs = [(C.c_uint8 * len(raws[i % distinct]))() for i in range(n)]
    status = (C.c_int32 * n)()
    res = {}
    for gen in (1, 2):
        sts = (C.c_void_p * n)()
        npieces = 0
        t0 = time.perf_counter()
        for i in range(n):
            st = lib.infl_init(outs[i], len(raws[i % distinct]), 1)
            sts[i] = st
            at = addr[i % distinct]
            for p in pieces[i]:
                lib.infl_include(st, at, len(p))
                at += len(p)
            npieces += len(pieces[i])
        t_inc = time.perf_counter() - t0
        if gen == 1:
            lib.infl_many(sts, n, status)              # warm-up (workspace allocation)
        t0 = time.perf_counter()
        rc = lib.infl_many(sts, n, status)
        t_many = time.perf_counter() - t0
        if rc != 0 or any(s != 0 for s in status) or any(bytes(outs[i]) != raws[i % distinct] for i in range(n)):
            raise SystemExit("parity gate failed: PNG IDAT through infl_include + infl_many")
        t0 = time.perf_counter()
        for i in range(n):
            lib.infl_destroy(sts[i])
        t_destroy = time.perf_counter() - t0
        key = "" if gen == 1 else "_second_generation"
        res["include_calls_ms" + key] = round(t_inc * 1e3, 1)
        res["infl_many_ms" + key] = round(t_many * 1e3, 2)
        res["destroy_ms" + key] = round(t_destroy * 1e3, 1)
    out_b = sum(len(raws[i % distinct]) for i in range(n))
    res.update({"config": "PNG IDAT: %d images, infl_init + infl_include per piece (ctypes calls) + one infl_many "
                          "(host memory; gather, PCIe and scatter included)" % n,
                "streams": n, "pieces": npieces, "pieces_per_image": round(npieces / n, 1), "out_bytes": out_b,
                "decompressed_MB_per_s": round(out_b / (res["infl_many_ms_second_generation"] * 1e-3) / 1e6, 1),
                "parity_gate": "every image byte for byte against its pixels' scanlines, both generations"})
    return res


def secondary_png_device(eng, torch, n, raws, comps, pieces, steps, height=128, rowbytes=512, bpp=4):
    """The IDAT configuration with the files already on the device: the IDAT payloads (12 bytes of chunk header and CRC
    between them) are gathered into one arena (infl_gather_dev), decoded (infl_batch_dev) and un-filtered
    (infl_png_unfilter_dev); every decoded and every rebuilt byte is compared."""
    lib = eng.lib
    distinct = len(raws)
    blob, po, pl, pd, so, caps = bytearray(), [], [], [], [0], []
    offs = {}
    for i in range(n):
        at = so[-1]
        for p in pieces[i]:
            blob += b"\0" * 12
            po.append(len(blob)); pl.append(len(p)); pd.append(at)
            blob += p
            at += len(p)
        so.append(at)
        caps.append(len(raws[i % distinct]))
    do = np.concatenate([[0], np.cumsum(caps)]).astype(np.uint64)
    oo = np.arange(n + 1, dtype=np.uint64) * np.uint64(height * rowbytes)
    geom = np.zeros((n, 4), dtype=np.uint32)
    geom[:, :3] = (height, rowbytes, bpp)
    b = eng.device_batch(np.zeros(so[-1], dtype=np.uint8), np.array(so, dtype=np.uint64), do)
    arrs = [np.frombuffer(bytes(blob), dtype=np.uint8), np.array(po, dtype=np.uint64), np.array(pl, dtype=np.uint32),
            np.array(pd, dtype=np.uint64), geom, oo]
    dev = [lib.infl_gpu_malloc(a.nbytes + 16) for a in arrs] + [lib.infl_gpu_malloc(int(oo[-1]) + 16), lib.infl_gpu_malloc(4 * n)]
    if not all(dev):
        raise SystemExit("infl_gpu_malloc failed")
    for d, a in zip(dev, arrs):
        assert lib.infl_gpu_upload(d, a.ctypes.data, a.nbytes) == 0
    d_blob, d_po, d_pl, d_pd, d_geom, d_oo, d_out, d_ust = dev
    stream = torch.cuda.current_stream().cuda_stream

    def gather():
        assert lib.infl_gather_dev_checked(d_blob, len(blob), d_po, d_pl, d_pd, len(po), b.d_src, so[-1], None, stream) == 0

    def decode():
        assert lib.infl_batch_dev(b.d_src, b.d_soff, b.d_dst, b.d_doff, 1, n, b.d_status, b.d_outlen, stream) == 0

    def unfilter():
        assert lib.infl_png_unfilter_dev_checked(b.d_dst, int(do[-1]), b.d_doff, b.d_outlen, b.d_status, d_out, int(oo[-1]),
                                                 d_oo, d_geom, n, d_ust, stream) == 0

    times = {}
    for name, fns in (("gather", (gather,)), ("decode", (decode,)), ("unfilter", (unfilter,)),
                      ("gather_decode_unfilter", (gather, decode, unfilter))):
        for f in fns:
            f()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a, e in ev:
            a.record()
            for f in fns:
                f()
            e.record()
        torch.cuda.synchronize()
        times[name] = float(np.median([a.elapsed_time(e) for a, e in ev]))
    dst, status, outlen, _ = b.results()
    pix = np.zeros(int(oo[-1]), dtype=np.uint8)
    ust = np.zeros(n, dtype=np.int32)
    assert lib.infl_gpu_download(pix.ctypes.data, d_out, pix.nbytes) == 0
    assert lib.infl_gpu_download(ust.ctypes.data, d_ust, ust.nbytes) == 0
    want_pix = [np.frombuffer(r, dtype=np.uint8).reshape(height, rowbytes + 1)[:, 1:].tobytes() for r in raws]   # filter type 0
    ok = bool((status == 0).all()) and bool((ust == 0).all())
    for i in range(n):
        if not ok:
            break
        a, o = int(do[i]), int(oo[i])
        ok = dst[a:a + caps[i]].tobytes() == raws[i % distinct] and pix[o:o + height * rowbytes].tobytes() == want_pix[i % distinct]
    for d in dev:
        lib.infl_gpu_free(d)
    b.free()
    if not ok:
        raise SystemExit("parity gate failed: PNG IDAT on the device")
    out_b, in_b = int(do[-1]), int(so[-1])
    algo = (in_b + out_b) / (times["decode"] * 1e-3) / 1e9
    return {"config": "PNG IDAT: %d images uploaded whole; infl_gather_dev_checked, infl_batch_dev, "
                      "infl_png_unfilter_dev_checked on the device" % n,
            "streams": n, "pieces": len(po), "out_bytes": out_b, "in_bytes": in_b,
            "gather_ms": round(times["gather"], 3), "kernel_ms": round(times["decode"], 3),
            "unfilter_ms": round(times["unfilter"], 3), "all_three_ms": round(times["gather_decode_unfilter"], 3),
            "gather_GB_per_s_read_plus_written": round(2 * in_b / (times["gather"] * 1e-3) / 1e9, 1),
            "decompressed_MB_per_s": round(out_b / times["decode"] / 1e3, 1), "algorithmic_GB_per_s": round(algo, 1),
            "roofline_frac": round(algo / HBM_PEAK_GBS, 4),
            "unfilter_GB_per_s_read_plus_written": round((out_b + int(oo[-1])) / (times["unfilter"] * 1e-3) / 1e9, 1),
            "parity_gate": "every image: decoded scanlines and rebuilt pixels byte for byte"}


def secondary_stream_bytewise(eng):
    """The reference's byte-wise streaming test as a timing (test/test_files.c:792-869): png_simulation fed to
    infl_stream() one byte per call."""
    lib = eng.lib
    data = os.path.join(ROOT, "tests", "golden", "defl_testdata")
    raw = open(os.path.join(data, "raw", "png_simulation"), "rb").read()
    comp = open(os.path.join(data, "compressed", "png_simulation"), "rb").read()
    out = (C.c_uint8 * (len(raw) + 1000))()
    src = (C.c_uint8 * len(comp)).from_buffer_copy(comp)
    base = C.addressof(src)
    st = lib.infl_init(out, len(raw) + 1000, 0)
    t0 = time.perf_counter()
    rc, pos = 2, 0
    while pos < len(comp) and rc == 2:
        rc = lib.infl_stream(st, base + pos, 1)
        pos += 1
    took = time.perf_counter() - t0
    ok = rc == 0 and bytes(out[:len(raw)]) == raw
    lib.infl_destroy(st)
    if not ok:
        raise SystemExit("parity gate failed: byte-wise infl_stream")
    return {"config": "infl_stream fed one byte per call (the reference's png_simulation test: %d calls)" % len(comp),
            "calls": pos, "seconds": round(took, 3), "us_per_call": round(took / pos * 1e6, 2), "out_bytes": len(raw),
            "parity_gate": "the whole output byte for byte"}


def run_secondary(eng, torch, dev, oracle, threads, steps):
    import zlib
    out = []

    def add(what, fn):
        note("secondary: " + what)
        t0 = time.perf_counter()
        r = fn()
        r["wall_s"] = round(time.perf_counter() - t0, 1)
        out.append(r)

    def stored():
        pairs = pool_map(lambda i: wl.stored_stream(i, STREAM_BYTES), range(256), threads)
        return secondary_device(eng, torch, dev, oracle, threads, "stored-only: 65 536 x 64 KiB incompressible payload as "
                                "stored blocks + zlib wrapper", [p[1] for p in pairs], [STREAM_BYTES] * 256, 65536, steps)

    def fixed():
        comps = pool_map(lambda i: wl.deflate(wl.text_stream(i), 6, True, zlib.Z_FIXED), range(512), threads)
        return secondary_device(eng, torch, dev, oracle, threads, "fixed-Huffman-only: 65 536 x 64 KiB text streams "
                                "(deflateInit2 Z_FIXED)", comps, [STREAM_BYTES] * 512, 65536, steps)

    add("stored-only batch", stored)
    add("fixed-Huffman-only batch", fixed)
    n_png = 4096
    raws, comps, pieces = png_workload(n_png)
    add("PNG IDAT through infl_include + infl_many", lambda: secondary_png_host(eng, n_png, raws, comps, pieces))
    add("PNG IDAT gathered, decoded and un-filtered on the device",
        lambda: secondary_png_device(eng, torch, n_png, raws, comps, pieces, steps))
    pool = 2048
    sizes = pool_sizes("mixed", pool)
    mcomps = build_pool(1000, sizes, threads)
    reps = FULL_STREAMS["mixed"] // pool
    by_size = np.argsort(-np.array(sizes), kind="stable")
    add("mixed sizes, pool order", lambda: secondary_device(
        eng, torch, dev, oracle, threads, "mixed sizes 1 KiB .. 256 KiB (log-uniform), 1 048 576 streams in pool order "
        "(sizes in no order)", mcomps, sizes, pool * reps, steps))
    add("mixed sizes, pool order, size classes made on the device", lambda: secondary_device(
        eng, torch, dev, oracle, threads, "mixed sizes 1 KiB .. 256 KiB (log-uniform), 1 048 576 streams in pool order through "
        "infl_batch_dev_sorted (size classes made on the device)", mcomps, sizes, pool * reps, steps, sort=True))
    add("mixed sizes, by falling size", lambda: secondary_device(
        eng, torch, dev, oracle, threads, "mixed sizes 1 KiB .. 256 KiB (log-uniform), 1 048 576 streams by falling size",
        mcomps, sizes, pool * reps, steps, by_size, reps))
    add("byte-wise infl_stream", lambda: secondary_stream_bytewise(eng))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("DEFL_BENCH_WORKLOAD", "text64k"), choices=sorted(FULL_STREAMS))
    ap.add_argument("--scaling", default=os.environ.get("DEFL_BENCH_SCALING", "weak"), choices=["weak", "strong"])
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DEFL_BENCH_STREAMS", 0)),
                    help="streams of the full batch (default: the workload's size); weak scaling gives every "
                         "GPU such a batch, strong scaling splits one")
    ap.add_argument("--pool", type=int, default=int(os.environ.get("DEFL_BENCH_POOL", 0)),
                    help="distinct streams generated (default 8192, mixed sizes 2048); the batch draws its streams "
                         "from them into distinct source and destination slots")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("DEFL_BENCH_LANES", 16)))
    ap.add_argument("--variant", type=int, default=int(os.environ.get("DEFL_BENCH_VARIANT", 2)),
                    help="0 = tables in the reference geometry, 1 = compact tables + deferred pass, "
                         "2 = wavefront-parallel block decode + deferred pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="the headline only (one GPU runs the other "
                    "configurations of the measurement plan behind it by default)")
    args = ap.parse_args(argv)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(relaunch(args, sys.argv[1:] if argv is None else list(argv)))

    # Exactly one line goes to stdout: the result.  Whatever a library prints there (the gloo transport announces
    # its connections on stdout, from C++) is sent to stderr by pointing descriptor 1 at it; the result line is
    # written to a private copy of the original descriptor.
    global RESULT_OUT
    sys.stdout.flush()
    RESULT_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE=%d" % (args.gpus, world))
    stub = os.environ.get("DEFL_BENCH_STUB") == "1"     # rehearsal on CPU tensors (the multi-process CPU test)
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # Rehearsal aids for a box with fewer GPUs than ranks (the measured runs use neither): all ranks on
    # device 0, and gloo (CPU tensors only) instead of RCCL, which refuses two ranks on one device.
    one_gpu = os.environ.get("DEFL_BENCH_ONE_GPU") == "1"
    backend = "gloo" if stub else os.environ.get("DEFL_BENCH_BACKEND", "nccl")
    if one_gpu:
        local = 0
    if stub:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    oracle = None
    if stub:
        from tests import oracle_lib
        oracle = oracle_lib.load()
        eng, lib = None, OracleEngine(oracle)
        note("REHEARSAL: the decode runs on the CPU oracle; nothing here is a measurement")
    else:
        from defl_amd import capi
        note("torch ready, loading engine")
        eng = capi.load()
        lib = eng.lib
        if lib.infl_gpu_init(local) != 0:
            raise SystemExit("infl_gpu_init failed")
        lib.infl_gpu_set_lanes_per_stream(args.lanes)
        lib.infl_gpu_set_table_variant(args.variant)

    full = args.streams or FULL_STREAMS[args.workload]
    pool = max(1, min(args.pool or (8192 if args.workload == "text64k" else 2048), full))
    threads = host_threads()
    workers = max(1, threads // max(1, local_world))    # the node's ranks generate at the same time
    note("generating %d distinct streams with %d threads" % (pool, workers))
    t_gen = time.perf_counter()
    sizes = pool_sizes(args.workload, pool)
    # weak scaling: every rank has its own distinct streams; strong scaling: one batch for all
    comps = build_pool(rank * pool if args.scaling == "weak" else 0, sizes, workers)
    t_gen = time.perf_counter() - t_gen

    # The batch is in pool order (stream g is pool stream g % pool): for the mixed workload that means sizes in no
    # order, and the decode goes through infl_batch_dev_sorted, which puts the rank's streams into size classes on the
    # device first; the rate of the plain call on the same batch is measured beside it.
    clen = np.array([len(c) for c in comps], dtype=np.float64)
    if args.scaling == "weak":
        lo, hi = 0, full
        cost_share = None
    else:
        cost = (clen + np.array(sizes, dtype=np.float64))[np.arange(full) % pool]
        parts = shard.partition(cost, world)
        lo, hi = parts[rank]
        cost_share = [float(cost[a:b].sum()) for a, b in parts]
    note("uploading streams [%d, %d)" % (lo, hi))
    b = Batch(torch, dev, comps, sizes, lo, hi)
    sort = args.workload == "mixed" and not stub
    work = torch.empty(1024 + 4 * max(1, b.n) + 16, dtype=torch.uint8, device=dev) if sort else None

    def barrier():
        if world > 1:
            dist.barrier()

    note("warm-up, then timing %d steps" % args.steps)
    elapsed, kern_ms = time_steps(torch, make_step(lib, torch, b, 1, stub, work), args.steps, args.warmup, barrier, stub)

    # parity gate, on what the timed steps left in memory: every stream of the rank against the oracle (the checker
    # is loaded here, behind the timed region)
    note("parity gate")
    if oracle is None:
        from tests import oracle_lib
        oracle = oracle_lib.load()
    ok, n_oracle, n_checked = b.verify(oracle, comps, sizes, threads)

    n_ok, n_err, bytes_ok = shard.summarize(b.d_status[: b.n], b.d_outlen[: b.n], dist if world > 1 else None, device=dev)
    unsorted_ms = 0.0
    if sort:                                                # the same batch through the plain call (no size classes)
        _, unsorted_ms = time_steps(torch, make_step(lib, torch, b, 1, stub), args.steps, 1, barrier, stub)
        ok2, _, n2 = b.verify(oracle, comps, sizes, threads)
        ok = ok and ok2 and n2 == n_checked
    per = torch.tensor([elapsed, float(ok), float(n_checked), float(b.comp_bytes + b.out_bytes), kern_ms, unsorted_ms],
                       dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        allr = [torch.zeros_like(per) for _ in range(world)]
        dist.all_gather(allr, per)
        allr = torch.stack(allr).cpu().numpy()
    else:
        allr = per.cpu().numpy()[None, :]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    elapsed = float(allr[:, 0].max())
    all_ok = bool((allr[:, 1] == 1.0).all()) and n_err == 0
    if not all_ok:
        raise SystemExit("parity gate failed: %d streams in error, per-rank check %s" % (n_err, allr[:, 1].tolist()))
    if rank != 0:
        return

    total_streams = n_ok
    value = bytes_ok * args.steps / elapsed / 1e6
    achieved = (b.comp_bytes + b.out_bytes) / (kern_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            if (t.get("streams") == b.n and t.get("workload", "text64k") == args.workload
                    and t.get("variant", args.variant) == args.variant):
                traffic = t.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    names = {"text64k": "64 KiB zlib level-6 streams", "mixed": "zlib level-6 streams of 1 KiB .. 256 KiB (log-uniform, sizes in no order)"}
    load = allr[:, 3]
    res = {
        "metric": "decompressed_MB_per_s",
        "value": round(value, 1),
        "unit": "MB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": "batch of %d independent %s %s (dynamic Huffman, synthetic Zipf text), one launch per "
                        "GPU, inputs and outputs resident in HBM"
                        % (full, names[args.workload], "per GPU" if args.scaling == "weak" else "split over the GPUs"),
            "streams_total": total_streams,
            "streams_rank0": b.n,
            "distinct_streams": pool if args.scaling == "strong" else pool * world,
            "compressed_ratio": round(b.comp_bytes / max(1, b.out_bytes), 4),
            "lanes_per_stream": args.lanes,
            "table_variant": args.variant,
            "parallelism": "stream-shard x%d, no data-path collective" % world,
            "bit_exact_vs_oracle": all_ok,
            "parity_gate": "after the timed steps: every distinct stream of a rank against the CPU oracle byte for "
                           "byte (%d on rank 0), every further occurrence against that one on the device (%d streams on "
                           "rank 0, %d over all ranks)" % (n_oracle, n_checked, int(allr[:, 2].sum())),
            "per_rank_ms_per_step": [round(x / args.steps * 1e3, 3) for x in allr[:, 0].tolist()],
            "per_rank_load_imbalance": round(float(load.max() / load.mean()), 4),
            "pool_generation_s": round(t_gen, 1),
            "pool_generation_threads": workers,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            "kernel": "infl_batch_kernel",
            "kernel_ms": round(kern_ms, 4),
            "algorithmic_bytes_per_launch": b.comp_bytes + b.out_bytes,
        },
    }
    if stub:
        res["config"]["REHEARSAL"] = "decoded by the CPU oracle on CPU tensors: not a measurement"
    if cost_share is not None:
        res["config"]["partition_cost_share"] = [round(c / max(1.0, sum(cost_share)), 4) for c in cost_share]
    if sort:
        res["config"]["entry_point"] = "infl_batch_dev_sorted: every rank's streams put into size classes on the device first"
        res["config"]["plain_infl_batch_dev_ms_per_step_max_over_ranks"] = round(float(allr[:, 5].max()), 3)
        res["config"]["plain_infl_batch_dev_MB_per_s"] = round(bytes_ok / (float(allr[:, 5].max()) * 1e-3) / 1e6, 1)
    if world == 1 and not stub and not args.no_secondary and args.workload == "text64k":
        del b
        torch.cuda.empty_cache()
        res["secondary"] = run_secondary(eng, torch, dev, oracle, threads, 5)
    if not args.no_cpu_baseline:
        note("cpu baseline on %d threads" % threads)
        k = min(pool, 1024)
        res["cpu_baseline"] = cpu_baseline(comps[:k], sizes[:k], oracle, threads, 12.0 if not stub else 0.5)
    RESULT_OUT.write(json.dumps(res) + "\n")
    RESULT_OUT.flush()


if __name__ == "__main__":
    main()
