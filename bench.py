#!/usr/bin/env python3
"""Headline benchmark: decompressed MB/s of a batch of independent zlib streams on N x MI355X.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--workload text64k|mixed]

Workloads (synthetic Zipf text, zlib level 6, dynamic Huffman):
  text64k  65 536 streams of 64 KiB (the configuration the metric is quoted on);
  mixed    1 048 576 streams of 1 KiB .. 256 KiB, log-uniform (the sharded mixed-size configuration).
Scaling:
  weak     every GPU decodes a batch of the full size (default: per-GPU work fixed as N grows);
  strong   ONE batch of the full size is cut into N contiguous ranges that balance the sum of
           compressed + decompressed bytes (defl_amd.shard.partition), one range per GPU.
One process per GPU, no data-path collective: streams are independent.  RCCL carries only the
result counters and the per-rank times.  Inputs and outputs stay in HBM; a step is one pass of the
hot path (one infl_batch_dev call) over the rank's streams.  Started without a launcher and with
--gpus N > 1, the script starts `torch.distributed.run` itself (before it touches any GPU) and
relays the result.  Prints ONE JSON line on rank 0.
"""
import argparse
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


def build_pool(first, sizes, workers):
    """Threads, not processes: a fork after the HIP runtime is up (a profiler preloads it before
    main) can deadlock the children.  zlib and numpy release the GIL for most of the work."""
    jobs = [(first + k, s) for k, s in enumerate(sizes)]
    if workers > 1:
        with ThreadPoolExecutor(workers) as ex:
            return list(ex.map(_make_stream, jobs))
    return [_make_stream(j) for j in jobs]


def host_threads():
    """Threads the CPU baseline may use: the cores this process is allowed on, capped by the
    share the machine gives the job (OMP_NUM_THREADS, where the operator sets it)."""
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
    the value is the median pass.  One thread and the system's zlib are timed beside it."""
    src, src_off, dst_off = wl.pack(comps, sizes)
    oracle.infl_batch(src, src_off, dst_off, 1, threads)            # warm-up (tables, pages)
    times, out_bytes = [], 0
    while (sum(times) * threads < budget_s or len(times) < 3) and len(times) < 64:
        t0 = time.perf_counter()
        _, status, outlen = oracle.infl_batch(src, src_off, dst_off, 1, threads)
        times.append(time.perf_counter() - t0)
        out_bytes = int(outlen.sum())
        assert (status == 0).all()
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
    """This rank's streams: global stream g of the batch is pool stream g % pool; the rank owns the
    contiguous range [lo, hi).  Device layout: one source arena + u64 offsets, one destination arena +
    u64 offsets (capacity = the stream's decompressed size), status and length arrays."""

    def __init__(self, torch, dev, comps, sizes, lo, hi):
        self.torch, self.dev, self.lo, self.hi = torch, dev, lo, hi
        pool = len(comps)
        self.pool, self.n = pool, hi - lo
        clen = np.array([len(c) for c in comps], dtype=np.uint64)
        ulen = np.array(sizes, dtype=np.uint64)
        idx = np.arange(lo, hi, dtype=np.int64) % pool
        self.src_off = np.zeros(self.n + 1, dtype=np.uint64)
        self.dst_off = np.zeros(self.n + 1, dtype=np.uint64)
        np.cumsum(clen[idx], out=self.src_off[1:])
        np.cumsum(ulen[idx], out=self.dst_off[1:])
        self.comp_bytes, self.out_bytes = int(self.src_off[-1]), int(self.dst_off[-1])
        # source arena: the pool block tiled on the device, cut to the rank's range
        blk = torch.from_numpy(np.frombuffer(b"".join(comps), dtype=np.uint8).copy()).to(dev)
        cs = np.concatenate([[0], np.cumsum(clen)]).astype(np.int64)       # pool prefix sums
        self.us = np.concatenate([[0], np.cumsum(ulen)]).astype(np.int64)
        b0, b1 = lo // pool, (hi + pool - 1) // pool                       # blocks touched
        tiled = blk.repeat(b1 - b0)
        start = int(cs[lo - b0 * pool])
        self.d_src = torch.cat([tiled[start:start + self.comp_bytes], torch.zeros(64, dtype=torch.uint8, device=dev)])
        del tiled, blk
        self.d_dst = torch.empty(self.out_bytes + 64, dtype=torch.uint8, device=dev)
        self.d_soff = torch.from_numpy(self.src_off.view(np.int64)).to(dev)
        self.d_doff = torch.from_numpy(self.dst_off.view(np.int64)).to(dev)
        self.d_status = torch.full((max(1, self.n),), -99, dtype=torch.int32, device=dev)
        self.d_outlen = torch.zeros(max(1, self.n), dtype=torch.int32, device=dev)
        self.want_len = torch.from_numpy(ulen[idx].astype(np.int32)).to(dev)

    def verify(self, oracle, comps, sizes, threads):
        """Every output of the rank against the CPU oracle: the first occurrence of every distinct
        stream byte for byte on the host, every further occurrence byte for byte against that first
        one on the device.  Returns (streams checked against the oracle, streams checked in all)."""
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
        # the rank's output is a run of (partial first block, whole blocks, partial last block)
        g, at, checked = lo, 0, 0
        while g < hi and ok:
            j = g % pool
            cnt = min(pool - j, hi - g)                                     # streams up to the block's end
            a, b = int(self.us[j]), int(self.us[j + cnt])
            ok = torch.equal(self.d_dst[at:at + (b - a)], d_want[a:b])
            at += b - a
            g += cnt
            checked += cnt
        return ok, min(pool, self.n), checked


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("DEFL_BENCH_WORKLOAD", "text64k"), choices=sorted(FULL_STREAMS))
    ap.add_argument("--scaling", default=os.environ.get("DEFL_BENCH_SCALING", "weak"), choices=["weak", "strong"])
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DEFL_BENCH_STREAMS", 0)),
                    help="streams of the full batch (default: the workload's size); weak scaling gives every "
                         "GPU such a batch, strong scaling splits one")
    ap.add_argument("--pool", type=int, default=int(os.environ.get("DEFL_BENCH_POOL", 8192)),
                    help="distinct streams generated; the batch tiles them into distinct source and "
                         "destination slots")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("DEFL_BENCH_LANES", 16)))
    ap.add_argument("--variant", type=int, default=int(os.environ.get("DEFL_BENCH_VARIANT", 2)),
                    help="0 = tables in the reference geometry, 1 = compact tables + deferred pass, "
                         "2 = wavefront-parallel block decode + deferred pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(relaunch(args, sys.argv[1:]))

    # Exactly one line goes to stdout: the result.  Whatever a library prints there (the gloo transport announces
    # its connections on stdout, from C++) is sent to stderr by pointing descriptor 1 at it; the result line is
    # written to a private copy of the original descriptor.
    global RESULT_OUT
    sys.stdout.flush()
    RESULT_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from defl_amd import capi

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # Rehearsal aids for a box with fewer GPUs than ranks (the measured runs use neither): all ranks on
    # device 0, and gloo (CPU tensors only) instead of RCCL, which refuses two ranks on one device.
    one_gpu = os.environ.get("DEFL_BENCH_ONE_GPU") == "1"
    backend = os.environ.get("DEFL_BENCH_BACKEND", "nccl")
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    note("torch ready, loading engine")
    eng = capi.load()
    if eng.lib.infl_gpu_init(local) != 0:
        raise SystemExit("infl_gpu_init failed")
    eng.lib.infl_gpu_set_lanes_per_stream(args.lanes)
    eng.lib.infl_gpu_set_table_variant(args.variant)

    full = args.streams or FULL_STREAMS[args.workload]
    pool = max(1, min(args.pool, full))
    threads = host_threads()
    workers = max(1, min(threads, 16) // max(1, min(world, 8)))
    note("generating %d distinct streams with %d threads" % (pool, workers))
    sizes = pool_sizes(args.workload, pool)
    # weak scaling: every rank has its own distinct streams; strong scaling: one batch for all
    comps = build_pool(rank * pool if args.scaling == "weak" else 0, sizes, workers)

    if args.scaling == "weak":
        lo, hi = 0, full
        cost_share = None
    else:
        clen = np.array([len(c) for c in comps], dtype=np.float64)
        cost = (clen + np.array(sizes, dtype=np.float64))[np.arange(full) % pool]
        parts = shard.partition(cost, world)
        lo, hi = parts[rank]
        cost_share = [float(cost[a:b].sum()) for a, b in parts]
    note("uploading streams [%d, %d)" % (lo, hi))
    b = Batch(torch, dev, comps, sizes, lo, hi)

    def step():
        stream = torch.cuda.current_stream().cuda_stream
        rc = eng.lib.infl_batch_dev(b.d_src.data_ptr(), b.d_soff.data_ptr(), b.d_dst.data_ptr(), b.d_doff.data_ptr(),
                                    1, b.n, b.d_status.data_ptr(), b.d_outlen.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError("infl_batch_dev failed")

    def barrier():
        if world > 1:
            dist.barrier()

    note("warm-up")
    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    note("timing %d steps" % args.steps)
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        step()
        ev[k][1].record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([x.elapsed_time(y) for x, y in ev]))

    # parity gate, on what the timed steps left in memory: every stream of the rank against the oracle
    note("parity gate")
    from tests import oracle_lib
    oracle = oracle_lib.load()
    ok, n_oracle, n_checked = b.verify(oracle, comps, sizes, threads)

    n_ok, n_err, bytes_ok = shard.summarize(b.d_status[: b.n], b.d_outlen[: b.n], dist if world > 1 else None, device=dev)
    per = torch.tensor([elapsed, float(ok), float(n_checked), float(b.comp_bytes + b.out_bytes), kern_ms],
                       dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        allr = [torch.zeros_like(per) for _ in range(world)]
        dist.all_gather(allr, per)
        allr = torch.stack(allr).cpu().numpy()
    else:
        allr = per.cpu().numpy()[None, :]
    elapsed = float(allr[:, 0].max())
    all_ok = bool((allr[:, 1] == 1.0).all()) and n_err == 0
    if not all_ok:
        raise SystemExit("parity gate failed: %d streams in error, per-rank check %s" % (n_err, allr[:, 1].tolist()))

    if rank == 0:
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
        names = {"text64k": "64 KiB zlib level-6 streams", "mixed": "zlib level-6 streams of 1 KiB .. 256 KiB (log-uniform)"}
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
                               "byte (%d on rank 0), every tile of it against that block on the device (%d streams on "
                               "rank 0, %d over all ranks)" % (n_oracle, n_checked, int(allr[:, 2].sum())),
                "per_rank_ms_per_step": [round(x / args.steps * 1e3, 3) for x in allr[:, 0].tolist()],
                "per_rank_load_imbalance": round(float(load.max() / load.mean()), 4),
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
        if cost_share is not None:
            res["config"]["partition_cost_share"] = [round(c / max(1.0, sum(cost_share)), 4) for c in cost_share]
        if not args.no_cpu_baseline and world == 1:
            note("cpu baseline on %d threads" % threads)
            k = min(pool, 1024)
            res["cpu_baseline"] = cpu_baseline(comps[:k], sizes[:k], oracle, threads, 12.0)
        RESULT_OUT.write(json.dumps(res) + "\n")
        RESULT_OUT.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
