#!/usr/bin/env python3
"""Headline benchmark: decompressed MB/s of a batch of 65 536 independent 64 KiB zlib streams
(dynamic Huffman, level 6, synthetic English-like text) on N x MI355X, one launch per step.

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  Streams are independent, so each rank decodes its own 65 536-stream
shard with no data-path collective (weak scaling); RCCL only sums the per-rank result
counters.  Inputs and outputs stay in HBM; a step is one pass of the hot path (one
infl_batch_dev launch) over the rank's batch.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
from concurrent.futures import ThreadPoolExecutor
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import workloads as wl  # noqa: E402

STREAM_BYTES = 65536
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (spec), GB/s


def _make_stream(i):
    raw = wl.text_stream(i, STREAM_BYTES)
    return wl.deflate(raw, 6, True)


def build_pool(first, count, workers):
    """Threads, not processes: a fork after the HIP runtime is up (a profiler preloads it before
    main) can deadlock the children.  zlib and numpy release the GIL for most of the work."""
    idx = list(range(first, first + count))
    if workers > 1:
        with ThreadPoolExecutor(workers) as pool:
            return list(pool.map(_make_stream, idx))
    return [_make_stream(i) for i in idx]


def cpu_baseline(comps, oracle, threads, budget_s):
    """The CPU oracle (a port of the reference's infl_buf path) timed on the host cores over
    the same streams: `threads` threads with static partitioning, repeated until about
    `budget_s` seconds of CPU-core time are spent."""
    src, src_off, dst_off = wl.pack(comps, [STREAM_BYTES] * len(comps))
    oracle.infl_batch(src, src_off, dst_off, 1, threads)            # warm-up (tables, pages)
    reps, t_total, out_bytes = 0, 0.0, 0
    while t_total * threads < budget_s and reps < 64:
        t0 = time.perf_counter()
        _, status, outlen = oracle.infl_batch(src, src_off, dst_off, 1, threads)
        t_total += time.perf_counter() - t0
        out_bytes += int(outlen.sum())
        reps += 1
        assert (status == 0).all()
    t0 = time.perf_counter()
    sub = max(1, len(comps) // 16)
    s1, so1, do1 = wl.pack(comps[:sub], [STREAM_BYTES] * sub)
    _, _, ol1 = oracle.infl_batch(s1, so1, do1, 1, 1)
    t1 = time.perf_counter() - t0
    import zlib
    t0 = time.perf_counter()
    zbytes = sum(len(zlib.decompress(c)) for c in comps[:sub])
    tz = time.perf_counter() - t0
    return {
        "value": round(out_bytes / t_total / 1e6, 1),
        "unit": "MB/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d distinct streams of this workload x %d passes, %d threads; 1 thread: %.0f MB/s "
                  "(system zlib, 1 thread: %.0f MB/s)"
                  % (len(comps), reps, threads, int(ol1.sum()) / t1 / 1e6, zbytes / tz / 1e6),
    }


def note(msg):
    """Progress on stderr (stdout carries only the JSON line)."""
    if os.environ.get("RANK", "0") == "0":
        print("[bench %s] %s" % (time.strftime("%H:%M:%S"), msg), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DEFL_BENCH_STREAMS", 65536)),
                    help="streams per GPU")
    ap.add_argument("--pool", type=int, default=int(os.environ.get("DEFL_BENCH_POOL", 2048)),
                    help="distinct streams generated per GPU; the batch tiles them into distinct "
                         "source and destination slots")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("DEFL_BENCH_LANES", 16)))
    ap.add_argument("--variant", type=int, default=int(os.environ.get("DEFL_BENCH_VARIANT", 2)),
                    help="0 = tables in the reference geometry, 1 = compact tables + deferred pass, "
                         "2 = wavefront-parallel block decode + deferred pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from defl_amd import capi

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    note("torch ready, loading engine")
    eng = capi.load()
    if eng.lib.infl_gpu_init(local) != 0:
        raise SystemExit("infl_gpu_init failed")
    eng.lib.infl_gpu_set_lanes_per_stream(args.lanes)
    eng.lib.infl_gpu_set_table_variant(args.variant)

    n, pool = args.streams, min(args.pool, args.streams)
    workers = 4
    note("generating %d streams with %d workers" % (pool, workers))
    comps = build_pool(rank * pool, pool, workers)
    note("uploading")

    # device layout: one contiguous src arena + u64 offsets, dst arena with 64 KiB stride
    reps = (n + pool - 1) // pool
    lens = np.array([len(c) for c in comps], dtype=np.uint64)
    lens_all = np.tile(lens, reps)[:n]
    src_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens_all, out=src_off[1:])
    dst_off = np.arange(n + 1, dtype=np.uint64) * STREAM_BYTES
    pool_bytes = torch.from_numpy(np.frombuffer(b"".join(comps), dtype=np.uint8).copy()).to(dev)
    d_src = pool_bytes.repeat(reps)[: int(src_off[-1])].contiguous()
    d_src = torch.cat([d_src, torch.zeros(64, dtype=torch.uint8, device=dev)])
    d_dst = torch.empty(n * STREAM_BYTES + 64, dtype=torch.uint8, device=dev)
    d_soff = torch.from_numpy(src_off.view(np.int64)).to(dev)
    d_doff = torch.from_numpy(dst_off.view(np.int64)).to(dev)
    d_status = torch.full((n,), -99, dtype=torch.int32, device=dev)
    d_outlen = torch.zeros(n, dtype=torch.int32, device=dev)
    comp_bytes = int(src_off[-1])
    out_bytes = n * STREAM_BYTES

    def step():
        stream = torch.cuda.current_stream().cuda_stream
        rc = eng.lib.infl_batch_dev(d_src.data_ptr(), d_soff.data_ptr(), d_dst.data_ptr(), d_doff.data_ptr(),
                                    1, n, d_status.data_ptr(), d_outlen.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError("infl_batch_dev failed")

    def barrier():
        if world > 1:
            dist.barrier()

    note("warm-up")
    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    note("parity gate")

    # parity gate: every stream complete, and the first `pool` outputs bit-exact vs the oracle
    from tests import oracle_lib
    oracle = oracle_lib.load()
    assert bool((d_status == 0).all()) and bool((d_outlen == STREAM_BYTES).all()), "decode failed"
    chk = min(pool, 256)
    s_np, so_np, do_np = wl.pack(comps[:chk], [STREAM_BYTES] * chk)
    want, wst, wol = oracle.infl_batch(s_np, so_np, do_np, 1, 8)
    got = d_dst[: chk * STREAM_BYTES].cpu().numpy()
    assert np.array_equal(got, want[: chk * STREAM_BYTES]), "output differs from the CPU oracle"
    # all tiles must equal the pool outputs (64-bit sums per stream as a checksum of checksums)
    sums = d_dst[: n * STREAM_BYTES].view(n, STREAM_BYTES).to(torch.int64).sum(dim=1) if n <= 8192 else None
    if sums is not None:
        assert bool((sums.view(-1)[:pool * (n // pool)].view(-1, pool) == sums[:pool]).all())

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
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    tot = torch.tensor([elapsed, float(bool((d_status == 0).all())), float(out_bytes)],
                       dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        all_ok = float(tot[1]) == world
        total_out = float(tot[2])
    else:
        all_ok, total_out = bool(tot[1]), float(out_bytes)
    assert all_ok

    if rank == 0:
        value = total_out * args.steps / elapsed / 1e6
        achieved = (comp_bytes + out_bytes) / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                t = json.load(open(tpath))
                if t.get("streams") == n and t.get("variant", args.variant) == args.variant:
                    traffic = t.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "decompressed_MB_per_s",
            "value": round(value, 1),
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "batch of %d independent 64 KiB zlib level-6 streams per GPU (dynamic Huffman, "
                            "synthetic Zipf text), one launch, inputs and outputs resident in HBM" % n,
                "streams_per_gpu": n,
                "distinct_streams_per_gpu": pool,
                "stream_bytes": STREAM_BYTES,
                "compressed_ratio": round(comp_bytes / out_bytes, 4),
                "lanes_per_stream": args.lanes,
                "table_variant": args.variant,
                "parallelism": "stream-shard x%d, no data-path collective" % world,
                "bit_exact_vs_oracle": True,
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
                "algorithmic_bytes_per_launch": comp_bytes + out_bytes,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            note("cpu baseline")
            threads = max(1, min(16, os.cpu_count() or 1))
            res["cpu_baseline"] = cpu_baseline(comps[: min(pool, 1024)], oracle, threads, 12.0)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
