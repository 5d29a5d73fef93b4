"""Multi-process (world_size 2, gloo, CPU) coverage of the N>1 path: stream partitioning and
the result-counter reduction.  The decode itself needs a GPU; here each rank 'decodes' its
shard with the CPU oracle, which is what the reduction is agnostic to."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

import workloads as wl
from defl_amd import shard


def test_partition_balances_and_covers():
    rng = np.random.default_rng(1)
    cost = rng.integers(1000, 300000, size=5000)
    for world in (1, 2, 3, 8):
        parts = shard.partition(cost, world)
        assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == len(cost)
        for (a, b), (c, d) in zip(parts, parts[1:]):
            assert b == c and a <= b
        loads = [cost[a:b].sum() for a, b in parts]
        assert max(loads) <= 1.05 * cost.sum() / world + cost.max()
    assert shard.partition([], 4) == [(0, 0)] * 4
    assert shard.partition([5], 2) in ([(0, 0), (0, 1)], [(0, 1), (1, 1)])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import oracle_lib
    oracle = oracle_lib.load()
    raws = [wl.text_stream(i, 3000 + 37 * i) for i in range(n)]
    comps = [wl.deflate(r) for r in raws]
    comps[3] = comps[3][:40]                       # one broken stream
    cost = [len(c) + len(r) for c, r in zip(comps, raws)]
    lo, hi = shard.partition(cost, world)[rank]
    src, so, do = wl.pack(comps[lo:hi], [len(r) for r in raws[lo:hi]])
    _, status, outlen = oracle.infl_batch(src, so, do, 1, 1)
    q.put((rank, shard.summarize(status, outlen, dist)))
    dist.destroy_process_group()


def _bench_worker(rank, world, port, q):
    """The data path of bench.py --scaling strong on CPU tensors: the same partition, the same Batch
    layout and parity gate, the same counter reduction; the decode itself is done by the oracle."""
    import torch
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import oracle_lib
    oracle = oracle_lib.load()
    pool, full = 24, 100                                   # 4 tiles of the pool and a partial fifth
    sizes = bench.pool_sizes("mixed", pool)
    sizes = [min(s, 20000) for s in sizes]
    comps = bench.build_pool(0, sizes, 1)
    cost = (np.array([len(c) for c in comps], dtype=np.float64) + np.array(sizes))[np.arange(full) % pool]
    lo, hi = shard.partition(cost, world)[rank]
    b = bench.Batch(torch, torch.device("cpu"), comps, sizes, lo, hi)
    dst, status, outlen = oracle.infl_batch(b.d_src.numpy(), b.src_off, b.dst_off, 1, 1)
    b.d_dst[: b.out_bytes] = torch.from_numpy(dst[: b.out_bytes])
    b.d_status[: b.n] = torch.from_numpy(status)
    b.d_outlen[: b.n] = torch.from_numpy(outlen.astype(np.int32))
    ok, n_oracle, n_checked = b.verify(oracle, comps, sizes, 1)
    # a wrong byte in the last stream of the rank must be seen
    b.d_dst[b.out_bytes - 1] ^= 1
    ok_bad, _, _ = b.verify(oracle, comps, sizes, 1)
    b.d_dst[b.out_bytes - 1] ^= 1
    tot = shard.summarize(b.d_status[: b.n], b.d_outlen[: b.n], dist)
    q.put((rank, lo, hi, ok, ok_bad, n_checked, tot, int(cost.sum()) - sum(len(comps[g % pool]) for g in range(full))))
    dist.destroy_process_group()


def _bench_main_worker(rank, world, port, outdir, extra):
    """bench.main() itself as rank `rank` of `world`, the way the launcher starts it (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in the environment), with the decode answered by the CPU oracle on CPU tensors (DEFL_BENCH_STUB=1: the
    rehearsal switch of bench.py) and gloo as the backend.  Descriptor 1 goes to a file: the result line."""
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "LOCAL_RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world), "DEFL_BENCH_STUB": "1",
                       "OMP_NUM_THREADS": "2"})
    fd = os.open(os.path.join(outdir, "rank%d.out" % rank), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    os.dup2(fd, 1)
    import bench
    bench.main(["--gpus", str(world), "--steps", "2", "--warmup", "1"] + extra)


def _run_bench_main(world, extra, tmp_path):
    import json
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_bench_main_worker, args=(r, world, port, str(tmp_path), extra)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    lines = [open(os.path.join(str(tmp_path), "rank%d.out" % r)).read().strip() for r in range(world)]
    assert all(not x for x in lines[1:])                       # only rank 0 prints
    assert len(lines[0].splitlines()) == 1                     # and exactly one line
    return json.loads(lines[0])


def test_world2_bench_main_end_to_end_weak(tmp_path):
    """The launch path of an N-GPU run executed end to end with world 2 on the CPU: argument handling, process group,
    barriers around the timed steps, parity gate, all_gather of the per-rank figures, one JSON line from rank 0 with
    the CPU baseline in it."""
    r = _run_bench_main(2, ["--streams", "96", "--pool", "24"], tmp_path)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["steps"] == 2
    assert r["config"]["streams_total"] == 192 and r["config"]["distinct_streams"] == 48
    assert r["config"]["bit_exact_vs_oracle"] is True and len(r["config"]["per_rank_ms_per_step"]) == 2
    assert "REHEARSAL" in r["config"] and r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["value"] > 0
    assert abs(r["value"] - 192 * 65536 * 2 / (r["ms_per_step"] * 2e-3) / 1e6) < 0.01 * r["value"]


def test_world2_bench_main_end_to_end_strong_mixed(tmp_path):
    """The same for the sharded mixed-size configuration: one batch ordered by falling size, cut by shard.partition."""
    r = _run_bench_main(2, ["--workload", "mixed", "--scaling", "strong", "--streams", "120", "--pool", "12",
                            "--no-cpu-baseline"], tmp_path)
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["config"]["streams_total"] == 120
    assert abs(sum(r["config"]["partition_cost_share"]) - 1.0) < 1e-3
    assert r["config"]["per_rank_load_imbalance"] < 1.5 and "cpu_baseline" not in r


def test_world2_bench_strong_scaling_path():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][1] == 0 and got[0][2] == got[1][1] and got[1][2] == 100      # the ranges cover the batch
    for rank, lo, hi, ok, ok_bad, n_checked, (n_ok, n_err, nbytes), want_bytes in got:
        assert ok and not ok_bad and n_checked == hi - lo
        assert (n_ok, n_err, nbytes) == (100, 0, want_bytes)


def test_world2_shard_and_reduce():
    n, world = 40, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    raws_total = sum(3000 + 37 * i for i in range(n)) - (3000 + 37 * 3)
    for _, (n_ok, n_err, nbytes) in got:
        assert (n_ok, n_err, nbytes) == (n - 1, 1, raws_total)
