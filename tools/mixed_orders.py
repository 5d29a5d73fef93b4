#!/usr/bin/env python3
"""Diagnostic: the mixed-size batch (1 KiB .. 256 KiB, log-uniform) decoded in several orders of the same streams, to
tell apart what a batch's order costs: pool order (sizes in no order), every distinct stream repeated in a row with the
groups by falling size, by rising size, in random order, and pool order over a pool sorted by size.

    python tools/mixed_orders.py [--pool 2048] [--reps 128]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=128)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    from defl_amd import capi
    eng = capi.load()
    assert eng.lib.infl_gpu_init(0) == 0
    dev = torch.device("cuda", 0)
    sizes = bench.pool_sizes("mixed", args.pool)
    comps = bench.build_pool(1000, sizes, bench.host_threads())
    full = args.pool * args.reps
    order = np.argsort(-np.array(sizes), kind="stable")
    rng = np.random.default_rng(3)
    cases = [("pool order", comps, sizes, None, 1),
             ("grouped, falling size", comps, sizes, order, args.reps),
             ("grouped, rising size", comps, sizes, order[::-1].copy(), args.reps),
             ("grouped, random order", comps, sizes, rng.permutation(args.pool), args.reps),
             ("pool order over a pool sorted by falling size", [comps[j] for j in order], [sizes[j] for j in order], None, 1)]
    # pool order regrouped into size classes (the stream's compressed + decompressed bytes, 8 classes per doubling),
    # pool order kept inside a class: what a counting sort by class gives; falling and rising classes
    seq = np.arange(full) % args.pool
    cost = np.array([len(c) for c in comps], dtype=np.float64) + np.array(sizes, dtype=np.float64)
    cls = np.floor(np.log2(cost[seq]) * 8).astype(np.int64)
    cases.append(("pool order, counting-sorted into size classes (falling)", comps, sizes, seq[np.argsort(-cls, kind="stable")], 1))
    cases.append(("pool order, counting-sorted into size classes (rising)", comps, sizes, seq[np.argsort(cls, kind="stable")], 1))
    cls1 = np.floor(np.log2(cost[seq])).astype(np.int64)
    cases.append(("pool order, counting-sorted into classes of one doubling (falling)", comps, sizes, seq[np.argsort(-cls1, kind="stable")], 1))
    for name, c, s, perm, reps in cases:
        b = bench.Batch(torch, dev, c, s, 0, full, perm, reps)
        _, ms = bench.time_steps(torch, bench.make_step(eng.lib, torch, b, 1, False), args.steps, 1, lambda: None, False)
        good = int(((b.d_status[: b.n] == 0) & (b.d_outlen[: b.n] == b.want_len)).sum())
        print("%-50s %9.3f ms  %7.1f GB/s decompressed  (%d of %d streams fine)" % (name, ms, b.out_bytes / ms / 1e6, good, b.n), flush=True)
        del b
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
