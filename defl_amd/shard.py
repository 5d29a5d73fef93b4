"""Multi-GPU sharding of a batch of independent streams (one process per GPU).

A DEFLATE stream depends on no other stream, so a batch is split by stream index and every
rank decodes its own contiguous range; there is no exchange step on the data path.  The only
collective is the sum of the per-rank result counters (RCCL through torch.distributed; `gloo`
on CPU tensors in the tests).  The reference has no notion of this (single-threaded library).
"""
import numpy as np


def partition(cost, world):
    """Contiguous ranges [lo, hi) per rank that balance sum(cost) (cost_i = compressed +
    decompressed bytes of stream i).  Returns a list of `world` (lo, hi) pairs covering 0..n."""
    cost = np.asarray(cost, dtype=np.float64)
    n = len(cost)
    if world <= 1 or n == 0:
        return [(0, n)] + [(n, n)] * (max(1, world) - 1)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        # pick the boundary whose prefix sum is closest to the target
        if k > 0 and abs(cum[k - 1] - target) <= abs(cum[min(k, n)] - target):
            k -= 1
        bounds.append(min(max(k, bounds[-1]), n))
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def summarize(status, outlen, dist=None, device=None):
    """(n_ok, n_err, bytes_out) summed over all ranks.  `status`/`outlen` are this rank's result
    arrays (numpy or torch).  With a process group the three counters are all-reduced; gloo
    gets CPU tensors, RCCL gets device tensors."""
    import torch
    st = torch.as_tensor(np.asarray(status) if not hasattr(status, "device") else status)
    ol = torch.as_tensor(np.asarray(outlen) if not hasattr(outlen, "device") else outlen)
    ok = (st == 0)
    t = torch.stack([ok.sum().to(torch.int64), (~ok).sum().to(torch.int64),
                     (ol.to(torch.int64) * ok.to(torch.int64)).sum()])
    if dist is not None and dist.is_initialized():
        if dist.get_backend() == "gloo":
            t = t.cpu()
        elif device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    n_ok, n_err, nbytes = (int(x) for x in t.cpu().tolist())
    return n_ok, n_err, nbytes
