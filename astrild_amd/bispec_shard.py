"""Bispectrum on several GPUs of one node: the triangle bins are distributed.

Every rank holds the (replicated) grid - 0.5 GiB at 512^3 fp32, broadcast from the root if need be - and
evaluates a contiguous chunk of the sorted triangle list with the single-GPU estimator, transforming only
the shells its own triangles touch; one all-gather of (B, N_tri) per call puts the table together.  There is
no exchange of fields: sorted triangles share shells, so a rank with T/P triangles needs about
min(S, 3 T / P) of the S masked inverse FFTs (31 shells, 75 bins, P = 8: 6-9 transforms per rank instead of
31) and exactly T/P of the triple-product sums.  A slab-decomposed inverse FFT per shell (one all-to-all per
shell, 31 of them) is the alternative §8e names; it pays only when one shell field does not fit a GPU.

The collective logic talks to an ``ops`` object; ``HipBispecOps`` (the product) calls
``device.bispectrum``, the CPU tests inject a numpy double over gloo.
"""
import numpy as np
import torch
import torch.distributed as dist


class HipBispecOps:
    def __init__(self):
        from . import device as dev
        self.dev = dev
        self.device = dev.device()

    def bispectrum(self, field, boxsize, edges, triangles):
        return self.dev.bispectrum(field, boxsize, edges, triangles)

    def tensor(self, a, dtype):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device)


def split_triangles(triangles, world):
    """Contiguous chunks of the lexicographically sorted triangle list, sizes differing by at most one.
    Returns (order, bounds): rank r evaluates triangles[order[bounds[r]:bounds[r+1]]]."""
    tri = [tuple(int(v) for v in t) for t in triangles]
    order = sorted(range(len(tri)), key=lambda i: tri[i])
    base, extra = divmod(len(tri), world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + base + (1 if r < extra else 0))
    return order, bounds


def bispectrum_sharded(field, boxsize, edges, triangles, group=None, root_has_field=False, ops=None):
    """``device.bispectrum`` with the triangle bins spread over the ranks of ``group``.  ``field``: the grid on
    every rank (``root_has_field``: only rank 0's is valid, it is broadcast).  Every rank returns the full
    dict(B, ntri, k) in the order of ``triangles``."""
    ops = ops or HipBispecOps()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    from .slab import comm_ready
    comm_ready(group)
    if root_has_field and world > 1:
        dist.broadcast(field, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    tri = [tuple(int(v) for v in t) for t in triangles]
    order, bounds = split_triangles(tri, world)
    mine = [tri[i] for i in order[bounds[rank]:bounds[rank + 1]]]
    width = max(bounds[r + 1] - bounds[r] for r in range(world))
    b_loc = np.full(width, np.nan)
    n_loc = np.zeros(width, dtype=np.int64)
    if mine:
        res = ops.bispectrum(field, boxsize, edges, mine)
        b_loc[:len(mine)] = res["B"]
        n_loc[:len(mine)] = res["ntri"]
    b_all = [ops.tensor(np.zeros(width), torch.float64) for _ in range(world)]
    n_all = [ops.tensor(np.zeros(width, dtype=np.int64), torch.int64) for _ in range(world)]
    dist.all_gather(b_all, ops.tensor(b_loc, torch.float64), group=group)
    dist.all_gather(n_all, ops.tensor(n_loc, torch.int64), group=group)
    b = np.empty(len(tri))
    ntri = np.empty(len(tri), dtype=np.int64)
    for r in range(world):
        idx = order[bounds[r]:bounds[r + 1]]
        b[idx] = b_all[r].cpu().numpy()[:len(idx)]
        ntri[idx] = n_all[r].cpu().numpy()[:len(idx)]
    kf = 2.0 * np.pi / boxsize
    edges = [int(e) for e in edges]
    kmid = np.array([[kf * 0.5 * (edges[s] + edges[s + 1]) for s in t] for t in tri])
    return {"B": b, "ntri": ntri, "k": kmid}
