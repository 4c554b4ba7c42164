"""kappa-plane stack sharded over the GPUs of one node (BASELINE config D).

The lens planes of ``RayRamses.sum_snapshots`` / ``SimulationCollection.sum_raytracing_snapshots``
(rayramses.py:186-232, simcoll.py:267-336) are independent until the final sum, so plane ``p``
lives on rank ``p mod P`` (each rank also only LOADS its own files).  Per output map:

  1. every rank forms the weighted partial sum of its planes (``ast_kappa_stack``: one pass over
     its planes, plane order increasing);
  2. the partial maps are cut into P chunks and exchanged with ONE all-to-all (chunk j of every
     rank goes to rank j): xGMI is point to point, so all 7 links of a GPU carry 1/P of a map at
     once (16.8 MB per link for a 4096^2 fp64 map at P = 8) - a ring all-reduce would push
     2 (P-1)/P of the map through one link per GPU;
  3. rank j adds the P chunks it received IN RANK ORDER (``ast_kappa_stack`` again);
  4. the summed chunks are gathered on the root (or on every rank, ``all_ranks=True``).

Summation order: sum over ranks r = 0..P-1 of (sum over the planes p = r, r+P, ... of rank r) - fixed,
so the result is bit-reproducible for a given P; it differs from the single-GPU running sum
(planes 0, 1, 2, ... in order) by re-association only (<= P * 2^-53 relative; the tests bound it).

The collective logic talks to an ``ops`` object; ``HipStackOps`` (the product) calls the C-ABI, the
CPU tests inject a numpy double and run over gloo.
"""
import numpy as np
import torch
import torch.distributed as dist


class HipStackOps:
    """Local arithmetic on the GPU through the C-ABI."""

    def __init__(self):
        from . import device as dev, lensing
        self.dev, self.lensing = dev, lensing
        self.device = dev.device()

    def to_device(self, a):
        return self.dev.as_device(np.ascontiguousarray(a, dtype=np.float64)) if not isinstance(a, torch.Tensor) else a

    def zeros(self, n):
        return torch.zeros(n, dtype=torch.float64, device=self.device)

    def empty(self, n):
        return torch.empty(n, dtype=torch.float64, device=self.device)

    def stack(self, planes, wnum=None, wden=None, out=None):
        return self.lensing.kappa_stack(planes, wnum, wden, out=out)


def my_plane_ids(nplanes, group=None):
    """Global indices of the planes this rank owns: p = rank, rank + P, ..."""
    return list(range(dist.get_rank(group), int(nplanes), dist.get_world_size(group)))


def kappa_stack_sharded(planes, wnum=None, wden=None, group=None, root=0, all_ranks=False, ops=None):
    """Weighted sum of ALL ranks' planes.  ``planes``: this rank's planes (flat or 2-D, equal shapes; may be
    empty), ``wnum`` / ``wden``: their weights (or None).  Returns the summed map (flat, length of one
    plane) on ``root`` - on every rank with ``all_ranks`` - and None elsewhere."""
    ops = ops or HipStackOps()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    planes = [ops.to_device(p).reshape(-1) for p in planes]
    n_local = planes[0].numel() if planes else 0
    nmax = torch.tensor([n_local], dtype=torch.int64)
    if dist.get_backend(group) == "nccl":
        nmax = nmax.to(ops.device)
    dist.all_reduce(nmax, op=dist.ReduceOp.MAX, group=group)
    n = int(nmax.item())
    if n == 0:
        raise ValueError("no rank holds a plane")
    if planes and n_local != n:
        raise ValueError(f"rank {rank}: planes of {n_local} pixels, other ranks have {n}")
    chunk = (n + world - 1) // world
    send = ops.zeros(world * chunk)                       # zero padding past n; ranks without planes send zeros
    if planes:
        ops.stack(planes, wnum, wden, out=send[:n])
    if world == 1:
        return send[:n]
    recv = ops.empty(world * chunk)
    from .slab import comm_ready
    comm_ready(group)
    dist.all_to_all_single(recv, send, group=group)       # chunk j of every rank -> rank j
    mine = ops.stack([recv[s * chunk:(s + 1) * chunk] for s in range(world)])      # rank order: fixed
    comm_ready(group)
    if all_ranks:
        full = ops.empty(world * chunk)
        dist.all_gather_into_tensor(full, mine, group=group)
        return full[:n]
    # `root` is a rank of `group`; dist.gather's dst is a GLOBAL rank
    parts = [ops.empty(chunk) for _ in range(world)] if rank == root else None
    dist.gather(mine, parts, dst=dist.get_global_rank(group, root) if group is not None else root, group=group)
    if rank != root:
        return None
    return torch.cat(parts)[:n]
