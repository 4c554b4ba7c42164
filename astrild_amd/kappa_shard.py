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

A stream of maps (``MapStream``) rotates the root and pipelines the maps with a lag of one: map m + 1's partial sum and
all-to-all are enqueued before map m is finished, rank m mod P receives map m and runs its per-map stages (smoothing,
kappa -> alpha, PDF) on a second stream.

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


class ShardedStacker:
    """The buffers and the collective sequence of one sharded stack of ``n``-pixel maps, reusable map after map (no
    allocation, no size negotiation per map), in two halves so that a stream of maps can overlap them:

      ``begin``     the rank's weighted partial sum into a send buffer, then the all-to-all of map chunks - ASYNCHRONOUS:
                    RCCL moves the chunks on its own stream while the caller's stream goes on (to the next map's partial
                    sum, say);
      ``complete``  waits for those chunks, adds them in rank order, gathers the summed chunks on the root (asynchronous
                    too) and returns (result buffer or None, gather handle or None).

    Send / receive / chunk buffers and result buffers rotate (``depth`` of each): map m + 1 may begin while map m is on
    the links, and a consumer on another stream may read map m's result while map m + 1 is gathered."""

    def __init__(self, n, group=None, ops=None, depth=2):
        self.ops = ops or HipStackOps()
        self.group, self.n = group, int(n)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.chunk = (self.n + self.world - 1) // self.world
        self.depth = max(1, int(depth))
        o, multi = self.ops, self.world > 1
        # zero padding past n stays zero (only [:n] is ever written); a rank without planes sends zeros
        self.send = [o.zeros(self.world * self.chunk) for _ in range(self.depth if multi else 1)]
        self.recv = [o.empty(self.world * self.chunk) for _ in range(self.depth)] if multi else None
        self.mine = [o.empty(self.chunk) for _ in range(self.depth)] if multi else None
        self._mine_busy = [None] * self.depth          # the gather still reading mine[i]
        self.full = [o.empty(self.world * self.chunk) for _ in range(self.depth)]
        self.begun = 0                     # maps begun (send / recv / mine rotate per map)
        self.uses = 0                      # result buffers handed out by THIS rank (they rotate per use, not per map)

    def begin(self, planes, wnum=None, wden=None):
        """First half for the next map; returns a ticket for ``complete``."""
        from .slab import comm_ready
        o, world, n, group = self.ops, self.world, self.n, self.group
        planes = [o.to_device(p).reshape(-1) for p in planes]
        if planes and planes[0].numel() != n:
            raise ValueError(f"rank {self.rank}: planes of {planes[0].numel()} pixels, the stacker was made for {n}")
        slot = self.begun % self.depth
        self.begun += 1
        if world == 1:
            if not planes:
                raise ValueError("no rank holds a plane")
            return {"slot": slot, "local": (planes, wnum, wden), "work": None}
        send = self.send[slot]
        if planes:
            o.stack(planes, wnum, wden, out=send[:n])
        else:
            send[:n].zero_()
        comm_ready(group)
        work = dist.all_to_all_single(self.recv[slot], send, group=group, async_op=True)       # chunk j of every rank -> rank j
        return {"slot": slot, "work": work}

    def complete(self, ticket, root=0, all_ranks=False):
        """Second half: (summed map - flat, n pixels, a view of a rotating buffer - on ``root`` / on every rank with
        ``all_ranks``, else None;  handle of the gather that fills it, or None).  The CURRENT stream is made to wait for the
        chunks; the gather runs asynchronously - wait for its handle on the stream that reads the result."""
        from .slab import comm_ready
        o, world, rank, chunk, n, group = self.ops, self.world, self.rank, self.chunk, self.n, self.group
        full = None
        if world == 1 or all_ranks or rank == root:
            full = self.full[self.uses % self.depth]
            self.uses += 1
        if world == 1:
            planes, wnum, wden = ticket["local"]
            o.stack(planes, wnum, wden, out=full[:n])
            return full[:n], None
        slot = ticket["slot"]
        ticket["work"].wait()                              # (the current stream waits; the host does not, with RCCL)
        if self._mine_busy[slot] is not None:              # the gather of `depth` maps ago still owns this chunk buffer
            self._mine_busy[slot].wait()
        mine, recv = self.mine[slot], self.recv[slot]
        o.stack([recv[s * chunk:(s + 1) * chunk] for s in range(world)], out=mine)      # rank order: fixed
        comm_ready(group)
        if all_ranks:
            work = dist.all_gather_into_tensor(full, mine, group=group, async_op=True)
        else:
            # `root` is a rank of `group`; dist.gather's dst is a GLOBAL rank
            parts = list(full.split(chunk)) if rank == root else None
            work = dist.gather(mine, parts, dst=dist.get_global_rank(group, root) if group is not None else root, group=group,
                               async_op=True)
        self._mine_busy[slot] = work
        return (full[:n] if full is not None else None), work

    def stack(self, planes, wnum=None, wden=None, root=0, all_ranks=False):
        """Both halves, one map: the summed map on ``root`` (on every rank with ``all_ranks``), None elsewhere; the
        current stream has been made to wait for it."""
        res, work = self.complete(self.begin(planes, wnum, wden), root=root, all_ranks=all_ranks)
        if work is not None:
            work.wait()
        return res


class MapStream:
    """A stream of output maps over the P ranks of ``group`` (the loops of simcoll.py:267-336 and rayramses.py:186-232,
    one stacked map per iteration): every rank stacks its planes of map m, the partial maps are reduced onto rank
    m mod P - the root ROTATES - and that rank runs ``tail(m, map)`` (smoothing, kappa -> alpha, PDF ...) on a second
    stream.  The stream is a software pipeline with a lag of one map: ``push(m + 1)`` enqueues map m + 1's partial sum and
    all-to-all FIRST and only then finishes map m (chunk sums, gather, tail) - so map m's chunks travel while map m + 1's
    planes are being added, and the per-map stages run on one GPU while all GPUs are already stacking the next maps.
    With one GPU per rank each rank runs a tail every P-th map instead of P - 1 of them idling behind rank 0.

    ``tail`` must not synchronise the host (queue results, e.g. lensing.PendingHistogram).  What the tails returned is in
    ``results`` (map index -> value, on the map's root) once ``finish()`` has been called."""

    def __init__(self, npix2, group=None, ops=None):
        self.stacker = ShardedStacker(npix2, group, ops, depth=2)
        self.group = group
        self.world, self.rank = self.stacker.world, self.stacker.rank
        self.m = 0
        self.on_gpu = torch.cuda.is_available() and self.stacker.full[0].is_cuda
        self.tail_stream = torch.cuda.Stream() if self.on_gpu else None
        self._tail_done = []               # events: the tails that read the rotating result buffers
        self._pending = None               # the map whose first half is in flight
        self.results = {}

    def root_of(self, m):
        return m % self.world

    def push(self, planes, wnum=None, wden=None, tail=None):
        ticket = self.stacker.begin(planes, wnum, wden)
        previous, self._pending = self._pending, (self.m, ticket, tail)
        self.m += 1
        if previous is not None:
            self._complete(*previous)

    def _complete(self, m, ticket, tail):
        root = self.root_of(m)
        mine = self.rank == root
        if self.on_gpu and mine and len(self._tail_done) >= self.stacker.depth:
            # the buffer this map is gathered into was read by the tail of my map before last
            torch.cuda.current_stream().wait_event(self._tail_done.pop(0))
        res, work = self.stacker.complete(ticket, root=root)
        if not mine or tail is None:
            return
        if not self.on_gpu:
            if work is not None:
                work.wait()
            self.results[m] = tail(m, res)
            return
        self.tail_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.tail_stream):
            if work is not None:
                work.wait()                               # the TAIL stream waits for the gather; the stacking stream goes on
            self.results[m] = tail(m, res)
            ev = torch.cuda.Event()
            ev.record(self.tail_stream)
        self._tail_done.append(ev)

    def finish(self):
        """Finishes the map still in flight and joins the tail stream; returns ``results``."""
        if self._pending is not None:
            previous, self._pending = self._pending, None
            self._complete(*previous)
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.tail_stream)
        self._tail_done = []
        return self.results


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
    return ShardedStacker(n, group, ops, depth=1).stack(planes, wnum, wden, root=root, all_ranks=all_ranks)
