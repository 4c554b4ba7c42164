"""Slab-decomposed 3D pipeline: paint -> FFT -> P(k) over P GPUs of one node.

The grid is split along axis 0 (the slowest axis of ``value_map[x, y, z]``; what
BASELINE.json calls z-slabs): rank r owns planes [r*N/P, (r+1)*N/P).  One process
per GPU, ``torch.distributed`` over RCCL/xGMI (backend "nccl").  Per step:

  1. paint the rank's particles into its slab buffer (owned planes + ghost planes);
  2. ghost fold: ghost planes go to the two ring neighbours and are added into
     their owned planes (grouped send/recv, N^2 elements per plane);
  3. batched 2D R2C over (y, z) of the owned planes, in `chunks` groups of planes;
  4. pack + all-to-all per chunk (grouped send/recv to every peer, issued asynchronously
     so chunk c travels while chunk c+1 is transformed): rank r keeps ky in
     [r*N/P, (r+1)*N/P) for ALL x.  With point-to-point xGMI every GPU talks to its 7
     peers at once, so all 7 links carry 1/P^2 of the spectrum each (67 MB per pair at
     1024^3 fp32, P = 8);
  5. strided 1D C2C along x (1/Ng folded into this pass);
  6. shell binning of the local (N, N/P, N/2+1) block, then one all-reduce of the
     N/2-1 shell sums (4 KB).

The collective logic is independent of where the local arithmetic runs: it talks
to an ``ops`` object.  ``HipSlabOps`` (the product) calls libastrild_hip.so; the
CPU tests inject a numpy double to exercise steps 2, 4 and 6 over gloo.
"""
import numpy as np
import torch
import torch.distributed as dist


class HipSlabOps:
    """Local arithmetic on the GPU through the C-ABI."""

    paint_hints = True          # paint() takes hint= (the numpy double of the CPU tests does not)

    def __init__(self, dtype=torch.float32):
        from . import device as dev
        self.dev = dev
        self.dtype = dtype
        self.cdtype = torch.complex64 if dtype == torch.float32 else torch.complex128
        self.device = dev.device()

    def zeros(self, shape, dtype=None):
        return torch.zeros(shape, dtype=dtype or self.dtype, device=self.device)

    def empty(self, shape, dtype=None):
        return torch.empty(shape, dtype=dtype or self.dtype, device=self.device)

    def paint(self, pos, mass, n, boxsize, window, out, x_start, nx_alloc, check=False, offset=0.0, owned=None, hint=None):
        return self.dev.paint(pos, mass, n, boxsize, window, out=out, x_start=x_start, nx_alloc=nx_alloc,
                              check_dropped=check, accumulate=False, offset=offset, offset_planes=owned, hint=hint)

    def lowk_supported(self, n):
        return self.dtype == torch.float32 and self._tile_ok(0, n)

    def lowk_modes(self, planes, n, x0):
        return self.dev.lowk_modes(planes.contiguous(), n, x0)

    def lowk_patch(self, modes, n, boxsize, psum):
        sums = self.dev.lowk_shell_sums(modes, n, boxsize)
        psum[:sums.numel()] = sums
        return psum

    def add_into(self, dst, src):
        from ._lib import check, lib
        check(lib().ast_accumulate(self.dev.ptr(dst), self.dev.ptr(src), self.dev.real_code(dst), dst.numel(),
                                   self.dev.stream()), "ast_accumulate")

    def _tile_ok(self, code, n):
        from ._lib import lib
        return bool(lib().ast_fft_tile_supported(code, n))

    def fft2d_planes(self, planes, out):
        from ._lib import check, lib
        nloc, n1, n2 = planes.shape
        code = self.dev.real_code(planes)
        if n1 == n2 and self._tile_ok(code, n1):
            # hand-written passes: z rows (R2C) then y columns of every local plane
            nz = n2 // 2 + 1
            check(lib().ast_fft_tile_rows_r2c(self.dev.ptr(planes), self.dev.ptr(out), code, n2, nloc * n1, n2, nz,
                                              1.0, self.dev.stream()), "ast_fft_tile_rows_r2c")
            check(lib().ast_fft_tile_c2c(self.dev.ptr(out), code, n1, nz, nz, nloc, n1 * nz, 1.0, self.dev.stream()),
                  "ast_fft_tile_c2c")
            return out
        self.dev.fft_plan(0, code, (n1, n2), nloc, 1.0, False).execute(planes, out)      # AST_FFT_R2C
        return out

    def packed_supported(self, planes, parts):
        nloc, n1, n2 = planes.shape
        return n1 == n2 and self._tile_ok(self.dev.real_code(planes), n1) and parts & (parts - 1) == 0

    def fft2d_planes_packed(self, planes, spec, packed, parts, self_part, self_dst):
        """fft2d_planes + pack in two kernels instead of three: the y pass stores straight into the send buffer
        (piece s of `packed`) and the rank's own piece into ``self_dst`` (its place in the receive block)."""
        from ._lib import check, lib
        nloc, n1, n2 = planes.shape
        code = self.dev.real_code(planes)
        nz = n2 // 2 + 1
        assert self_dst.is_contiguous() and packed.is_contiguous() and spec.is_contiguous()
        check(lib().ast_fft_tile_rows_r2c(self.dev.ptr(planes), self.dev.ptr(spec), code, n2, nloc * n1, n2, nz,
                                          1.0, self.dev.stream()), "ast_fft_tile_rows_r2c")
        check(lib().ast_fft_tile_c2c_packed(self.dev.ptr(spec), self.dev.ptr(packed), code, n1, nz, nloc, parts,
                                            self_part, self.dev.ptr(self_dst), 1.0, self.dev.stream()),
              "ast_fft_tile_c2c_packed")

    def pack(self, spec, out, parts):
        from ._lib import check, lib
        n0, n1, n2 = spec.shape
        check(lib().ast_slab_pack(self.dev.ptr(spec), self.dev.ptr(out), 0 if spec.dtype == torch.complex64 else 1,
                                  n0, n1, n2, parts, self.dev.stream()), "ast_slab_pack")
        return out

    def fft1d_axis0(self, block, scale):
        from ._lib import check, lib
        n0, n1, n2 = block.shape
        code = 0 if block.dtype == torch.complex64 else 1
        if self._tile_ok(code, n0):
            check(lib().ast_fft_tile_c2c(self.dev.ptr(block), code, n0, n1 * n2, n1 * n2, 1, 0, scale,
                                         self.dev.stream()), "ast_fft_tile_c2c")
            return block
        plan = self.dev.fft_plan(2, code, (n0,), n1 * n2, scale, True, strided=(n0, n1 * n2, 1))   # AST_FFT_C2C_FWD
        plan.execute(block, None)
        return block

    def axis0_power_supported(self, n):
        return self.dtype == torch.float32 and self._tile_ok(0, n)

    def fft1d_axis0_power(self, block, scale, n, boxsize, ky0, psum, first_bin):
        """The axis-0 pass fused with the shell binning of the block (ast_fft_tile_block_power): delta_k never goes
        back to HBM.  psum is overwritten with the block's shell sums (shells below first_bin left at zero)."""
        from ._lib import check, lib
        n0, nloc, pitch = block.shape
        key = (n, nloc)
        if getattr(self, "_bp_key", None) != key:
            self._bp_key = key
            self._bp_scratch = torch.empty(int(lib().ast_fft_tile_block_power_scratch_bytes(n, nloc)), dtype=torch.uint8,
                                           device=self.device)
        psum.zero_()
        check(lib().ast_fft_tile_block_power(self.dev.ptr(block), self.dev.ptr(self._bp_scratch), self._bp_scratch.numel(), 0, n,
                                             nloc, int(ky0), pitch, float(scale), float(boxsize), int(first_bin),
                                             self.dev._bin_code(None), self.dev.ptr(psum), self.dev.stream()),
              "ast_fft_tile_block_power")
        return psum

    def shell_geometry(self, n, boxsize, i0, i1):
        return self.dev.shell_geometry(n, boxsize, i0, i1)

    def power_bin(self, block, n, boxsize, i0, i1, psum):
        psum.zero_()
        self.dev.power_bin_1d(block, None, n, boxsize, i0, i1, psum=psum)
        return psum

    def route_count(self, pos, n, boxsize, window, parts):
        from ._lib import WIN, check, lib
        counts = torch.zeros(parts, dtype=torch.int64, device=self.device)
        check(lib().ast_route_count(self.dev.ptr(pos), self.dev.real_code(pos), pos.shape[0], n, float(boxsize),
                                    WIN[window.lower()], parts, self.dev.ptr(counts), self.dev.stream()), "ast_route_count")
        return counts

    def route_scatter(self, pos, mass, n, boxsize, window, parts, counts):
        from ._lib import WIN, check, lib
        cursor = torch.cumsum(counts, 0) - counts                  # exclusive prefix sums (P values)
        out_pos = torch.empty_like(pos)
        out_mass = None if mass is None else torch.empty_like(mass)
        check(lib().ast_route_scatter(self.dev.ptr(pos), self.dev.ptr(mass), self.dev.real_code(pos), pos.shape[0], n,
                                      float(boxsize), WIN[window.lower()], parts, self.dev.ptr(cursor), self.dev.ptr(out_pos),
                                      self.dev.ptr(out_mass), self.dev.stream()), "ast_route_scatter")
        return out_pos, out_mass

    def synth(self, npside, n, boxsize, seed, shuffle, first, count):
        return self.dev.synth_lattice_particles(npside, n, boxsize, seed=seed, shuffle=shuffle, dtype=self.dtype,
                                                first=first, count=count)


def comm_ready(group=None):
    """RCCL orders a collective after the kernels already queued on the current stream.  The gloo backend (CPU tests,
    and the one-GPU rehearsals that put several ranks on one card) reads device buffers from its own threads: there
    the producers have to be finished first."""
    if torch.cuda.is_available() and torch.cuda.is_initialized() and dist.get_backend(group) != "nccl":
        torch.cuda.synchronize()


class GhostExchange:
    """Step 2 in two halves.  buf: (gl + nloc + gh, N, N) with the owned planes in the middle.  Lower ghosts belong
    to rank r-1 (its top gl planes), upper ghosts to rank r+1 (its bottom gh planes).  ``start`` posts the sends and
    receives (both ghost blocks are contiguous plane ranges of buf: nothing is copied), ``finish`` waits for them
    and adds the incoming planes - only the first gh and the last gl owned planes change, so everything that reads
    the planes in between may run while the ghosts travel."""

    def __init__(self, buf, nloc, gl, gh, ops, group=None):
        self.buf, self.nloc, self.gl, self.gh, self.ops, self.group = buf, nloc, gl, gh, ops, group
        self.from_right = torch.empty_like(buf[:gl])       # right neighbour's lower ghosts -> my top planes
        self.from_left = torch.empty_like(buf[gl + nloc:])  # left neighbour's upper ghosts -> my bottom planes
        self.reqs = None

    def start(self):
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        left, right = (rank - 1) % world, (rank + 1) % world
        comm_ready(self.group)
        self.reqs = dist.batch_isend_irecv([
            dist.P2POp(dist.isend, self.buf[:self.gl], left, self.group),
            dist.P2POp(dist.isend, self.buf[self.gl + self.nloc:], right, self.group),
            dist.P2POp(dist.irecv, self.from_right, right, self.group),
            dist.P2POp(dist.irecv, self.from_left, left, self.group),
        ])

    def finish(self):
        if self.reqs is None:
            return
        for q in self.reqs:
            q.wait()
        self.reqs = None
        owned = self.buf[self.gl: self.gl + self.nloc]
        self.ops.add_into(owned[self.nloc - self.gl:], self.from_right)
        self.ops.add_into(owned[:self.gh], self.from_left)


def ghost_fold(buf, nloc, gl, gh, ops, group=None):
    """Step 2 as one blocking call (see GhostExchange)."""
    ex = GhostExchange(buf, nloc, gl, gh, ops, group)
    ex.start()
    ex.finish()
    return buf[gl: gl + nloc]


def exchange_chunk(packed_c, block, chunk, pc, nloc, group=None, self_done=False):
    """Step 4 for one chunk of `pc` local planes.  packed_c: (P, pc, nly, nz) — piece s goes
    to rank s; it lands in block[s_src*nloc + chunk*pc : ... + pc] of the receiver.  Returns
    the outstanding work handles (the local piece is copied right away, unless the producer has
    already written it into the block: ``self_done``)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ops_list = []
    if world > 1:
        comm_ready(group)
    for s in range(world):
        dst = block[s * nloc + chunk * pc: s * nloc + (chunk + 1) * pc]
        if s == rank:
            if not self_done:
                dst.copy_(packed_c[s])
            continue
        # complex payload moved as (re, im) pairs of the real dtype: every c10d backend takes that
        ops_list.append(dist.P2POp(dist.isend, torch.view_as_real(packed_c[s]), s, group))
        ops_list.append(dist.P2POp(dist.irecv, torch.view_as_real(dst), s, group))
    return dist.batch_isend_irecv(ops_list) if ops_list else []


def route_particles(pos, mass, n, boxsize, window, ops, group=None):
    """particles that are not partitioned by slab go to the rank that owns their base plane -
    count per destination, group by destination on the device, one all-to-all of the counts and one all-to-all-v of
    the 12-16 B per particle (positions, then masses).  Returns this rank's (pos, mass); a deposit then reaches at
    most the window's own width beyond the slab (ghost = 0)."""
    world = dist.get_world_size(group)
    if world == 1:
        return pos, mass
    counts = ops.route_count(pos, n, boxsize, window, world)
    spos, smass = ops.route_scatter(pos, mass, n, boxsize, window, world, counts)
    incoming = torch.empty_like(counts)
    comm_ready(group)
    dist.all_to_all_single(incoming, counts, group=group)
    send, recv = [int(v) for v in counts.tolist()], [int(v) for v in incoming.tolist()]
    out_pos = spos.new_empty((sum(recv), 3))
    dist.all_to_all_single(out_pos, spos, output_split_sizes=recv, input_split_sizes=send, group=group)
    out_mass = None
    if mass is not None:
        out_mass = smass.new_empty((sum(recv),))
        dist.all_to_all_single(out_mass, smass, output_split_sizes=recv, input_split_sizes=send, group=group)
    return out_pos, out_mass


class SlabPowerPipeline:
    """CIC/TSC + slab FFT + P(k) for the synthetic lattice workload of bench.py."""

    def __init__(self, n, boxsize, npside, window="cic", dtype=torch.float32, seed=20240601, shuffle=False,
                 ghost=4, ops=None, group=None, pos=None, chunks=None, route=False):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        P = self.world
        if n % P or npside % P:
            raise ValueError(f"grid {n} and particle lattice {npside} must be divisible by the number of ranks {P}")
        self.n, self.L, self.window = n, float(boxsize), window
        self.ops = ops or HipSlabOps(dtype)
        if route:
            # arbitrary input (`pos`, or the synthetic set in shuffled order): every particle goes to the rank that
            # owns its base plane first, so the deposits reach no further than the window itself
            ghost = 0
        self.nloc = n // P
        self.nz = n // 2 + 1
        # particles jitter across slab boundaries: ghost planes take base cells up to
        # `ghost` planes outside, plus the window's own reach of one plane
        self.gl = self.gh = ghost + 1
        if P == 1:
            self.gl = self.gh = 0          # one rank owns the whole periodic grid: no ghosts, no fold
        elif self.nloc + self.gl + self.gh > n:
            raise ValueError("slab too thin for the ghost zone")
        self.x_start = (self.rank * self.nloc - self.gl) % n
        self.nx_alloc = self.nloc + self.gl + self.gh
        ppr = npside ** 3 // P
        self.pos = pos if pos is not None else self.ops.synth(npside, n, self.L, seed, shuffle, self.rank * ppr, ppr)
        self.npart_total = float(npside) ** 3
        if route:
            self.pos, _ = route_particles(self.pos, None, n, self.L, window, self.ops, group)
        o = self.ops
        self.buf = o.empty((self.nx_alloc, n, n))
        self.spec2d = o.empty((self.nloc, n, self.nz), o.cdtype)
        if chunks is None:
            chunks = 4 if self.nloc % 4 == 0 and self.nloc >= 16 else 1
        if self.nloc % chunks:
            raise ValueError(f"{self.nloc} local planes do not split into {chunks} chunks")
        self.chunks = chunks
        self.pc = self.nloc // chunks
        self.packed = o.empty((chunks, P, self.pc, self.nloc, self.nz), o.cdtype)
        self.block = o.empty((n, self.nloc, self.nz), o.cdtype)
        self.psum = o.zeros((n // 2 - 1,), torch.float64)
        self._side = None
        self._stage_s = {}
        # (hint="xsorted" - grouping and walks overlapped chunk by chunk - is available but buys nothing measurable:
        # both paint kernels are bound by their index arithmetic)
        self.hint = None
        self.ghosts = GhostExchange(self.buf, self.nloc, self.gl, self.gh, o, group) if P > 1 else None
        # chunks whose planes the incoming ghosts do not touch are transformed (and sent) first, while the ghosts travel
        edge = {c for c in range(chunks) if c * self.pc < self.gh or (c + 1) * self.pc > self.nloc - self.gl}
        self.chunk_order = [c for c in range(chunks) if c not in edge] + sorted(edge)
        self.first_edge = len(self.chunk_order) - len(edge)
        # the rank's OWN planes hold rho - mean (subtracted before the fp32 rounding); its ghost planes, which are
        # added onto the neighbours' cells, stay plain sums
        lowk_fn = getattr(self.ops, "lowk_supported", None)
        self.lowk = bool(lowk_fn and lowk_fn(n))
        total = torch.tensor([float(self.pos.shape[0])], dtype=torch.float64, device=self.pos.device)
        comm_ready(group)
        dist.all_reduce(total, group=group)
        self.mean_offset = float(total.item()) / float(n) ** 3 if self.lowk else 0.0
        self.i0 = (0, n)
        self.i1 = (self.rank * self.nloc, self.nloc)
        ksum, nmodes = o.shell_geometry(n, self.L, self.i0, self.i1)
        self.ksum, self.nmodes = ksum.clone(), nmodes.clone()
        dist.all_reduce(self.ksum, group=group)
        dist.all_reduce(self.nmodes, group=group)

    def wire_bytes(self):
        """What one step of this rank puts on the links: ghost planes to the two ring neighbours, its pieces of the
        half spectrum to the P - 1 peers, and the all-reduced sums."""
        if self.world == 1:
            return {"ghost": 0, "transpose": 0, "allreduce": 0}
        esz = torch.empty((), dtype=self.ops.dtype).element_size() if hasattr(self.ops, "dtype") else 4
        return {"ghost": (self.gl + self.gh) * self.n * self.n * esz,
                "transpose": (self.world - 1) * self.nloc * self.nloc * self.nz * 2 * esz,
                "allreduce": (self.n // 2 - 1) * 8 + (1183 * 16 if self.lowk else 0)}

    def stage_ms(self, steps):
        """Host-side wall time per step spent in each stage's calls since the last reset (enqueue time with RCCL,
        real time with the synchronising gloo rehearsal): where a step's host thread goes."""
        out = {k: v / max(1, steps) for k, v in self._stage_s.items()}
        self._stage_s = {}
        return {k: v * 1e3 for k, v in out.items()}

    def _tick(self, name, t0):
        import time
        self._stage_s[name] = self._stage_s.get(name, 0.0) + (time.perf_counter() - t0)

    def paint(self, check=False, fold=True):
        """check=True synchronises and raises if a deposit fell outside the ghost zone.  The owned cells hold
        rho - mean (the mean is subtracted in double before the rounding to the grid dtype; ghost planes and halo
        records stay additive, so the fold still adds up): only the DC mode differs.  fold=False: the ghost exchange
        is only STARTED (self.ghosts.finish() completes the first and last owned planes)."""
        kw = {"hint": self.hint} if self.hint and getattr(self.ops, "paint_hints", False) else {}
        if self.mean_offset:
            self.ops.paint(self.pos, None, self.n, self.L, self.window, self.buf, self.x_start, self.nx_alloc, check,
                           offset=self.mean_offset, owned=(self.gl, self.nloc), **kw)
        else:
            self.ops.paint(self.pos, None, self.n, self.L, self.window, self.buf, self.x_start, self.nx_alloc, check, **kw)
        if self.world == 1:
            return self.buf
        self.ghosts.start()
        if fold:
            self.ghosts.finish()
        return self.buf[self.gl: self.gl + self.nloc]

    def forward_fft(self, owned, last_pass=True, before_edge=None):
        """Steps 3-5.  before_edge(): called once, before the first chunk that holds planes the ghost exchange
        changes (interior chunks are transformed and sent while the ghosts are still travelling)."""
        import time
        o = self.ops
        pending = []
        t0 = time.perf_counter()
        for i, c in enumerate(self.chunk_order):
            if i == self.first_edge and before_edge is not None:
                before_edge()
            planes = owned[c * self.pc:(c + 1) * self.pc]
            spec = self.spec2d[c * self.pc:(c + 1) * self.pc]
            packed_fn = getattr(o, "packed_supported", None)
            if packed_fn and packed_fn(planes, self.world):
                # y pass stores in send order; the rank's own piece goes straight into the receive block
                mine = self.block[self.rank * self.nloc + c * self.pc: self.rank * self.nloc + (c + 1) * self.pc]
                o.fft2d_planes_packed(planes, spec, self.packed[c], self.world, self.rank, mine)
                pending += exchange_chunk(self.packed[c], self.block, c, self.pc, self.nloc, self.group, self_done=True)
                continue
            o.fft2d_planes(planes, spec)
            o.pack(spec, self.packed[c], self.world)
            pending += exchange_chunk(self.packed[c], self.block, c, self.pc, self.nloc, self.group)
        if self.first_edge >= len(self.chunk_order) and before_edge is not None:
            before_edge()                     # (one rank: no chunk waits for ghosts)
        self._tick("fft2d+exchange.enqueue", t0)
        t0 = time.perf_counter()
        for work in pending:
            work.wait()
        self._tick("exchange.wait", t0)
        if not last_pass:
            return self.block
        return o.fft1d_axis0(self.block, 1.0 / float(self.n) ** 3)

    def step(self, check=False):
        import time
        t0 = time.perf_counter()
        owned = self.paint(check, fold=False)
        self._tick("paint.enqueue", t0)
        state = {"modes": None, "side": None}

        def finish_ghosts():
            t1 = time.perf_counter()
            if self.ghosts is not None:
                self.ghosts.finish()
            self._tick("ghost.wait+add", t1)
            if self.lowk:
                # the lowest shells from double-precision DFT sums of the rank's own (complete) planes
                # (device.power_sums_fused's low-k channel, split over the slabs): one more all-reduce, of 1183
                # complex numbers.  It only reads the planes: on the GPU it runs on a side stream beside the FFT
                # chunks and their exchange.
                if owned.is_cuda:
                    if self._side is None:
                        self._side = torch.cuda.Stream()
                    state["side"] = self._side
                    self._side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(self._side):
                        state["modes"] = self.ops.lowk_modes(owned, self.n, self.rank * self.nloc)
                else:
                    state["modes"] = self.ops.lowk_modes(owned, self.n, self.rank * self.nloc)

        fused_fn = getattr(self.ops, "axis0_power_supported", None)
        if fused_fn and fused_fn(self.n):
            # axis-0 pass and shell binning in one kernel (the spectrum block is not written back)
            block = self.forward_fft(owned, last_pass=False, before_edge=finish_ghosts)
            self.ops.fft1d_axis0_power(block, 1.0 / float(self.n) ** 3, self.n, self.L, self.rank * self.nloc, self.psum,
                                       5 if self.lowk else 0)
        else:
            block = self.forward_fft(owned, before_edge=finish_ghosts)
            self.ops.power_bin(block, self.n, self.L, self.i0, self.i1, self.psum)
        modes, side = state["modes"], state["side"]
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            modes.record_stream(torch.cuda.current_stream())      # allocated under the side stream, used from here on
        t0 = time.perf_counter()
        comm_ready(self.group)
        dist.all_reduce(self.psum, group=self.group)
        if modes is not None:
            modes_r = torch.view_as_real(modes)
            dist.all_reduce(modes_r, group=self.group)
            self.ops.lowk_patch(modes, self.n, self.L, self.psum)
        self._tick("allreduce.enqueue", t0)
        return self.ksum, self.psum, self.nmodes
