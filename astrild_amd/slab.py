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


def disc_layout(n, parts, r1, tile=16):
    """The DISC layout of the half spectrum between the k_y pass and the last pass (include/astrild_hip.h,
    ast_fft_tile_disc_layout; csrc/fft_tile.hip builds the same table - tests compare them): FFTPower(mode="1d", kmin=k_F)
    drops |m| >= n/2 (power_spectrum_3d.py:189-195), so a row (k_y, k_z tile from k_z0) with k_y^2 + k_z0^2 > (n/2)^2 is neither
    stored nor sent.  The k_y rows go to `parts` owners in blocks of `r1` rows, balanced by disc area; a part's plane is
    tile-major: for every tile of `tile` columns the part's rows inside the disc, in row order.

    Returns a dict: r1, tile, nblk, tiles, part_of[row block], gk[part] (its row blocks, ascending), S[part] (complex elements
    per plane), cumS[part], total, lo / hi / offb [tile][row block] (sub-rows lo <= sub < hi of the block exist; sub-row 0
    would land at element offb of the part's plane) and rows[part] / cols[part]: for every element of a part's plane the
    (k_y row, k_z column) it holds (columns past n/2 are padding)."""
    n, parts, r1, tile = int(n), int(parts), int(r1), int(tile)
    nblk, tiles, half = n // r1, (n // 2 + 1 + tile - 1) // tile, n // 2
    if nblk * r1 != n or nblk % parts or half % r1:
        raise ValueError(f"{n} rows in blocks of {r1} do not deal out to {parts} parts")
    nb = nblk // parts

    def inside(row, t):
        ky = row - n if row > half else row
        return ky * ky + (tile * t) ** 2 <= half * half

    area = [sum(inside(k * r1 + sub, t) for sub in range(r1) for t in range(tiles)) for k in range(nblk)]
    order = sorted(range(nblk), key=lambda k: (-area[k], k))
    sums, cnt, part_of = [0] * parts, [0] * parts, [0] * nblk
    for k in order:
        best = min((r for r in range(parts) if cnt[r] < nb), key=lambda r: (sums[r], r))
        part_of[k] = best
        sums[best] += area[k]
        cnt[best] += 1
    gk = [[k for k in range(nblk) if part_of[k] == r] for r in range(parts)]
    lo = np.zeros((tiles, nblk), dtype=np.int64)
    hi = np.zeros((tiles, nblk), dtype=np.int64)
    offb = np.zeros((tiles, nblk), dtype=np.int64)
    S, rows, cols = [], [], []
    for r in range(parts):
        count, prow, pcol = 0, [], []
        for t in range(tiles):
            for k in gk[r]:
                valid = [sub for sub in range(r1) if inside(k * r1 + sub, t)]
                if not valid:
                    continue
                assert valid == list(range(valid[0], valid[-1] + 1)), "the rows of a block inside the disc are one range"
                lo[t, k], hi[t, k] = valid[0], valid[-1] + 1
                offb[t, k] = tile * (count - valid[0])
                for sub in valid:
                    prow += [k * r1 + sub] * tile
                    pcol += list(range(tile * t, tile * t + tile))
                count += len(valid)
        S.append(tile * count)
        rows.append(np.array(prow, dtype=np.int64))
        cols.append(np.array(pcol, dtype=np.int64))
    cumS = [sum(S[:r]) for r in range(parts)]
    return dict(n=n, parts=parts, r1=r1, tile=tile, nblk=nblk, tiles=tiles, part_of=part_of, gk=gk, S=S, cumS=cumS,
                total=sum(S), lo=lo, hi=hi, offb=offb, rows=rows, cols=cols)


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

    def staged_paint(self, pos, mass, n, boxsize, window, out, x_start, nx_alloc, offset=0.0, owned=None, hint=None):
        """The same paint in parts (device.StagedPaint): group once, then walk / fold tile row by tile row."""
        return self.dev.StagedPaint(pos, mass, n, boxsize, window, out, x_start=x_start, nx_alloc=nx_alloc, offset=offset,
                                    offset_planes=owned, hint=hint)

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
        return n1 == n2 and self._tile_ok(self.dev.real_code(planes), n1) and parts & (parts - 1) == 0 \
            and n1 // parts >= (32 if n1 == 1024 else 16)

    def spectrum_pitch(self, n, parts):
        """Row pitch (complex elements) of the slab's half-spectrum buffers - the local planes after the z pass, the send
        buffer, the received block: n/2+1 rounded up to whole 128-byte lines where the hand-written passes run (every row
        piece of the y pass and of the axis-0 pass then sits on whole lines; 3 % more on the wire at n = 1024), else n/2+1."""
        nz = n // 2 + 1
        if self.dtype == torch.float32 and self._tile_ok(0, n) and parts & (parts - 1) == 0 \
                and n // parts >= (32 if n == 1024 else 16) and self.axis0_power_supported(n):
            return (nz + 15) // 16 * 16
        return nz

    def lowz_work(self, n, nloc):
        """Work area of the low-k channel for nloc planes; its first nloc * n * 7 complex128 are the z sums [plane][y][kz]
        that fft2d_planes_packed(..., lowz=) fills plane range by plane range."""
        from ._lib import lib
        return torch.empty(int(lib().ast_lowk_work_bytes(n, nloc)) // 16, dtype=torch.complex128, device=self.device)

    def lowk_modes_from_z(self, work, n, x0, nloc):
        from ._lib import check, lib
        out = torch.empty(int(lib().ast_lowk_mode_count()), dtype=torch.complex128, device=self.device)
        check(lib().ast_lowk_modes_from_z(n, int(x0), nloc, 0, self.dev.ptr(out), self.dev.ptr(work), work.numel() * 16,
                                          self.dev.stream()), "ast_lowk_modes_from_z")
        return out

    def fft2d_planes_packed(self, planes, spec, packed, parts, self_part, self_dst, lowz=None):
        """fft2d_planes + pack in two kernels instead of three: the y pass stores straight into the send buffer
        (piece s of `packed`) and the rank's own piece into ``self_dst`` (its place in the receive block).
        lowz: (nplanes * n * 7) complex128 that receives the low-k channel's z sums of these planes from the z pass."""
        from ._lib import check, lib
        nloc, n1, n2 = planes.shape
        code = self.dev.real_code(planes)
        nz, pitch = n2 // 2 + 1, spec.shape[-1]
        assert self_dst.is_contiguous() and packed.is_contiguous() and spec.is_contiguous()
        assert packed.shape[-1] == pitch and self_dst.shape[-1] == pitch
        if lowz is not None:
            assert lowz.is_contiguous() and lowz.dtype == torch.complex128 and lowz.numel() == nloc * n1 * 7
            check(lib().ast_fft_tile_rows_r2c_lowz(self.dev.ptr(planes), self.dev.ptr(spec), code, n2, nloc * n1, n2, pitch,
                                                   1.0, self.dev.ptr(lowz), self.dev.stream()), "ast_fft_tile_rows_r2c_lowz")
        else:
            check(lib().ast_fft_tile_rows_r2c(self.dev.ptr(planes), self.dev.ptr(spec), code, n2, nloc * n1, n2, pitch,
                                              1.0, self.dev.stream()), "ast_fft_tile_rows_r2c")
        check(lib().ast_fft_tile_c2c_packed(self.dev.ptr(spec), self.dev.ptr(packed), code, n1, nz, pitch, nloc, parts,
                                            self_part, self.dev.ptr(self_dst), 1.0, self.dev.stream()),
              "ast_fft_tile_c2c_packed")

    def disc_layout(self, n, parts):
        """The disc layout of the slab transpose for (n, parts) from the library (plane sizes, owners of the row blocks),
        or None where the hand-written passes or the geometry do not allow it (ASTRILD_SLAB_DISC=0: never)."""
        import ctypes as ct
        import os
        from ._lib import check, lib
        if os.environ.get("ASTRILD_SLAB_DISC", "1") == "0" or self.dtype != torch.float32 or not self._tile_ok(0, n):
            return None
        r1 = 32 if n == 1024 else 16
        if (n // r1) % parts:
            return None
        S = (ct.c_uint * parts)()
        owner = (ct.c_ubyte * (n // r1))()
        r1_out = ct.c_int()
        check(lib().ast_fft_tile_disc_layout(n, parts, S, owner, ct.byref(r1_out)), "ast_fft_tile_disc_layout")
        S = [int(v) for v in S]
        return dict(n=n, parts=parts, r1=int(r1_out.value), tile=16, S=S, cumS=[sum(S[:r]) for r in range(parts)], total=sum(S),
                    part_of=[int(v) for v in owner])

    def fft2d_planes_disc(self, planes, spec, packed, layout, self_part, self_dst, lowz=None, halo=None):
        """z rows of the planes into `spec`, then the k_y pass storing in the disc layout: part q's rows into
        packed[npl * cumS[q] : npl * (cumS[q] + S[q])] (what goes to rank q), the rank's own part into self_dst."""
        from ._lib import check, lib
        nloc, n1, n2 = planes.shape
        code = self.dev.real_code(planes)
        pitch = spec.shape[-1]
        assert spec.is_contiguous() and packed.is_contiguous() and self_dst.is_contiguous()
        assert packed.numel() == nloc * layout["total"] and self_dst.numel() == nloc * layout["S"][self_part]
        if halo is not None:
            # the staged paint left these planes' halo records unfolded: the z pass adds them as it loads the rows
            rec, win, row_lo, row_hi, xb0, nx_alloc = halo
            check(lib().ast_fft_tile_rows_r2c_slab_halo(self.dev.ptr(planes), self.dev.ptr(spec), code, n2, nloc * n1, pitch, 1.0, rec, win,
                                                        int(xb0), int(nx_alloc), int(row_lo), int(row_hi),
                                                        self.dev.ptr(lowz) if lowz is not None else None, self.dev.stream()),
                  "ast_fft_tile_rows_r2c_slab_halo")
        elif lowz is not None:
            assert lowz.is_contiguous() and lowz.dtype == torch.complex128 and lowz.numel() == nloc * n1 * 7
            check(lib().ast_fft_tile_rows_r2c_lowz(self.dev.ptr(planes), self.dev.ptr(spec), code, n2, nloc * n1, n2, pitch,
                                                   1.0, self.dev.ptr(lowz), self.dev.stream()), "ast_fft_tile_rows_r2c_lowz")
        else:
            check(lib().ast_fft_tile_rows_r2c(self.dev.ptr(planes), self.dev.ptr(spec), code, n2, nloc * n1, n2, pitch,
                                              1.0, self.dev.stream()), "ast_fft_tile_rows_r2c")
        check(lib().ast_fft_tile_c2c_disc(self.dev.ptr(spec), self.dev.ptr(packed), code, n1, pitch, nloc, layout["parts"],
                                          int(self_part), self.dev.ptr(self_dst), 1.0, self.dev.stream()), "ast_fft_tile_c2c_disc")

    def axis0_power_disc(self, block, scale, n, boxsize, layout, part, psum, first_bin):
        """The last pass over the rank's block in the disc layout fused with its shell binning (ast_fft_tile_disc_block_power);
        psum is overwritten (shells below first_bin left at zero)."""
        from ._lib import check, lib
        parts = layout["parts"]
        key = (n, parts, "disc")
        if getattr(self, "_bp_key", None) != key:
            self._bp_key = key
            self._bp_scratch = torch.empty(int(lib().ast_fft_tile_disc_power_scratch_bytes(n, parts)), dtype=torch.uint8,
                                           device=self.device)
        assert block.is_contiguous() and block.numel() == n * layout["S"][part]
        psum.zero_()
        check(lib().ast_fft_tile_disc_block_power(self.dev.ptr(block), self.dev.ptr(self._bp_scratch), self._bp_scratch.numel(), 0,
                                                  n, parts, int(part), float(scale), float(boxsize), int(first_bin),
                                                  self.dev._bin_code(None), self.dev.ptr(psum), self.dev.stream()),
              "ast_fft_tile_disc_block_power")
        return psum

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


def _peer(group, r):
    """P2POp / send / recv take GLOBAL ranks; r is a rank of `group`."""
    return r if group is None else dist.get_global_rank(group, r)


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

    def _peers(self):
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        return (rank - 1) % world, (rank + 1) % world

    def start(self):
        self.start_upper()
        self.start_lower()

    def start_upper(self):
        """My UPPER ghost planes go to the right neighbour; the left neighbour's upper ghosts arrive for my first planes.
        (Every rank posts the same pair, so the operations of each pair of ranks match up.)"""
        left, right = self._peers()
        comm_ready(self.group)
        self.reqs = (self.reqs or []) + dist.batch_isend_irecv([
            dist.P2POp(dist.isend, self.buf[self.gl + self.nloc:], _peer(self.group, right), self.group),
            dist.P2POp(dist.irecv, self.from_left, _peer(self.group, left), self.group),
        ])

    def start_lower(self):
        """My LOWER ghost planes go to the left neighbour; the right neighbour's lower ghosts arrive for my last planes."""
        left, right = self._peers()
        comm_ready(self.group)
        self.reqs = (self.reqs or []) + dist.batch_isend_irecv([
            dist.P2POp(dist.isend, self.buf[:self.gl], _peer(self.group, left), self.group),
            dist.P2POp(dist.irecv, self.from_right, _peer(self.group, right), self.group),
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


def exchange_planes(packed_c, block, p0, npl, nloc, group=None, self_done=False):
    """Step 4 for the local planes [p0, p0 + npl).  packed_c: (P, npl, nly, nz) - piece s goes to rank s; it lands in
    block[s_src*nloc + p0 : ... + npl] of the receiver.  Returns the outstanding work handles (the local piece is copied
    right away, unless the producer has already written it into the block: ``self_done``).  Every rank calls this for
    the same plane ranges in the same order, so the point-to-point operations of a pair match up."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ops_list = []
    if world > 1:
        comm_ready(group)
    for s in range(world):
        dst = block[s * nloc + p0: s * nloc + p0 + npl]
        if s == rank:
            if not self_done:
                dst.copy_(packed_c[s])
            continue
        # complex payload moved as (re, im) pairs of the real dtype: every c10d backend takes that
        ops_list.append(dist.P2POp(dist.isend, torch.view_as_real(packed_c[s]), _peer(group, s), group))
        ops_list.append(dist.P2POp(dist.irecv, torch.view_as_real(dst), _peer(group, s), group))
    return dist.batch_isend_irecv(ops_list) if ops_list else []


def exchange_planes_disc(packed, block, p0, npl, nloc, layout, group=None):
    """Step 4 in the disc layout for the local planes [p0, p0 + npl).  packed: flat, npl * total elements - the piece for
    rank s is packed[npl * cumS[s] : npl * (cumS[s] + S[s])] (npl planes of S[s]); it lands in planes s_src * nloc + p0 ... of
    the receiver's block (n, S[receiver]).  The rank's own piece was written into the block by the producer."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    S, cumS = layout["S"], layout["cumS"]
    ops_list = []
    if world > 1:
        comm_ready(group)
    for s in range(world):
        if s == rank:
            continue
        piece = packed[npl * cumS[s]: npl * (cumS[s] + S[s])]
        dst = block[s * nloc + p0: s * nloc + p0 + npl]
        ops_list.append(dist.P2POp(dist.isend, torch.view_as_real(piece), _peer(group, s), group))
        ops_list.append(dist.P2POp(dist.irecv, torch.view_as_real(dst), _peer(group, s), group))
    return dist.batch_isend_irecv(ops_list) if ops_list else []


def exchange_chunk(packed_c, block, chunk, pc, nloc, group=None, self_done=False):
    """exchange_planes for chunk `chunk` of `pc` local planes."""
    return exchange_planes(packed_c, block, chunk * pc, pc, nloc, group, self_done)


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


class Watchdog:
    """First contact with RCCL must not hang a node: a daemon thread that ENDS THE PROCESS (os._exit - a fresh exit, never a
    re-exec) with the rank, the host-side stage and the last schedule entry the GPU has completed on stderr when ``beat()``
    has not been called for ``timeout_s`` seconds.  ``watch(pipe)`` names the pipeline whose ``stage_name`` / progress
    markers are reported."""

    def __init__(self, timeout_s=None, rank=0, exit_code=3):
        import os
        import threading
        import time
        self.timeout = float(timeout_s if timeout_s is not None else os.environ.get("ASTRILD_SLAB_TIMEOUT_S", "240"))
        self.rank, self.exit_code, self.pipe, self.note = rank, exit_code, None, "start"
        self.last = time.monotonic()
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, name="astrild-slab-watchdog", daemon=True)
        self._thread.start()

    def watch(self, pipe):
        self.pipe = pipe

    def beat(self, note=None):
        import time
        self.last = time.monotonic()
        if note is not None:
            self.note = note

    def stop(self):
        self._stop.set()

    def report(self):
        pipe = self.pipe
        host = getattr(pipe, "stage_name", None)
        dev = pipe.progress_report() if pipe is not None and hasattr(pipe, "progress_report") else None
        return f"[astrild slab watchdog] rank {self.rank}: no progress for {self.timeout:.0f} s after '{self.note}'; " \
               f"host stage: {host}; device: {dev}"

    def _run(self):
        import os
        import sys
        import time
        while not self._stop.wait(1.0):
            if time.monotonic() - self.last > self.timeout:
                try:
                    sys.stderr.write(self.report() + "\n")
                    sys.stderr.flush()
                    import faulthandler
                    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)     # where the host threads stand
                    sys.stderr.flush()
                finally:
                    os._exit(self.exit_code)


class SlabPowerPipeline:
    """CIC/TSC + slab FFT + P(k) for the synthetic lattice workload of bench.py."""

    def __init__(self, n, boxsize, npside, window="cic", dtype=torch.float32, seed=20240601, shuffle=False,
                 ghost=4, ops=None, group=None, pos=None, chunks=None, route=False, pipeline=None, rows_per_stage=None,
                 xsorted=None, group_chunks=None):
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
        pitch_fn = getattr(o, "spectrum_pitch", None)
        self.nzp = pitch_fn(n, P) if pitch_fn else self.nz          # row pitch of the half-spectrum buffers (>= nz)
        self.spec2d = o.empty((self.nloc, n, self.nzp), o.cdtype)
        if chunks is None:
            chunks = 4 if self.nloc % 4 == 0 and self.nloc >= 16 else 1
        if self.nloc % chunks:
            raise ValueError(f"{self.nloc} local planes do not split into {chunks} chunks")
        self.chunks = chunks
        self.pc = self.nloc // chunks
        # the transpose's wire format: the disc layout (only what FFTPower keeps, owners balanced by disc area) where the
        # ops offer it, else `parts` blocks of whole pitched rows
        disc_fn = getattr(o, "disc_layout", None)
        self.disc = disc_fn(n, P) if disc_fn else None
        if self.disc is not None:
            self.packed = o.empty((self.nloc * self.disc["total"],), o.cdtype)
            self.block = o.empty((n, self.disc["S"][self.rank]), o.cdtype)
        else:
            self.packed = o.empty((chunks, P, self.pc, self.nloc, self.nzp), o.cdtype)
            self.block = o.empty((n, self.nloc, self.nzp), o.cdtype)
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
        # "staged" (default where the ops can paint in parts and more than one rank takes part): the slab is painted,
        # transformed and sent plane range by plane range - group once, then per stage walk -> fold -> z rows -> y pass in
        # send order -> grouped send, ghost rows first - so the exchange starts after the grouping and one stage instead
        # of after the whole paint.  "bulk": paint everything, then transform and send chunk by chunk (kept for A/B runs
        # on hardware: ASTRILD_SLAB_PIPELINE=bulk).
        import os
        if pipeline is None:
            pipeline = os.environ.get("ASTRILD_SLAB_PIPELINE") or ("staged" if P > 1 and hasattr(o, "staged_paint") else "bulk")
        if pipeline not in ("staged", "bulk"):
            raise ValueError(pipeline)
        if pipeline == "staged" and not hasattr(o, "staged_paint"):
            raise ValueError("these ops cannot paint in stages")
        self.pipeline = pipeline
        self.rows_per_stage = int(rows_per_stage or os.environ.get("ASTRILD_SLAB_ROWS_PER_STAGE") or 0)
        # grouping in parts (staged, P > 1): only for particles that come in ascending x within `ghost` planes of a uniform
        # lattice - the synthetic set in natural order, or `pos` with xsorted=True (a violation is counted as dropped
        # particles and step(check=True) raises).  ASTRILD_SLAB_GROUP_CHUNKS=1 groups everything first, as in the bulk order.
        if xsorted is None:
            xsorted = pos is None and not shuffle and not route
        def parts_ok(want):
            return pipeline == "staged" and P > 1 and xsorted and not self.rows_per_stage and want > 1 and self.nloc % want == 0 \
                and self.nloc // want >= 2 * (ghost + 1) and hasattr(o, "staged_paint")
        asked = int(group_chunks or os.environ.get("ASTRILD_SLAB_GROUP_CHUNKS") or 0)
        if asked:
            self.group_chunks = asked if parts_ok(asked) else 1
        else:
            # 8 equal parts (else 4) where the slab allows it - a part must be deeper than the particles' reach; the stages
            # take unequal numbers of them (_make_schedule).  16 parts forecast the same within the box-to-box spread
            self.group_chunks = next((k for k in (8, 4) if xsorted and parts_ok(k)), 1)
        # ASTRILD_SLAB_STAGES="7|0,1,2,3|4,5|6": which of the group_chunks equal parts each stage groups (consecutive parts of
        # a stage go in one launch) - stages of unequal size; default: see _make_schedule
        self.group_stages = None
        spec = os.environ.get("ASTRILD_SLAB_STAGES")
        if spec and self.group_chunks > 1:
            self.group_stages = [[int(v) for v in st.split(",")] for st in spec.split("|")]
        self.send_planes = max(1, int(os.environ.get("ASTRILD_SLAB_SEND_PLANES") or 24))
        self.staged = None
        self.schedule = None
        self.stage_name = "init"          # what the rank is doing (read by bench.py's watchdog)
        self.trace = None                 # a list: (entry, event recorded after it was enqueued) per staged entry (perf scripts)
        self._progress = None             # pinned host words the GPU writes after each schedule entry (step(progress=True))
        # the rank's OWN planes hold rho - mean (subtracted before the fp32 rounding); its ghost planes, which are
        # added onto the neighbours' cells, stay plain sums
        lowk_fn = getattr(self.ops, "lowk_supported", None)
        self.lowk = bool(lowk_fn and lowk_fn(n))
        # the low-k channel's z sums come out of the FFT's z pass (no second read of the planes) where the packed
        # transform runs; ASTRILD_SLAB_LOWK_SEPARATE=1 keeps the separate kernels on a side stream
        self.lowz = None
        if self.lowk and hasattr(o, "lowz_work") and not os.environ.get("ASTRILD_SLAB_LOWK_SEPARATE") \
                and o.packed_supported(torch.empty((1, n, n), dtype=o.dtype, device="meta"), P):
            self.lowz = o.lowz_work(n, self.nloc)
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

    # ------------------------------------------------------------------ staged pipeline
    def _make_schedule(self, sp):
        """The static order of a staged step - the same on every rank (it depends on the geometry only), which is what
        lets the ranks' point-to-point operations match up.  Entries: ("group",) or ("reset",) and ("group_part", k, K,
        closed_row0, closed_nrows, span); ("walk", row0, nrows); ("fold", row0, nrows); ("ghost_start",) or ("ghost_start_upper",) /
        ("ghost_start_lower",); ("ghost_finish",); ("fft", p0, npl) with p0 counted in OWNED planes."""
        R, PR = sp.nrows_total, sp.row_planes
        gl, gh, nloc = self.gl, self.gh, self.nloc
        lo, hi = gl, gl + nloc                                       # owned planes in buffer coordinates
        ghost_in = set(range(lo, lo + gh)) | set(range(hi - gl, hi)) if self.world > 1 else set()
        sched, walked, folded, sent = [], set(), set(), set()
        state = {"ghosts_in": self.world == 1}

        def runs(items):
            items = sorted(items)
            out = []
            for v in items:
                if out and v == out[-1][0] + out[-1][1]:
                    out[-1][1] += 1
                else:
                    out.append([v, 1])
            return out

        def walk(rows):
            for r0, nr in runs(set(rows) - walked):
                sched.append(("walk", r0, nr))
            walked.update(rows)

        def fold_ready(only=None):
            ready = [r for r in range(R) if r not in folded and (only is None or r in only) and set(sp.fold_needs(r)) <= walked]
            for r0, nr in runs(ready):
                sched.append(("fold", r0, nr))
            folded.update(ready)
            return ready

        def fft_ready():
            planes = [p for r in folded for p in range(r * PR, min((r + 1) * PR, self.nx_alloc))
                      if lo <= p < hi and p not in sent and (state["ghosts_in"] or p not in ghost_in)]
            for p0, npl in runs(planes):
                # at most send_planes planes per transform + send: a piece travels while the next one is transformed, and
                # what is left on the links after the last transform is one small piece
                while npl > 0:
                    take = min(npl, self.send_planes)
                    sched.append(("fft", p0 - lo, take))
                    p0, npl = p0 + take, npl - take
            sent.update(planes)

        upper_rows = {p // PR for p in range(hi, self.nx_alloc)}     # tile rows that hold the ghost planes
        lower_rows = {p // PR for p in range(0, gl)}
        K = self.group_chunks
        if K > 1:
            # GROUPING IN PARTS (x-ordered particles): part k holds the particles of the owned planes [k nloc / K,
            # (k + 1) nloc / K) of the lattice they started from, displaced by at most `ghost` planes; its particles' tiles lie
            # in the rows key_rows(k).  A row is walked once every part that can hold one of its particles is grouped.  The
            # LAST part goes first: it completes the rows of the upper ghost planes, whose exchange can then start; then
            # parts 0, 1, ... with the rows they complete - walk, fold, transform, send - so the first planes reach the links
            # after a quarter of the grouping instead of all of it.
            ghost = gl - 1

            def key_rows(k):
                first = (gl - ghost + k * nloc // K) // PR
                last = (gl + ghost + (k + 1) * nloc // K - 1) // PR
                return range(max(0, first), min(R - 1, last) + 1)

            grouped = set()

            def group(k, span=1):
                # parts k .. k + span - 1 in ONE launch.  The walked rows form one range modulo R: a block that ends at
                # R - 1 (walked first) and a block that starts at 0
                high = sorted(r for r in walked if all(q in walked for q in range(r, R)))
                row0 = high[0] if high else 0
                sched.append(("group_part", k, K, row0 if walked else 0, len(walked), span))
                grouped.update(range(k, k + span))
                assert not walked or set((row0 + i) % R for i in range(len(walked))) == walked, "walked rows must stay one cyclic range"

            def walk_complete():
                ok = [r for r in range(R) if r not in walked and all(k in grouped for k in range(K) if r in key_rows(k))]
                # keep the walked set one cyclic range: the top block grows downwards only as far as it is contiguous with
                # R - 1, the bottom block upwards from 0
                top, r = [], R - 1
                while r in ok or r in walked:
                    if r in ok:
                        top.append(r)
                    r -= 1
                bottom, r = [], 0
                while (r in ok or r in walked) and r not in top:
                    if r in ok:
                        bottom.append(r)
                    r += 1
                walk(sorted(top))
                walk(bottom)

            sched.append(("reset",))
            started = {"upper": False, "lower": False}
            # parts per stage: the last part alone (the rows of the upper ghost planes), then two at a time, and the
            # LAST stage one part again - what is still to be transformed and sent after the last walk is then small
            stages = self.group_stages
            if stages is None:
                # FIRST the top parts that complete the tile rows of the upper ghost planes (their exchange starts, the first
                # planes leave; the ghost planes share the two links to the ring neighbours with a seventh of the spectrum and
                # must be in before the last planes can be transformed).  Then the parts 0, 1, ... in stages of DECREASING
                # size, one grouping launch each: a stage's planes travel while the next stage is grouped and walked, and what
                # is still to be transformed and sent after the last walk is small.
                # 1024^3 on 8 ranks (scripts/perf_slab_staged.py with STAGE_SPECS=..., forecast at 60 / 50 GB/s per link
                # with the ghost planes on the neighbour links, two boxes): 4 parts 3 | 0 1 | 2: 5.21-5.22x / 4.73x;  8 parts (the
                # default) 7 | 0 1 2 | 3 4 5 | 6: 5.35-5.45x / 4.99-5.03x;  16 parts 14 15 | 0-5 | 6-10 | 11 12 | 13: 5.32-5.40x /
                # 5.03-5.05x (14 15 | 0-4 | 5-8 | 9-12 | 13: 5.26x / 5.09x; 15 | 0-4 | 5-9 | 10-13 | 14, whose upper ghost planes
                # leave last: 4.91x / 4.63x);  everything grouped first: 5.16x / 4.94x.
                need = set()
                for r in upper_rows:
                    need.update(sp.fold_needs(r))
                first = [K - 1]
                while first[0] > 1 and need & set(key_rows(first[0] - 1)):
                    first.insert(0, first[0] - 1)
                m = first[0]                                            # parts 0 .. m - 1 remain
                sizes = {(16, 14): [6, 5, 2, 1], (8, 7): [3, 3, 1], (4, 3): [2, 1]}.get((K, m))
                if sizes is None:                                       # other geometries: stages of <= 5 parts, the last part alone
                    ng = max(1, -(-(m - 1) // 5))
                    cut = [-(-i * (m - 1) // ng) for i in range(ng + 1)]
                    sizes = [cut[i + 1] - cut[i] for i in range(ng) if cut[i + 1] > cut[i]] + [1]
                stages, k0 = [first], 0
                for sz in sizes:
                    stages.append(list(range(k0, k0 + sz)))
                    k0 += sz
                stages = [st for st in stages if st]
            assert sorted(k for st in stages for k in st) == list(range(K)), "every part in exactly one stage"
            for stage in stages:
                for k0, span in runs(stage):          # consecutive parts of a stage: one grouping launch
                    group(k0, span)
                walk_complete()
                fold_ready()
                if not started["upper"] and upper_rows <= folded:
                    sched.append(("ghost_start_upper",))
                    started["upper"] = True
                if not started["lower"] and lower_rows <= folded:
                    sched.append(("ghost_start_lower",))
                    started["lower"] = True
                fft_ready()
            assert started["upper"] and started["lower"]
        else:
            sched.append(("group",))
            if self.world > 1:
                first = upper_rows | lower_rows
                need = set()
                for r in first:
                    need.update(sp.fold_needs(r))
                walk(need)
                fold_ready(only=first)
                sched.append(("ghost_start",))
                fold_ready()
                fft_ready()
            rest = [r for r in range(R) if r not in walked]
            per = self.rows_per_stage or max(1, (len(rest) + 3) // 4)
            # the FIRST stage is half as long (when rows_per_stage is not given): its planes reach the links sooner, and the
            # exchange - the longer leg below ~70 GB/s per link - is serialised behind its first send
            sizes = []
            left = len(rest)
            if not self.rows_per_stage and per >= 2 and self.world > 1:
                sizes.append(per // 2)
                left -= per // 2
            while left > 0:
                sizes.append(min(per, left))
                left -= sizes[-1]
            i = 0
            for size in sizes:
                walk(rest[i:i + size])
                i += size
                fold_ready()
                fft_ready()
        assert walked == set(range(R)) and folded == set(range(R))
        if self.world > 1:
            sched.append(("ghost_finish",))
            state["ghosts_in"] = True
            fft_ready()
        assert sent == set(range(lo, hi))
        return sched

    def progress_report(self):
        """Which schedule entries the GPU streams have completed (from the markers of a step(progress=True))."""
        if self._progress is None or self.schedule is None:
            return "no progress markers"
        done = [int(v) for v in self._progress.tolist()]
        names = ["%s%s" % (e[0], tuple(e[1:])) for e in self.schedule]

        def name(i):
            return "nothing yet" if i <= 0 else names[i - 1] if i <= len(names) else str(i)
        return f"paint stream completed #{done[0]} ({name(done[0])}), transform stream completed #{done[1]} ({name(done[1])}) " \
               f"of {len(names)} entries"

    def _step_staged(self, check=False, progress=False):
        import os
        import time
        o = self.ops
        if self.staged is None:
            self.staged = o.staged_paint(self.pos, None, self.n, self.L, self.window, self.buf, self.x_start, self.nx_alloc,
                                         offset=self.mean_offset, owned=(self.gl, self.nloc) if self.mean_offset else None)
            self.schedule = self._make_schedule(self.staged)
            self._schedule_checked = False
            # (HIP ops, disc layout) only the tile rows that hold ghost planes are folded by the paint - their planes travel
            # before any transform; every other row's halo records are added by the z pass that loads its planes
            # (ASTRILD_SLAB_DEFER_FOLD=0: every row folded by the paint's own kernel, as in round 4)
            self._defer_fold = False
            if self.disc is not None and hasattr(self.staged, "defer_folds") and os.environ.get("ASTRILD_SLAB_DEFER_FOLD", "1") != "0":
                PR = self.staged.row_planes
                ghost_rows = set()
                if self.world > 1:
                    ghost_rows = {p // PR for p in range(0, self.gl)} | {p // PR for p in range(self.gl + self.nloc, self.nx_alloc)}
                self.staged.defer_folds(ghost_rows)
                self._defer_fold = True
            self.packed_flat = self.packed.reshape(-1)
            # ASTRILD_SLAB_STREAMS=2: the paint stages (walk, fold) on the caller's stream, the transforms and the exchange
            # of finished plane ranges on a second one, so that a stage's transform runs beside the next stage's walk.
            # Measured on one GPU (rank 3 of 8, no exchange): 2.28 against 2.22 ms per step on ONE stream - both kinds of
            # kernel are bandwidth bound and gain nothing from each other's company; one stream is the default (RCCL moves
            # the data on its own stream either way)
            self._fft_stream = None
            if self.buf.is_cuda and os.environ.get("ASTRILD_SLAB_STREAMS", "1") == "2":
                self._fft_stream = torch.cuda.Stream()
        sp = self.staged
        if check and not self._schedule_checked and self.world > 1:
            # the ranks' point-to-point operations only match up if every rank runs the SAME schedule (and the same wire
            # layout): before the first checked step posts any, a digest of both is compared across the ranks
            import hashlib
            text = repr(self.schedule) + repr(None if self.disc is None else (self.disc["S"], self.disc["part_of"]))
            digest = int.from_bytes(hashlib.sha256(text.encode()).digest()[:7], "little")
            both = torch.tensor([digest, -digest], dtype=torch.int64, device=self.psum.device)
            comm_ready(self.group)
            dist.all_reduce(both, op=dist.ReduceOp.MAX, group=self.group)
            if int(both[0].item()) != digest or int(both[1].item()) != -digest:
                raise RuntimeError(f"rank {self.rank}: the staged schedule differs between the ranks (digest {digest})")
            self._schedule_checked = True
        owned = self.buf[self.gl: self.gl + self.nloc] if self.world > 1 else self.buf
        P, nloc, nz = self.world, self.nloc, self.nzp
        pending = []
        state = {"modes": None, "side": None}
        packed_fn = getattr(o, "packed_supported", None)
        main = torch.cuda.current_stream() if self.buf.is_cuda else None
        side = self._fft_stream

        if progress and self.buf.is_cuda and self._progress is None:
            self._progress = torch.zeros(2, dtype=torch.int32).pin_memory()
            self._progress_ids = torch.arange(0, len(self.schedule) + 2, dtype=torch.int32, device=self.buf.device)
        if progress and self._progress is not None:
            self._progress.zero_()
        counter = {"i": 0}

        def mark(entry, lane=0):
            if entry[0] != "begin":
                counter["i"] += 1
            if progress and self._progress is not None and entry[0] != "begin":
                # a 4-byte device-to-host copy in stream order: the host sees how far each stream has come
                self._progress[lane:lane + 1].copy_(self._progress_ids[counter["i"]:counter["i"] + 1], non_blocking=True)
            if self.trace is not None and self.buf.is_cuda:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream())
                self.trace.append((entry, ev))

        def transform(p0, npl):
            planes = owned[p0:p0 + npl]
            spec = self.spec2d[p0:p0 + npl]
            if self.disc is not None:
                tot = self.disc["total"]
                packed = self.packed_flat[p0 * tot:(p0 + npl) * tot]
                mine = self.block[self.rank * nloc + p0: self.rank * nloc + p0 + npl]
                halo = None
                if self._defer_fold:
                    rec, win, row_lo, row_hi = sp.halo_args()
                    halo = (rec, win, row_lo, row_hi, (self.gl if self.world > 1 else 0) + p0, self.nx_alloc)
                o.fft2d_planes_disc(planes, spec, packed, self.disc, self.rank, mine,
                                    lowz=self._lowz_rows(p0, npl) if self.lowz is not None else None, halo=halo)
                return exchange_planes_disc(packed, self.block, p0, npl, nloc, self.disc, self.group)
            nly = self.n // P
            packed = self.packed_flat[P * p0 * nly * nz: P * (p0 + npl) * nly * nz].view(P, npl, nly, nz)
            if packed_fn and packed_fn(planes, P):
                mine = self.block[self.rank * nloc + p0: self.rank * nloc + p0 + npl]
                if self.lowz is not None:
                    o.fft2d_planes_packed(planes, spec, packed, P, self.rank, mine, lowz=self._lowz_rows(p0, npl))
                else:
                    o.fft2d_planes_packed(planes, spec, packed, P, self.rank, mine)
                return exchange_planes(packed, self.block, p0, npl, nloc, self.group, self_done=True)
            o.fft2d_planes(planes, spec)
            o.pack(spec, packed, P)
            return exchange_planes(packed, self.block, p0, npl, nloc, self.group)

        if side is not None:
            side.wait_stream(main)            # the previous step's consumers of spec2d / packed / block are on `main`
        mark(("begin",))
        for entry in self.schedule:
            kind = entry[0]
            t0 = time.perf_counter()
            self.stage_name = "%s%s" % (kind, tuple(entry[1:]))
            if kind == "walk":
                sp.walk(entry[1], entry[2])
            elif kind == "fold":
                sp.fold(entry[1], entry[2])
            elif kind == "group":
                sp.group()
            elif kind == "reset":
                sp.reset()
            elif kind == "group_part":
                sp.group_part(entry[1], entry[2], entry[3], entry[4], entry[5])
            elif kind == "ghost_start":
                self.ghosts.start()
            elif kind == "ghost_start_upper":
                self.ghosts.start_upper()
            elif kind == "ghost_start_lower":
                self.ghosts.start_lower()
            elif kind == "ghost_finish":
                self.ghosts.finish()
                self._lowk_start(owned, state)
            elif side is None:
                pending += transform(entry[1], entry[2])
            else:
                side.wait_stream(main)        # the planes' folds (and the ghost add) are on `main`
                with torch.cuda.stream(side):
                    pending += transform(entry[1], entry[2])
                    mark(entry, 1)
            self._tick(kind + ".enqueue", t0)
            if kind != "fft" or side is None:
                mark(entry)
        if self.world == 1:
            self._lowk_start(owned, state)
        if self.lowz is not None:             # every plane's z sums are in: the low-k modes' y and x sums
            self._lowk_from_z(state, side)
        if check:
            sp.check()
        if side is not None:
            main.wait_stream(side)
        self.stage_name = "exchange.wait"
        t0 = time.perf_counter()
        for work in pending:
            work.wait()
        self._tick("exchange.wait", t0)
        self.stage_name = "axis0"
        return self._finish_step(state)

    def _lowz_rows(self, p0, npl):
        """The z sums of the owned planes [p0, p0 + npl) inside the low-k work area ([plane][y][kz])."""
        return self.lowz[p0 * self.n * 7:(p0 + npl) * self.n * 7]

    def _lowk_from_z(self, state, after):
        """The y and x sums of the low-k modes from the z sums the z passes left (a few latency-bound launches on 15 MB):
        on a side stream behind stream `after` (default: the current one), beside the exchange wait and the axis-0 pass."""
        if not self.buf.is_cuda:
            state["modes"] = self.ops.lowk_modes_from_z(self.lowz, self.n, self.rank * self.nloc, self.nloc)
            return
        if self._side is None:
            self._side = torch.cuda.Stream()
        self._side.wait_stream(after if after is not None else torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            state["modes"], state["side"] = self.ops.lowk_modes_from_z(self.lowz, self.n, self.rank * self.nloc, self.nloc), self._side

    def _lowk_start(self, owned, state):
        """The lowest shells from double-precision DFT sums of the rank's own (complete) planes
        (device.power_sums_fused's low-k channel, split over the slabs): one more all-reduce, of 1183 complex numbers.
        It only reads the planes: on the GPU it runs on a side stream beside the FFT chunks and their exchange."""
        if not self.lowk or self.lowz is not None:       # (lowz: the z pass of every plane range leaves the z sums)
            return
        if owned.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream()
            state["side"] = self._side
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                state["modes"] = self.ops.lowk_modes(owned, self.n, self.rank * self.nloc)
        else:
            state["modes"] = self.ops.lowk_modes(owned, self.n, self.rank * self.nloc)

    def _finish_step(self, state, block=None):
        """Axis-0 pass + shell binning of the received block, all-reduces, low-k patch."""
        import time
        fused_fn = getattr(self.ops, "axis0_power_supported", None)
        if self.disc is not None:
            self.ops.axis0_power_disc(self.block, 1.0 / float(self.n) ** 3, self.n, self.L, self.disc, self.rank, self.psum,
                                      5 if self.lowk else 0)
        elif fused_fn and fused_fn(self.n):
            # axis-0 pass and shell binning in one kernel (the spectrum block is not written back)
            self.ops.fft1d_axis0_power(self.block, 1.0 / float(self.n) ** 3, self.n, self.L, self.rank * self.nloc, self.psum,
                                       5 if self.lowk else 0)
        else:
            blk = self.ops.fft1d_axis0(self.block, 1.0 / float(self.n) ** 3)
            self.ops.power_bin(blk, self.n, self.L, self.i0, self.i1, self.psum)
        modes, side = state["modes"], state["side"]
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            modes.record_stream(torch.cuda.current_stream())      # allocated under the side stream, used from here on
        self.stage_name = "allreduce"
        t0 = time.perf_counter()
        comm_ready(self.group)
        dist.all_reduce(self.psum, group=self.group)
        if modes is not None:
            modes_r = torch.view_as_real(modes)
            dist.all_reduce(modes_r, group=self.group)
            self.ops.lowk_patch(modes, self.n, self.L, self.psum)
        self._tick("allreduce.enqueue", t0)
        self.stage_name = "idle"
        return self.ksum, self.psum, self.nmodes

    def wire_bytes(self):
        """What one step of this rank puts on the links: ghost planes to the two ring neighbours, its pieces of the
        half spectrum to the P - 1 peers, and the all-reduced sums."""
        if self.world == 1:
            return {"ghost": 0, "transpose": 0, "allreduce": 0}
        esz = torch.empty((), dtype=self.ops.dtype).element_size() if hasattr(self.ops, "dtype") else 4
        if self.disc is not None:         # sent: the other parts' planes; received: the own part's planes from the others
            sent = self.nloc * (self.disc["total"] - self.disc["S"][self.rank]) * 2 * esz
            recv = (self.world - 1) * self.nloc * self.disc["S"][self.rank] * 2 * esz
        else:
            sent = recv = (self.world - 1) * self.nloc * self.nloc * self.nzp * 2 * esz
        return {"ghost": (self.gl + self.gh) * self.n * self.n * esz,
                "transpose": sent, "transpose_received": recv,
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
            if self.disc is not None:
                tot, p0 = self.disc["total"], c * self.pc
                packed = self.packed[p0 * tot:(p0 + self.pc) * tot]
                mine = self.block[self.rank * self.nloc + p0: self.rank * self.nloc + p0 + self.pc]
                o.fft2d_planes_disc(planes, spec, packed, self.disc, self.rank, mine,
                                    lowz=self._lowz_rows(p0, self.pc) if self.lowz is not None else None)
                pending += exchange_planes_disc(packed, self.block, p0, self.pc, self.nloc, self.disc, self.group)
                continue
            packed_fn = getattr(o, "packed_supported", None)
            if packed_fn and packed_fn(planes, self.world):
                # y pass stores in send order; the rank's own piece goes straight into the receive block
                mine = self.block[self.rank * self.nloc + c * self.pc: self.rank * self.nloc + (c + 1) * self.pc]
                if self.lowz is not None:
                    o.fft2d_planes_packed(planes, spec, self.packed[c], self.world, self.rank, mine,
                                          lowz=self._lowz_rows(c * self.pc, self.pc))
                else:
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

    def step(self, check=False, progress=False):
        """One paint -> FFT -> P(k) step.  check: synchronise and raise if a deposit fell outside the ghost zone.
        progress (staged, GPU): the streams report every completed schedule entry to pinned host memory (a 4-byte copy
        each) for the watchdog - for the first, untimed step."""
        if self.pipeline == "staged":
            return self._step_staged(check, progress)
        import time
        t0 = time.perf_counter()
        self.stage_name = "paint"
        owned = self.paint(check, fold=False)
        self._tick("paint.enqueue", t0)
        state = {"modes": None, "side": None}

        def finish_ghosts():
            t1 = time.perf_counter()
            if self.ghosts is not None:
                self.ghosts.finish()
            self._tick("ghost.wait+add", t1)
            self._lowk_start(owned, state)

        self.stage_name = "fft2d+exchange"
        self.forward_fft(owned, last_pass=False, before_edge=finish_ghosts)
        if self.lowz is not None:
            self._lowk_from_z(state, None)
        self.stage_name = "axis0"
        return self._finish_step(state)
