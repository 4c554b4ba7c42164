"""Numpy double of HipSlabOps for the CPU (gloo) tests of the slab pipeline's
collective logic.  Test infrastructure: built on the oracle."""
import numpy as np
import torch

from oracle import fftpower as offt, mesh as omesh


class NumpySlabOps:
    dtype = torch.float64
    cdtype = torch.complex128
    device = torch.device("cpu")

    def zeros(self, shape, dtype=None):
        return torch.zeros(shape, dtype=dtype or self.dtype)

    def empty(self, shape, dtype=None):
        return torch.zeros(shape, dtype=dtype or self.dtype)

    def paint(self, pos, mass, n, boxsize, window, out, x_start, nx_alloc, check=False):
        full = omesh.paint(pos.numpy(), None if mass is None else mass.numpy(), n, boxsize, window)
        planes = (x_start + np.arange(nx_alloc)) % n
        out.copy_(torch.from_numpy(full[planes]))
        if check and not np.isclose(full.sum(), full[planes].sum(), rtol=1e-12):
            raise RuntimeError("deposits fell outside the slab buffer")
        return out

    def staged_paint(self, pos, mass, n, boxsize, window, out, x_start, nx_alloc, offset=0.0, owned=None, hint=None):
        return StagedPaintDouble(self, pos, mass, n, boxsize, window, out, x_start, nx_alloc, offset, owned)

    def add_into(self, dst, src):
        dst += src

    def fft2d_planes(self, planes, out):
        out.copy_(torch.from_numpy(np.fft.rfft2(planes.numpy(), axes=(1, 2))))
        return out

    def pack(self, spec, out, parts):
        n0, n1, n2 = spec.shape
        out.copy_(spec.reshape(n0, parts, n1 // parts, n2).permute(1, 0, 2, 3).contiguous().reshape(out.shape))
        return out

    # ---- the disc wire format (slab.disc_layout): only what FFTPower keeps travels, owners balanced by disc area
    disc_geometry = None               # (r1, tile) switches it on

    def disc_layout(self, n, parts):
        if self.disc_geometry is None:
            return None
        from astrild_amd import slab
        r1, tile = self.disc_geometry
        return slab.disc_layout(n, parts, r1, tile)

    def fft2d_planes_disc(self, planes, spec, packed, layout, self_part, self_dst, lowz=None, halo=None):
        npl, n, _ = planes.shape
        full = np.fft.rfft2(planes.numpy(), axes=(1, 2))
        spec.copy_(torch.from_numpy(full))
        tiles, tile = layout["tiles"], layout["tile"]
        padded = np.zeros((npl, n, tiles * tile), dtype=full.dtype)
        padded[:, :, :full.shape[2]] = full
        # (what lies outside the disc is in no piece: it never reaches the other side)
        for q in range(layout["parts"]):
            piece = padded[:, layout["rows"][q], layout["cols"][q]]                 # (npl, S[q]) in the part's layout order
            if q == self_part:
                self_dst.copy_(torch.from_numpy(piece).reshape(self_dst.shape))
            else:
                a = npl * layout["cumS"][q]
                packed[a:a + npl * layout["S"][q]] = torch.from_numpy(piece).reshape(-1)

    def axis0_power_disc(self, block, scale, n, boxsize, layout, part, psum, first_bin):
        tiles, tile = layout["tiles"], layout["tile"]
        full = np.zeros((n, n, tiles * tile), dtype=np.complex128)                  # rows of other parts / outside the disc: zero
        full[:, layout["rows"][part], layout["cols"][part]] = block.numpy().reshape(n, -1)
        full = np.fft.fft(full[:, :, :n // 2 + 1], axis=0) * scale
        p3d = (full * full.conj()).real * boxsize ** 3
        _, ps, _ = offt.project_block(p3d, n, boxsize, 0, 0)
        psum.copy_(torch.from_numpy(ps))
        return psum

    def fft1d_axis0(self, block, scale):
        block.copy_(torch.from_numpy(np.fft.fft(block.numpy(), axis=0) * scale))
        return block

    def shell_geometry(self, n, boxsize, i0, i1):
        z = np.zeros((i0[1], i1[1], n // 2 + 1))
        ks, _, nm = offt.project_block(z, n, boxsize, i0[0], i1[0])
        return torch.from_numpy(ks), torch.from_numpy(nm)

    def power_bin(self, block, n, boxsize, i0, i1, psum):
        b = block.numpy()
        p3d = (b * b.conj()).real * boxsize ** 3
        _, ps, _ = offt.project_block(p3d, n, boxsize, i0[0], i1[0])
        psum.copy_(torch.from_numpy(ps))
        return psum

    @staticmethod
    def _dest(pos, n, boxsize, window, parts):
        sgrid = pos.numpy()[:, 0] * (n / boxsize)
        base = np.floor(sgrid if window == "cic" else sgrid + 0.5)
        return (np.mod(base, n).astype(np.int64)) // (n // parts)

    def route_count(self, pos, n, boxsize, window, parts):
        return torch.from_numpy(np.bincount(self._dest(pos, n, boxsize, window, parts), minlength=parts).astype(np.int64))

    def route_scatter(self, pos, mass, n, boxsize, window, parts, counts):
        order = np.argsort(self._dest(pos, n, boxsize, window, parts), kind="stable")
        return pos[torch.from_numpy(order)].contiguous(), None if mass is None else mass[torch.from_numpy(order)].contiguous()


class StagedPaintDouble:
    """device.StagedPaint for the CPU tests: tile rows of ROW_PLANES buffer planes; a row's planes hold NaN until the
    row is FOLDED (and folding asserts that the rows it depends on have been walked), so a pipeline that transforms
    or sends a plane too early poisons its result."""
    ROW_PLANES = 2

    def __init__(self, ops, pos, mass, n, boxsize, window, out, x_start, nx_alloc, offset, owned):
        self.ops, self.pos, self.mass, self.n, self.L, self.window = ops, pos, mass, n, boxsize, window.lower()
        self.out, self.x_start, self.nx, self.offset, self.owned = out, x_start, nx_alloc, offset, owned
        self.row_planes = self.ROW_PLANES
        self.nrows_total = (nx_alloc + self.row_planes - 1) // self.row_planes
        self.periodic = nx_alloc == n and x_start == 0
        self.walked, self.folded, self.full = set(), set(), None
        self.lost = False
        self.parts_grouped, self.parts_total = None, None

    def fold_needs(self, row):
        rows = [row - 1, row] + ([row + 1] if self.window == "tsc" else [])
        if self.periodic:
            return sorted({r % self.nrows_total for r in rows})
        return [r for r in rows if 0 <= r < self.nrows_total]

    def _key_rows(self):
        """Tile row (of the buffer) of every particle's base cell, -1 outside the buffer."""
        sgrid = self.pos.numpy()[:, 0] * (self.n / self.L)
        base = np.floor(sgrid if self.window == "cic" else sgrid + 0.5)
        plane = np.mod(base - self.x_start, self.n).astype(np.int64)
        return np.where(plane < self.nx, plane // self.row_planes, -1)

    def reset(self):
        self.group()
        self.parts_grouped = set()
        self.parts_total = None

    def group_part(self, k, parts, closed_row0=0, closed_nrows=0, span=1):
        """Parts k .. k + span - 1 of `parts` (x-ordered input): no particle of them may belong to a row that has been
        walked - the closed range handed over must be exactly the rows walked so far."""
        assert span >= 1 and 0 <= k and k + span <= parts
        assert self.parts_total in (None, parts) and not (set(range(k, k + span)) & self.parts_grouped)
        self.parts_total = parts
        closed = {(closed_row0 + i) % self.nrows_total for i in range(closed_nrows)}
        assert closed == self.walked, (closed, self.walked)
        npart = self.pos.shape[0]
        rows = self._key_rows()[k * npart // parts:(k + span) * npart // parts]
        if np.isin(rows, list(closed)).any():
            self.lost = True                       # what the kernel counts as dropped
        self.parts_grouped.update(range(k, k + span))

    def group(self):
        self.parts_grouped, self.parts_total = None, None
        full = omesh.paint(self.pos.numpy(), None if self.mass is None else self.mass.numpy(), self.n, self.L, self.window)
        planes = (self.x_start + np.arange(self.nx)) % self.n
        self.lost = not np.isclose(full.sum(), full[planes].sum(), rtol=1e-12)
        self.full = torch.from_numpy(full[planes])
        if self.offset:
            first, count = self.owned if self.owned is not None else (0, self.nx)
            self.full[first:first + count] -= self.offset
        self.out.fill_(float("nan"))
        self.walked, self.folded = set(), set()

    def walk(self, row0, nrows):
        assert self.full is not None, "walk before group"
        for r in range(row0, row0 + nrows):
            assert r not in self.walked, "row walked twice"
            if self.parts_grouped is not None:     # grouping in parts: every particle of this row must have been grouped
                npart, rows = self.pos.shape[0], self._key_rows()
                owners = {int(i * self.parts_total // npart) for i in np.nonzero(rows == r)[0]} if self.parts_total else set()
                # (particle i belongs to part k iff k npart // K <= i < (k + 1) npart // K)
                owners = {k for k in range(self.parts_total or 0)
                          if (rows[k * npart // self.parts_total:(k + 1) * npart // self.parts_total] == r).any()}
                assert owners <= self.parts_grouped, f"row {r} walked before parts {owners - self.parts_grouped} were grouped"
            self.walked.add(r)

    def fold(self, row0, nrows):
        for r in range(row0, row0 + nrows):
            assert set(self.fold_needs(r)) <= self.walked, f"row {r} folded before its neighbours were walked"
            assert r not in self.folded, "row folded twice"
            self.folded.add(r)
            a, b = r * self.row_planes, min((r + 1) * self.row_planes, self.nx)
            self.out[a:b] = self.full[a:b]

    def check(self):
        if self.lost:
            raise RuntimeError("deposits fell outside the slab buffer")
