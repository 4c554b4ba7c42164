"""HBM-resident building blocks of the hot path, one Python call per C-ABI entry.

PyTorch-ROCm tensors are used only as device-memory holders (allocation,
streams, lifetime); every computation is a call into ``libastrild_hip.so``.
Arrays follow the reference's layouts: grids are ``(N, N, N)`` C order with
axis 0 slowest (``value_map[x, y, z]``, power_spectrum_3d.py:142-148), particle
positions ``(Np, 3)`` like pmesh's ``paint`` (stats_subfind.py:125-131).
"""
import ctypes as ct

import numpy as np
import torch

from . import _lib
from ._lib import F32, F64, check

_REAL = {torch.float32: F32, torch.float64: F64}
_CPLX = {torch.complex64: F32, torch.complex128: F64}
_TO_CPLX = {torch.float32: torch.complex64, torch.float64: torch.complex128}
_TO_REAL = {torch.complex64: torch.float32, torch.complex128: torch.float64}


def device():
    if not torch.cuda.is_available():
        raise _lib.AstrildHipError("astrild_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                                   "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream():
    return ct.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else ct.c_void_p(t.data_ptr())


def as_device(a, dtype=None):
    """numpy / torch -> contiguous CUDA tensor (the H2D hop of the Python API)."""
    if isinstance(a, torch.Tensor):
        t = a.to(device=device(), dtype=dtype or a.dtype)
    else:
        arr = np.ascontiguousarray(a)
        t = torch.from_numpy(arr).to(device=device(), dtype=dtype)
    return t.contiguous()


def upload_planes(arrays, dtype=torch.float64, flat=True):
    """Equal-sized host arrays -> views into ONE device allocation, plane after plane.  A kernel that reads the same pixel
    of all planes at once (`ast_kappa_stack`: 64 streams) runs 8-10 % faster over one large allocation than over 64
    separate 134 MB blocks of the caching allocator (scripts/micro/kappa_stack_skew.py: 6.3 against 5.7 TB/s - larger
    page-table fragments, fewer translation misses for the 64 concurrent streams).  Arrays of different sizes: one
    allocation each, as before."""
    arrays = [a if isinstance(a, torch.Tensor) else np.ascontiguousarray(a) for a in arrays]
    sizes = {int(np.prod(a.shape)) for a in arrays}
    if len(arrays) < 2 or len(sizes) != 1:
        out = [as_device(a, dtype) for a in arrays]
        return [t.reshape(-1) for t in out] if flat else out
    count = sizes.pop()
    slab = torch.empty((len(arrays), count), dtype=dtype, device=device())
    for p, a in enumerate(arrays):
        src = a if isinstance(a, torch.Tensor) else torch.from_numpy(a)
        slab[p].copy_(src.reshape(-1))
    return [slab[p] if flat else slab[p].view(tuple(arrays[p].shape)) for p in range(len(arrays))]


def to_numpy(t):
    """CUDA tensor -> numpy array (the D2H hop of the Python API).  A large result lands in page-locked memory - torch's caching
    host allocator; the block goes back to its pool when the array is dropped - so that the copy runs at the link's rate: a
    pageable destination gets a third of it (4096^2 float64, 134 MB: 5.6 ms against 16.4)."""
    if t.is_cuda and t.numel() * t.element_size() >= (1 << 20):
        buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        buf.copy_(t)
        return buf.numpy()
    return t.cpu().numpy()


def real_code(t):
    try:
        return _REAL[t.dtype]
    except KeyError:
        raise TypeError(f"expected float32/float64 tensor, got {t.dtype}") from None


def profile_enable(on=True):
    check(_lib.lib().ast_profile_enable(int(bool(on))), "ast_profile_enable")


def profile_report():
    """{launch site: (calls, total_ms)} measured with HIP events on the launch stream."""
    buf = ct.create_string_buffer(1 << 16)
    check(_lib.lib().ast_profile_report(buf, len(buf)), "ast_profile_report")
    out = {}
    for line in buf.value.decode().splitlines():
        name, calls, ms = line.rsplit(",", 2)
        out[name] = (int(calls), float(ms))
    return out


# ------------------------------------------------------------------ FFT plans
class FFTPlan:
    """Owns one rocFFT plan created through the C-ABI."""

    def __init__(self, kind, dtype_code, lengths, batch=1, scale=1.0, inplace=False, strided=None):
        self.handle = ct.c_void_p()
        L = _lib.lib()
        if strided is None:
            arr = (ct.c_size_t * len(lengths))(*[int(v) for v in lengths])
            check(L.ast_fft_plan_create(ct.byref(self.handle), kind, dtype_code, len(lengths), arr,
                                        int(batch), float(scale), int(bool(inplace))), "ast_fft_plan_create")
        else:
            length, stride, dist = strided
            check(L.ast_fft_plan_create_strided_1d(ct.byref(self.handle), kind, dtype_code, int(length),
                                                   int(stride), int(batch), int(dist), float(scale)),
                  "ast_fft_plan_create_strided_1d")

    @property
    def work_bytes(self):
        return int(_lib.lib().ast_fft_plan_work_bytes(self.handle))

    def execute(self, src, dst=None):
        check(_lib.lib().ast_fft_exec(self.handle, ptr(src), ptr(dst), stream()), "ast_fft_exec")

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().ast_fft_plan_destroy(self.handle)
                self.handle = ct.c_void_p()
        except Exception:
            pass


_plan_cache = {}


def fft_plan(kind, dtype_code, lengths, batch=1, scale=1.0, inplace=False, strided=None):
    key = (torch.cuda.current_device(), kind, dtype_code, tuple(lengths), batch, scale, inplace, strided)
    p = _plan_cache.get(key)
    if p is None:
        p = _plan_cache[key] = FFTPlan(kind, dtype_code, lengths, batch, scale, inplace, strided)
    return p


def clear_plan_cache():
    _plan_cache.clear()


# ------------------------------------------------------------ mass assignment
def ngp_assign(x, y, z, values, npar, dtype=torch.float64):
    """``value_map[(npar*x).astype(int), ...] = values`` with last-write-wins
    (PowerSpectrum3D._read_data, power_spectrum_3d.py:142-148)."""
    x, y, z, values = (as_device(a, dtype) for a in (x, y, z, values))
    n = int(npar)
    grid = torch.empty((n, n, n), dtype=dtype, device=device())
    owner = torch.empty(n * n * n, dtype=torch.int32, device=device())
    dropped = torch.zeros(1, dtype=torch.int64, device=device())
    check(_lib.lib().ast_ngp_assign(ptr(x), ptr(y), ptr(z), ptr(values), real_code(grid), x.numel(), n,
                                    ptr(grid), ptr(owner), ptr(dropped), stream()), "ast_ngp_assign")
    nd = int(dropped.item())
    if nd:
        raise IndexError(f"{nd} particles have coordinates outside [0, 1) (numpy raises for >= 1 and wraps negative "
                         f"indices, i.e. coordinates in [-1, 0), silently; both are rejected here)")
    return grid


def total_mass(mass, npart):
    """Sum of the particle masses in double, fixed order (``npart`` for unit masses)."""
    if mass is None:
        return float(npart)
    out = torch.empty(1 + _lib.SUM_PARTS, dtype=torch.float64, device=mass.device)
    check(_lib.lib().ast_sum(ptr(mass), real_code(mass), mass.numel(), ptr(out), stream()), "ast_sum")
    return float(out[0].item())


class PaintHalo:
    """Halo records of a ``paint(..., defer_fold=True)``: the grid is complete only once they are
    folded in, which ``power_sums_fused(grid, ..., halo=)`` does while its z pass loads the rows.
    Keeps the paint's workspace alive."""

    def __init__(self, workspace, rec_ptr, window_code):
        self.workspace, self.rec_ptr, self.window_code = workspace, rec_ptr, window_code


def sample_run_starts(npart, windows=256):
    """First particles of ``windows`` runs of 32 spread evenly over ``npart`` particles, each a multiple of 32 and at most
    npart - 32.  Host integers: a float32 ``linspace`` rounds 2^30 - 32 UP and the last run would start past the end."""
    npart, windows = int(npart), max(2, int(windows))
    assert npart >= 32
    return [min(i * (npart - 32) // (windows - 1) // 32 * 32, npart - 32) for i in range(windows)]


def sample_is_unordered(pos, nmesh, boxsize, shift=0.0, windows=256, fraction=False):
    """Looks at ``windows`` runs of 32 consecutive particles spread over ``pos``: in input with spatial order in memory
    (lattice order, cell- or curve-sorted snapshots, halo by halo) at least 8 particles of a run share the 8 x 8 x 32-cell
    tile of the run's middle particle - that is what the grouping kernel of the tiled paint turns into group records; in
    shuffled input next to none do and the paint belongs on the two-level bucket scatter from the start.  True when fewer
    than a quarter of the runs are groupable (``fraction=True``: that fraction itself).  On the device: ``ast_paint_order_probe`` and ONE 4-byte fetch; CPU tensors
    (tests) go through the same arithmetic in torch."""
    npart = int(pos.shape[0])
    if npart < 64:
        return False
    n = int(nmesh)
    if pos.is_cuda:                                    # one small kernel + a 4-byte fetch (the torch ops below: 0.25 ms)
        cnt = torch.empty(1, dtype=torch.int32, device=pos.device)
        check(_lib.lib().ast_paint_order_probe(ptr(pos), real_code(pos), npart, n, float(boxsize), float(shift), int(windows),
                                               ptr(cnt), stream()), "ast_paint_order_probe")
        frac = int(cnt.item()) / max(2, int(windows))
        return frac if fraction else frac < 0.25
    starts = torch.tensor(sample_run_starts(npart, windows), dtype=torch.int64, device=pos.device)
    idx = (starts[:, None] + torch.arange(32, device=pos.device)[None, :]).reshape(-1)
    cell = torch.floor(pos[idx].double() * (n / float(boxsize)) + float(shift)).long() % n
    tile = ((cell[:, 0] // 8) * n + cell[:, 1] // 8) * n + cell[:, 2] // 32
    tile = tile.view(-1, 32)
    groupable = float(((tile == tile[:, 15:16]).sum(dim=1) >= 8).double().mean().item())
    return groupable if fraction else groupable < 0.25


def probe_input(pos, nmesh, boxsize, shift=0.0, windows=256):
    """Looks at the particle array BEFORE a tiled paint (two small kernels, one 24-byte fetch): returns a dict with
    ``groupable`` - the fraction of sampled runs of 32 consecutive particles the grouping kernel could turn into group records
    (:func:`sample_is_unordered`; < 0.25: no spatial order in memory) -, ``overflow`` - the estimated number of particles beyond
    the single pass's fixed tile segments (ast_paint_occupancy_probe: a sample of >= 8 particles per 8 x 8 x 32-cell tile) -
    and ``max_tile`` - the largest estimated tile occupancy.  Evolved snapshots and halo catalogues
    (stats_subfind.py:125-131) are clustered: their dense tiles overflow the segments, and the paint belongs on the exact
    two-pass variant from the start."""
    L = _lib.lib()
    n, npart = int(nmesh), int(pos.shape[0])
    cbytes = int(L.ast_paint_occupancy_probe_bytes(n))
    if cbytes == 0 or npart < 64:
        return None
    ntiles = cbytes // 4
    samples = min(npart, max(65536, 8 * ntiles))
    counts = torch.empty(cbytes, dtype=torch.uint8, device=pos.device)
    out = torch.zeros(3, dtype=torch.int64, device=pos.device)       # [overflow, max tile, groupable runs (low 4 bytes)]
    check(L.ast_paint_occupancy_probe(ptr(pos), real_code(pos), npart, n, float(boxsize), float(shift), samples, ptr(counts),
                                      cbytes, ptr(out), stream()), "ast_paint_occupancy_probe")
    check(L.ast_paint_order_probe(ptr(pos), real_code(pos), npart, n, float(boxsize), float(shift), int(windows),
                                  ct.c_void_p(out.data_ptr() + 16), stream()), "ast_paint_order_probe")
    o = out.cpu().tolist()
    return {"overflow": int(o[0]), "max_tile": int(o[1]), "groupable": (int(o[2]) & 0xffffffff) / max(2, int(windows)),
            "mean_tile": npart / ntiles, "samples": samples}


def synth_clustered_particles(npside, nmesh, boxsize, seed=20240601, sigma_cells=0.5, nattractors=256, amplitude=0.95,
                              shuffle=False, dtype=torch.float32):
    """A clustered synthetic set (ast_synth_clustered_particles): the lattice collapsing onto ``nattractors`` centres - tile
    occupancies ~100 x the mean, like an evolved snapshot - in lattice order or (shuffle) in pseudo-random order."""
    count = int(npside) ** 3
    pos = torch.empty((count, 3), dtype=dtype, device=device())
    check(_lib.lib().ast_synth_clustered_particles(ptr(pos), real_code(pos), count, int(npside), float(boxsize),
                                                   float(sigma_cells) * boxsize / nmesh, int(seed), int(nattractors),
                                                   float(amplitude), int(bool(shuffle)), stream()), "ast_synth_clustered_particles")
    return pos


def auto_paint_method(npart, n, nx, window, hint=None, accumulate=False):
    """What ``paint(method="auto")`` runs (measured on the MI355X, scripts/perf_sparse.py: 256^3 and 512^3 grids, halo-like
    catalogues of 1 ... 64 objects per 8 x 8 x 32-cell tile).  The tiled paint walks every column of the grid, so below ~16
    (CIC) / ~8 (TSC) objects per tile - SubFind haloes on nbins = 1024 - global atomics on a zero-filled grid win:
    "direct".  Above that, catalogues too small for the input probe (< 2^20 objects), or still sparse (< 64 per tile:
    halo catalogues, stats_subfind.py:125-131, clumpy by nature), take the exact two-pass lists - as fast as the single
    pass there and without its per-tile capacity (512^3, 32 per tile, TSC float64: direct 2.7 ms, tiled2 1.45; a clumpy
    256^3 set at 64 per tile: single pass 1.0 ms through its overflow list, tiled2 0.2).  Dense input goes to "tiled",
    where the probe picks single pass / scatter levels / two-pass from the input itself."""
    per_tile = npart * 2048 / max(1, nx * n * n)
    if accumulate:                                   # (adding onto a grid: index lists + atomic tile flush; the round-1 threshold)
        return "tiled" if n % 32 == 0 and npart >= 65536 and per_tile >= 64 else "direct"
    if n % 32 or npart < 65536 or per_tile < (8 if window.lower() == "tsc" else 16):
        return "direct"
    if hint is None and (npart < (1 << 20) or per_tile < 64):
        return "tiled2"
    return "tiled"


def paint(pos, mass, nmesh, boxsize, window="cic", scale=1.0, out=None, method="auto",
          x_start=0, nx_alloc=None, check_dropped=True, accumulate=None, defer_fold=False, offset=0.0,
          hint=None, stats=None, shift=0.0, offset_planes=None):
    """pmesh ``ParticleMesh.paint(pos, mass=, resampler=)`` on the GPU.

    pos: (Np, 3) CUDA tensor (float32/float64); mass: (Np,) or None.
    Returns the grid ``(nx_alloc, nmesh, nmesh)`` in pos.dtype.
    method: "direct" (global float atomics), "tiled" (LDS tiles, single pass over the
    particles), "tiled2" (LDS tiles, exact two-pass counting) or "auto" (= tiled when possible).
    accumulate: add into ``out`` (default when ``out`` is given) or overwrite it (default
    for a fresh grid; the tiled path then needs no zero-fill and flushes without atomics).
    defer_fold: (tiled overwrite of the whole grid only) skip the paint's last kernel and return
    ``(grid, PaintHalo)`` for ``power_sums_fused(..., halo=)``; the grid alone is incomplete.
    offset: (tiled overwrite only) owned cells are stored as ``sum - offset``, subtracted in double
    before the one rounding to the grid dtype; ``offset="mean"`` uses total mass * scale / nmesh^3,
    i.e. the grid holds rho - mean (only the DC mode changes, which FFTPower discards).
    hint: "scattered" sizes the tiled overwrite paint's workspace for particles without spatial order in memory
    (AST_PAINT_SCATTERED); "clustered" goes to the exact two-pass variant (no capacity limit per tile); "ordered" the
    plain single pass.  Without a hint a paint of >= 2^20 particles onto the whole grid looks at the input first
    (:func:`probe_input`: order in memory, tile occupancy tail) and picks single pass / scattered (no order in memory,
    clustered or not) / two-pass (ordered and clustered) up front - also when ``check_dropped`` is False - instead of
    finding out from the overflow list of a wasted attempt;
    "xsorted" says they come in ascending x (lattice order, slab-ordered files): grouping and
    column walk then overlap chunk by chunk (AST_PAINT_XSORTED; a wrong hint costs time, never correctness).  stats: a dict that receives the list statistics of the tiled overwrite paint.
    shift: added to every coordinate in grid units (0.5 paints the second mesh of an interlaced pair).
    offset_planes: (first, count) of the buffer planes the offset applies to (default: all) - a slab buffer's
    ghost planes are added onto other ranks' cells and must stay plain sums.
    """
    L = _lib.lib()
    n = int(nmesh)
    nx = n if nx_alloc is None else int(nx_alloc)
    assert pos.is_cuda and pos.dim() == 2 and pos.shape[1] == 3 and pos.is_contiguous()
    code = real_code(pos)
    if mass is not None:
        assert mass.is_cuda and mass.dtype == pos.dtype and mass.numel() == pos.shape[0] and mass.is_contiguous()
    if accumulate is None:
        accumulate = out is not None
    # The tile lists hold 32-bit particle indices: more than 2^32 - 65 particles (2048^3 on the largest grid one GPU holds)
    # are painted in chunks - the first one as asked for, the others accumulated onto it through the same LDS tiles.
    # ASTRILD_PAINT_CHUNK (particles) forces chunking at smaller sizes (tests).
    import os
    chunk = int(os.environ.get("ASTRILD_PAINT_CHUNK", 0)) or (2 ** 31 if pos.shape[0] >= 2 ** 32 - 65 else 0)
    if chunk and pos.shape[0] > chunk and method != "direct" and _lib.WIN[window.lower()] != 0:
        if defer_fold:
            raise _lib.AstrildHipError("defer_fold is not available for a paint in chunks (more than 2^32 - 65 particles)")
        if isinstance(offset, str):              # "mean" means the mean of ALL particles, not of the first chunk
            if offset != "mean":
                raise ValueError(offset)
            offset = total_mass(mass, pos.shape[0]) * float(scale) / float(n) ** 3
        for a in range(0, pos.shape[0], chunk):
            part = paint(pos[a:a + chunk], None if mass is None else mass[a:a + chunk], n, boxsize, window, scale=scale, out=out,
                         method=method, x_start=x_start, nx_alloc=nx_alloc, check_dropped=check_dropped,
                         accumulate=accumulate if a == 0 else True, offset=offset if a == 0 else 0.0, hint=hint if a == 0 else None,
                         shift=shift, offset_planes=offset_planes)
            out = part
        return out
    win = _lib.WIN[window.lower()]
    npart = pos.shape[0]
    dropped = torch.zeros(1, dtype=torch.int64, device=pos.device)
    ws_bytes = 0
    # TWO_PASS | OVERWRITE | DEFER_FOLD | SCATTERED | XSORTED
    if hint not in (None, "scattered", "xsorted", "clustered", "ordered"):
        raise ValueError(hint)
    if hint == "clustered" and method in ("auto", "tiled"):
        method = "tiled2"
    was_auto = method == "auto"
    if was_auto and win != 0:
        method = auto_paint_method(npart, n, nx, window, hint, accumulate)
    tflags = (1 if method == "tiled2" else 0) | (0 if accumulate else 2) | (4 if defer_fold else 0) | \
             (8 if hint == "scattered" and not accumulate and method != "tiled2" else 0) | \
             (16 if hint == "xsorted" and not accumulate and method != "tiled2" else 0)
    if method in ("auto", "tiled", "tiled2") and win != 0 and npart < 2**32 - 65:
        ws_bytes = int(L.ast_paint_tiled_workspace_bytes(win, code, npart, n, nx, tflags))
    if method in ("tiled", "tiled2") and ws_bytes == 0:
        if not was_auto:
            raise _lib.AstrildHipError("tiled paint needs a CIC/TSC window and nmesh a multiple of 32")
        method, tflags = "direct", 0                  # a buffer geometry the tiles do not cover
    use_tiled = ws_bytes > 0 and method in ("tiled", "tiled2")      # ("auto" was resolved above: auto_paint_method)
    if defer_fold and not (use_tiled and not accumulate and x_start == 0 and nx == n):
        raise _lib.AstrildHipError("defer_fold needs the tiled overwrite paint of the whole periodic grid")
    if out is None:
        alloc = torch.empty if (use_tiled and not accumulate) else torch.zeros
        out = alloc((nx, n, n), dtype=pos.dtype, device=pos.device)
    else:
        assert out.is_cuda and out.dtype == pos.dtype and out.numel() == nx * n * n and out.is_contiguous()
        if not accumulate and not use_tiled:
            out.zero_()
    if offset != 0.0 and not (use_tiled and not accumulate):
        raise _lib.AstrildHipError("offset needs the tiled overwrite paint")
    if isinstance(offset, str):
        if offset != "mean":
            raise ValueError(offset)
        offset = total_mass(mass, npart) * float(scale) / float(n) ** 3
    off_planes = (0, -1) if offset_planes is None else offset_planes      # (first buffer plane, count) that get the offset
    compact = use_tiled and not accumulate and method != "tiled2"       # single pass + overwrite: group / stray lists
    attempts, probed = 0, None
    if compact and hint is None and npart >= (1 << 20) and nx == n and int(x_start) == 0:
        probed = probe_input(pos, n, boxsize, shift)                    # (a wrong guess costs time, never correctness)
        if probed is not None and probed["groupable"] < 0.25 and probed["overflow"] <= npart // 5:
            # no spatial order in memory: the bucket scatter, clustered or not - the two-pass variant makes two global
            # atomics per particle on such input (1024^3 clustered + shuffled: 130 ms against 24); tiles that overflow
            # their segments go through the late list, reserved once per workgroup and chunk (the list holds a quarter of
            # the particles: beyond a fifth estimated, the two-pass variant below, slow but without any capacity)
            tflags |= 8
        elif probed is not None and probed["overflow"] > npart // 64:
            tflags = (tflags & ~(8 | 16)) | 1                           # clustered, in file order: the exact two-pass variant at once
            compact = False
        ws_bytes = int(L.ast_paint_tiled_workspace_bytes(win, code, npart, n, nx, tflags))
    elif compact and hint is None and check_dropped and npart >= (1 << 20) and sample_is_unordered(pos, n, boxsize, shift):
        tflags |= 8                                                     # (slab buffers: the order probe alone)
        ws_bytes = int(L.ast_paint_tiled_workspace_bytes(win, code, npart, n, nx, tflags))
    if use_tiled:
        mass_bound = 1.0
        if mass is not None and not accumulate and npart:
            lo_hi = torch.empty(2, dtype=torch.float64, device=pos.device)      # bound for the fixed-point tiles
            check(L.ast_minmax(ptr(mass), code, npart, ptr(lo_hi), stream()), "ast_minmax")
            mass_bound = float(lo_hi.abs().max()) or 1.0
        while True:
            attempts += 1
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=pos.device)
            check(L.ast_paint_tiled(win, code, ptr(pos), ptr(mass), npart, n, float(boxsize), float(scale),
                                    int(x_start), nx, ptr(out), ptr(ws), ws_bytes, ptr(dropped), tflags,
                                    mass_bound, float(offset), int(off_planes[0]), int(off_planes[1]), float(shift), stream()),
                  "ast_paint_tiled")
            st = None
            if compact and (stats is not None or check_dropped):
                st = torch.empty(4, dtype=torch.int64, device=pos.device)
                check(L.ast_paint_tiled_list_stats(ptr(ws), win, code, npart, n, nx, tflags, ptr(st), stream()),
                      "ast_paint_tiled_list_stats")
                st = dict(zip(("groups", "strays", "overflow", "max_strays_per_tile"), st.cpu().tolist()))
            # Too much went through the overflow list.  Particles without spatial order in memory overflow the
            # default stray segments: paint again with the two-level bucket scatter (AST_PAINT_SCATTERED).  If that
            # still overflows, the input is strongly CLUSTERED (tiles far above twice the mean occupancy): the exact
            # two-pass variant has no capacity limit.  (Callers that synchronise anyway.)
            if st is not None and check_dropped and st["overflow"] > npart // 64 and not (probed is not None and tflags & 8):
                # (a scattered paint the probe chose knowingly - unordered AND clustered - keeps its late list: it is the
                # fastest path for such input, and the result is complete either way)
                dropped.zero_()
                del ws
                if not tflags & 8:
                    tflags = (tflags | 8) & ~16
                elif not tflags & 1:
                    tflags = (tflags & ~8) | 1
                    compact = False
                ws_bytes = int(L.ast_paint_tiled_workspace_bytes(win, code, npart, n, nx, tflags))
                continue
            if stats is not None:
                stats.update(st or {}, scattered=bool(tflags & 8), attempts=attempts,
                             path="two-pass" if tflags & 1 else "scattered" if tflags & 8 else "single-pass")
                if probed is not None:
                    stats["probe"] = probed
            break
    else:
        check(L.ast_paint(win, code, ptr(pos), ptr(mass), npart, n, float(boxsize), float(scale),
                          int(x_start), nx, ptr(out), ptr(dropped), float(shift), stream()), "ast_paint")
    if check_dropped:
        nd = int(dropped.item())
        if nd:
            raise _lib.AstrildHipError(f"{nd} deposits fell outside the grid buffer "
                                       f"(x_start={x_start}, nx_alloc={nx})")
    if defer_fold:
        rec = ct.c_void_p()
        check(L.ast_paint_tiled_halo(ptr(ws), win, code, npart, n, nx, tflags, ct.byref(rec)), "ast_paint_tiled_halo")
        return out, PaintHalo(ws, rec, win)
    return out


class StagedPaint:
    """The single-pass overwrite paint of ``paint(..., method="tiled", accumulate=False)`` in parts
    (``ast_paint_tiled_stage``): ``group()`` once, then ``walk(row0, nrows)`` and ``fold(row0, nrows)`` per range of tile
    rows (``row_planes`` consecutive buffer planes each, ``nrows_total`` rows), in any order that walks a row's
    neighbours (``fold_needs``) before the row is folded.  After every row has been walked and folded ``out`` equals
    the one-call paint bit for bit.  Calls go to the current stream; the workspace lives as long as the object.
    The reference paints on one rank in one call (stats_subfind.py:130-131); this is what lets the slab pipeline send
    finished planes while the rest of the slab is still being painted."""

    def __init__(self, pos, mass, nmesh, boxsize, window, out, x_start=0, nx_alloc=None, scale=1.0, offset=0.0,
                 offset_planes=None, hint=None, shift=0.0):
        L = _lib.lib()
        self.n = int(nmesh)
        self.nx = self.n if nx_alloc is None else int(nx_alloc)
        assert pos.is_cuda and pos.dim() == 2 and pos.shape[1] == 3 and pos.is_contiguous()
        assert out.is_cuda and out.dtype == pos.dtype and out.numel() == self.nx * self.n * self.n and out.is_contiguous()
        if hint not in (None, "scattered"):
            raise ValueError(hint)
        self.pos, self.mass, self.out = pos, mass, out
        self.code = real_code(pos)
        self.win = _lib.WIN[window.lower()]
        self.window = window.lower()
        self.flags = 2 | (8 if hint == "scattered" else 0)                  # OVERWRITE [| SCATTERED]
        self.npart = pos.shape[0]
        self.ws_bytes = int(L.ast_paint_tiled_workspace_bytes(self.win, self.code, self.npart, self.n, self.nx, self.flags))
        if self.ws_bytes == 0 or self.win == 0 or self.npart >= 2**32 - 65:
            raise _lib.AstrildHipError("staged paint needs a CIC/TSC window and nmesh a multiple of 32")
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=pos.device)
        self.dropped = torch.zeros(1, dtype=torch.int64, device=pos.device)
        self.mass_bound = 1.0
        if mass is not None and self.npart:
            lo_hi = torch.empty(2, dtype=torch.float64, device=pos.device)
            check(L.ast_minmax(ptr(mass), self.code, self.npart, ptr(lo_hi), stream()), "ast_minmax")
            self.mass_bound = float(lo_hi.abs().max()) or 1.0
        off = (0, -1) if offset_planes is None else offset_planes
        self.args = (float(boxsize), float(scale), int(x_start), self.nx)
        self.tail = (self.mass_bound, float(offset), int(off[0]), int(off[1]), float(shift))
        self.row_planes = int(L.ast_paint_tile_row_planes())
        self.nrows_total = int(L.ast_paint_tile_rows(self.nx))
        self.periodic = self.nx == self.n and int(x_start) == 0

    def _stage(self, stage, row0, nrows, closed_row0=0, closed_nrows=0):
        check(_lib.lib().ast_paint_tiled_stage(self.win, self.code, ptr(self.pos), ptr(self.mass), self.npart, self.n,
                                               *self.args, ptr(self.out), ptr(self.ws), self.ws_bytes, ptr(self.dropped),
                                               self.flags, *self.tail, int(stage), int(row0), int(nrows), int(closed_row0),
                                               int(closed_nrows), stream()),
              "ast_paint_tiled_stage")

    def fold_needs(self, row):
        """Tile rows whose walk must be complete before ``fold(row, 1)``: the row itself and the neighbours whose
        window reaches into it (CIC deposits reach one plane up, TSC one plane either way)."""
        rows = [row - 1, row] + ([row + 1] if self.window == "tsc" else [])
        if self.periodic:
            return sorted({r % self.nrows_total for r in rows})
        return [r for r in rows if 0 <= r < self.nrows_total]

    def group(self):
        self.dropped.zero_()
        self._stage(0, 0, 0)

    def reset(self):
        """Instead of group(), before the first group_part()."""
        self.dropped.zero_()
        self._stage(4, 0, 0)

    def group_part(self, k, parts, closed_row0=0, closed_nrows=0, span=1):
        """The lists of parts k .. k + span - 1 of `parts` equal parts of the particle array, in one launch (x-ordered input,
        slab buffers).  closed_*: the tile rows walked so far, one range modulo the buffer's rows; a particle that turns up
        for one of them is counted as dropped (check() raises): the order that was promised did not hold."""
        assert 1 <= parts <= 65535 and span >= 1 and 0 <= k and k + span <= parts
        self._stage(3, k, parts | ((span - 1) << 16), closed_row0, closed_nrows)

    def walk(self, row0, nrows):
        self._stage(1, row0, nrows)

    def fold(self, row0, nrows):
        """FOLD of the tile rows [row0, row0 + nrows).  After :meth:`defer_folds` only the rows named there get their records
        added here; the others get their overflow-list deposits only (AST_PAINT_STAGE_LATE) and their records when the
        consumer's z pass loads the planes (:meth:`halo_args`)."""
        if self._explicit_rows is None:
            self._stage(2, row0, nrows)
            return
        r = row0
        while r < row0 + nrows:
            kind = r in self._explicit_rows
            e = r
            while e < row0 + nrows and (e in self._explicit_rows) == kind:
                e += 1
            self._stage(2 if kind else 5, r, e - r)
            r = e

    _explicit_rows = None

    def defer_folds(self, explicit_rows):
        """From now on only ``explicit_rows`` (the tile rows that hold ghost planes: their planes travel before any transform)
        are folded by fold(); every other row's halo records are added by the z pass that reads its planes
        (ast_fft_tile_rows_r2c_slab_halo) - one kernel and one read-modify-write of the border lines less per row.  The rows
        folded on load must form one range [lo, hi) of the buffer's rows."""
        explicit = set(int(r) for r in explicit_rows)
        rest = [r for r in range(self.nrows_total) if r not in explicit]
        if rest and rest != list(range(rest[0], rest[-1] + 1)):
            raise ValueError("the rows folded on load must be one contiguous range")
        rec = ct.c_void_p()
        check(_lib.lib().ast_paint_tiled_halo(ptr(self.ws), self.win, self.code, self.npart, self.n, self.nx, self.flags, ct.byref(rec)),
              "ast_paint_tiled_halo")
        self._explicit_rows = explicit
        self._halo = (rec, self.win, rest[0] if rest else 0, rest[-1] + 1 if rest else 0)

    def halo_args(self):
        """(record pointer, window code, first tile row folded on load, one past the last) or None."""
        return None if self._explicit_rows is None else self._halo

    def check(self):
        nd = int(self.dropped.item())
        if nd:
            raise _lib.AstrildHipError(f"{nd} deposits fell outside the grid buffer (x_start={self.args[2]}, nx_alloc={self.nx}) "
                                       f"or, with group_part(), belong to a tile row that had been walked already (the particles "
                                       f"do not come in ascending x as promised)")


def synth_lattice_particles(npside, nmesh, boxsize, seed=20240601, sigma_cells=0.5, shuffle=False,
                            dtype=torch.float32, first=0, count=None):
    """Synthetic particle set of SURVEY.md §8(d), generated directly in HBM.  shuffle=True: the particles of the range in
    a pseudo-random order (a fixed permutation keyed by seed + 1); shuffle="stride": t -> t * stride mod count, the
    low-discrepancy order of rounds 1-4 (every chunk of the array feeds every tile almost evenly: a friendlier input)."""
    n3 = int(npside) ** 3
    count = n3 - first if count is None else int(count)
    pos = torch.empty((count, 3), dtype=dtype, device=device())
    stride = 0
    if shuffle == "stride" and count > 1:
        stride = 2654435761 % count or 1
        while np.gcd(stride, count) != 1:
            stride += 1
    elif shuffle and count > 1:
        stride = 2 ** 64 - 1
    check(_lib.lib().ast_synth_lattice_particles(ptr(pos), real_code(pos), int(first), count, int(npside),
                                                 float(boxsize), float(sigma_cells) * boxsize / nmesh,
                                                 int(seed), int(stride), stream()), "ast_synth_lattice_particles")
    return pos


# ---------------------------------------------------------------- 3D spectra
def r2c(field, out=None, engine="auto"):
    """pmesh-normalised forward transform: ``rfftn(field) / Ng``.

    engine "tile": the hand-written three-pass LDS FFT (fp32 cubes of side 256/512/1024, fp64 cubes of side 128 ... 2048);
    "rocfft": the rocFFT 3D R2C plan; "auto": tile when supported."""
    assert field.is_cuda and field.dim() == 3 and field.is_contiguous()
    n0, n1, n2 = field.shape
    code = real_code(field)
    if out is None:
        out = torch.empty((n0, n1, n2 // 2 + 1), dtype=_TO_CPLX[field.dtype], device=field.device)
    L = _lib.lib()
    tile_ok = n0 == n1 == n2 and bool(L.ast_fft_tile_supported(code, n0))
    tile64 = n0 == n1 == n2 and field.dtype == torch.float64 and bool(L.ast_fft64_supported(n0)) and out.is_contiguous()
    if tile64 and engine in ("auto", "tile"):
        check(L.ast_fft64_r2c_3d(ptr(field), ptr(out), n0, 1.0 / float(n0) ** 3, stream()), "ast_fft64_r2c_3d")
        return out
    if engine == "tile" and not tile_ok:
        raise _lib.AstrildHipError("tile FFT supports cubes of side 256, 512 or 1024")
    if tile_ok and engine in ("auto", "tile"):
        check(L.ast_fft_tile_r2c_3d(ptr(field), ptr(out), code, n0, 1.0 / float(n0) ** 3, stream()),
              "ast_fft_tile_r2c_3d")
        return out
    plan = fft_plan(_lib.FFT_R2C, code, (n0, n1, n2), 1, 1.0 / (n0 * n1 * n2), False)
    plan.execute(field, out)
    return out


_geom_cache = {}

#: How lattice vectors of exactly integer norm (they sit on a shell edge) are assigned: "float64" follows the
#: rounding of nbodykit's float64 comparison (the reference's behaviour, as far as its published source fixes it),
#: "integer" assigns them to the shell they open.  See ast_power_bin_1d in include/astrild_hip.h.
DEFAULT_BINNING = "float64"


def _bin_code(binning):
    return _lib.BIN[binning or DEFAULT_BINNING]


def shell_geometry(nmesh, boxsize, i0=None, i1=None, binning=None):
    """(sum w|k|, sum w) per shell of a spectrum block — data independent, cached."""
    n = int(nmesh)
    i0 = (0, n) if i0 is None else tuple(i0)
    i1 = (0, n) if i1 is None else tuple(i1)
    key = (torch.cuda.current_device(), n, float(boxsize), i0, i1, _bin_code(binning))
    hit = _geom_cache.get(key)
    if hit is None:
        nb = n // 2 - 1
        ksum = torch.zeros(nb, dtype=torch.float64, device=device())
        nmodes = torch.zeros(nb, dtype=torch.int64, device=device())
        check(_lib.lib().ast_power_bin_1d(None, None, F64, n, float(boxsize), int(i0[0]), int(i0[1]),
                                          int(i1[0]), int(i1[1]), ptr(ksum), None, ptr(nmodes), _bin_code(binning), stream()),
              "ast_power_bin_1d[geometry]")
        hit = _geom_cache[key] = (ksum, nmodes)
    return hit


def power_bin_1d(spec1, spec2, nmesh, boxsize, i0=None, i1=None, psum=None, binning=None):
    """Shell sums (ksum, psum, nmodes) of a block of the half spectrum (device tensors)."""
    n = int(nmesh)
    nb = n // 2 - 1
    i0 = (0, n) if i0 is None else tuple(i0)
    i1 = (0, n) if i1 is None else tuple(i1)
    assert spec1.is_cuda and spec1.is_contiguous() and spec1.numel() == i0[1] * i1[1] * (n // 2 + 1)
    code = _CPLX[spec1.dtype]
    if spec2 is not None:
        assert spec2.dtype == spec1.dtype and spec2.numel() == spec1.numel() and spec2.is_contiguous()
    if psum is None:
        psum = torch.zeros(nb, dtype=torch.float64, device=spec1.device)
    ksum, nmodes = shell_geometry(n, boxsize, i0, i1, binning)
    check(_lib.lib().ast_power_bin_1d(ptr(spec1), ptr(spec2), code, n, float(boxsize), int(i0[0]), int(i0[1]),
                                      int(i1[0]), int(i1[1]), None, ptr(psum), None, _bin_code(binning), stream()),
          "ast_power_bin_1d")
    return ksum, psum, nmodes


def finish_power(ksum, psum, nmodes):
    """k = ksum/N, P = psum/N on the host; empty shells are NaN like nbodykit's 0/0."""
    ks, ps, nm = (t.cpu().numpy() for t in (ksum, psum, nmodes))
    with np.errstate(invalid="ignore", divide="ignore"):
        return {"k": ks / nm, "power": ps / nm, "modes": nm, "shotnoise": 0.0}


_power_scratch = {}


def power_sums_fused(field, boxsize, psum=None, mean=0.0, halo=None, lowk=True, binning=None):
    """(ksum, psum, nmodes) of the auto power of an fp32 cube of side 256/512/1024 through
    the fused tile-FFT + shell-binning path (the spectrum is never written to HBM).
    ``mean`` is subtracted from the cells on load: it only touches the discarded DC mode
    and keeps fp32 round-off from scaling with the mean density (pass total_mass/Ng).
    ``lowk``: the five lowest shells come from double-precision DFT sums of the modes |m_i| <= 5 (one more read
    of the grid): the fp32 transform's round-off floor would otherwise cap them at ~2e-6 / |m|^2."""
    n = field.shape[0]
    L = _lib.lib()
    key = (torch.cuda.current_device(), n)
    scratch = _power_scratch.get(key)
    if scratch is None:
        _power_scratch.clear()
        scratch = _power_scratch[key] = torch.empty(int(L.ast_fft_tile_power_scratch_bytes(n)), dtype=torch.uint8,
                                                    device=field.device)
    if psum is None:
        psum = torch.zeros(n // 2 - 1, dtype=torch.float64, device=field.device)
    ksum, nmodes = shell_geometry(n, boxsize, binning=binning)
    if halo is not None:                      # grid from paint(..., defer_fold=True)
        check(L.ast_fft_tile_power_3d_halo(ptr(field), halo.rec_ptr, halo.window_code, ptr(scratch), scratch.numel(),
                                           real_code(field), n, float(boxsize), float(mean), int(bool(lowk)), _bin_code(binning), ptr(psum), stream()),
              "ast_fft_tile_power_3d_halo")
        return ksum, psum, nmodes
    check(L.ast_fft_tile_power_3d(ptr(field), ptr(scratch), scratch.numel(), real_code(field), n, float(boxsize),
                                  float(mean), int(bool(lowk)), _bin_code(binning), ptr(psum), stream()), "ast_fft_tile_power_3d")
    return ksum, psum, nmodes


def lowk_modes(planes, nmesh, x0=0, out=None):
    """Contribution of the planes x0 .. x0 + nx - 1 (an (nx, n, n) fp32 tensor) to the low-k modes |m_i| <= 6 in
    double (complex128 tensor of ast_lowk_mode_count() entries; accumulated into ``out`` when given)."""
    L = _lib.lib()
    n, nx = int(nmesh), planes.shape[0]
    assert planes.is_cuda and planes.is_contiguous() and planes.dtype == torch.float32 and tuple(planes.shape[1:]) == (n, n)
    work_bytes = int(L.ast_lowk_work_bytes(n, nx))
    work = torch.empty(work_bytes, dtype=torch.uint8, device=planes.device)
    acc = out is not None
    if out is None:
        out = torch.empty(int(L.ast_lowk_mode_count()), dtype=torch.complex128, device=planes.device)
    check(L.ast_lowk_modes(ptr(planes), F32, n, int(x0), nx, int(acc), ptr(out), ptr(work), work_bytes, stream()),
          "ast_lowk_modes")
    return out


def lowk_shell_sums(modes, nmesh, boxsize, binning=None):
    """psum entries of the lowest shells from the (complete) low-k modes."""
    L = _lib.lib()
    sums = torch.empty(int(L.ast_lowk_shell_count()), dtype=torch.float64, device=modes.device)
    check(L.ast_lowk_shell_sums(ptr(modes), int(nmesh), float(boxsize), _bin_code(binning), ptr(sums), stream()),
          "ast_lowk_shell_sums")
    return sums


def fused_power64_supported(field, allow_f32=False):
    """float64 cubes of side 128 ... 2048; with allow_f32 also float32 cubes (transformed in double, widened on load)."""
    n = field.shape[0]
    ok = (torch.float64, torch.float32) if allow_f32 else (torch.float64,)
    return field.dim() == 3 and tuple(field.shape) == (n, n, n) and field.dtype in ok and field.is_contiguous() \
        and bool(_lib.lib().ast_fft64_supported(n))


def power_sums_fused64(field, boxsize, psum=None, binning=None, halo=None, mean=None):
    """(ksum, psum, nmodes) of the auto power of a float64 cube of side 128 ... 2048 through the hand-written
    double-precision passes (ast_fft64_power_3d): one pass per axis, the last one fused with the shell binning.  fp32 cubes
    of the sides the fp32 tile passes do not cover take the same passes, widened on load - or, at side 2048 and with the
    grid's ``mean`` given (0.0 for a grid that holds rho - mean), single-precision passes of their own
    (ast_fft32_big_power_3d)."""
    import os
    n = field.shape[0]
    L = _lib.lib()
    if psum is None:
        psum = torch.zeros(n // 2 - 1, dtype=torch.float64, device=field.device)
    ksum, nmodes = shell_geometry(n, boxsize, binning=binning)
    if field.dtype == torch.float32 and mean is not None and bool(L.ast_fft32_big_supported(n)) \
            and not os.environ.get("ASTRILD_FFT32_BIG_OFF"):
        # side 2048: single-precision passes (half the bytes of the double route below); the grid's mean leaves as the rows
        # are loaded and the sixteen lowest shells come from double-precision sums over the grid (inside the call) - fp32
        # round-off of an O(1) field on shells of little power - as the fp32 tile pipeline takes its five lowest
        key32 = (torch.cuda.current_device(), n, "f32big")
        scratch = _power_scratch.get(key32)
        if scratch is None:
            _power_scratch.clear()
            scratch = _power_scratch[key32] = torch.empty(int(L.ast_fft32_big_power_scratch_bytes(n)), dtype=torch.uint8, device=field.device)
        if halo is not None:                  # grid from paint(..., defer_fold=True): the fold rides on the z rows
            check(L.ast_fft32_big_power_3d_halo(ptr(field), halo.rec_ptr, halo.window_code, ptr(scratch), scratch.numel(), n,
                                                float(boxsize), _bin_code(binning), float(mean), ptr(psum), stream()),
                  "ast_fft32_big_power_3d_halo")
        else:
            check(L.ast_fft32_big_power_3d(ptr(field), ptr(scratch), scratch.numel(), n, float(boxsize), _bin_code(binning), float(mean),
                                           ptr(psum), stream()), "ast_fft32_big_power_3d")
        return ksum, psum, nmodes
    key = (torch.cuda.current_device(), n, "f64")
    scratch = _power_scratch.get(key)
    if scratch is None:
        _power_scratch.clear()
        scratch = _power_scratch[key] = torch.empty(int(L.ast_fft64_power_scratch_bytes(n)), dtype=torch.uint8, device=field.device)
    if field.dtype == torch.float32:          # an fp32 grid through the double passes (sizes without fp32 tile passes)
        assert halo is None
        check(L.ast_fft64_power_3d_f32(ptr(field), ptr(scratch), scratch.numel(), n, float(boxsize), _bin_code(binning), ptr(psum),
                                       stream()), "ast_fft64_power_3d_f32")
        return ksum, psum, nmodes
    if halo is not None:                      # grid from paint(..., defer_fold=True)
        check(L.ast_fft64_power_3d_halo(ptr(field), halo.rec_ptr, halo.window_code, ptr(scratch), scratch.numel(), n,
                                        float(boxsize), _bin_code(binning), ptr(psum), stream()), "ast_fft64_power_3d_halo")
        return ksum, psum, nmodes
    check(L.ast_fft64_power_3d(ptr(field), ptr(scratch), scratch.numel(), n, float(boxsize), _bin_code(binning), ptr(psum), stream()),
          "ast_fft64_power_3d")
    return ksum, psum, nmodes


def fused_power_supported(field):
    n = field.shape[0]
    return field.dim() == 3 and tuple(field.shape) == (n, n, n) and field.dtype == torch.float32 \
        and bool(_lib.lib().ast_fft_tile_supported(F32, n))


def paint_power_1d(pos, mass, nmesh, boxsize, window="cic", scale=1.0, binning=None, defer_fold64=True, pos_scale=1.0):
    """``pm.paint(...)`` followed by ``FFTPower(ArrayMesh(grid), mode="1d")`` (stats_subfind.py:130-150)
    as one pipeline: where the fused fp32 path applies, the paint's halo fold rides on the FFT's z pass.
    pos_scale: the positions are in units of 1 / pos_scale box units (``pos * pos_scale`` is what the reference paints,
    stats_subfind.py:121-122: kpc -> Mpc/h): folded into the cell lookup - the catalogue is painted as it was read."""
    n = int(nmesh)
    paint_box = float(boxsize) / float(pos_scale)
    # which tiled variant (None: a catalogue too sparse for the tiles - global atomics, then the plain transform)
    tiled = auto_paint_method(pos.shape[0], n, n, window) if n % 32 == 0 else "direct"
    tiled = None if tiled == "direct" else tiled
    fast = pos.dtype == torch.float32 and tiled is not None and bool(_lib.lib().ast_fft_tile_supported(F32, n))
    if fast:
        # the grid holds rho - mean (subtracted before the fp32 rounding): only the discarded DC mode differs
        grid, halo = paint(pos, mass, n, paint_box, window, scale=scale, method=tiled, defer_fold=True,
                           offset="mean")
        return finish_power(*power_sums_fused(grid, boxsize, halo=halo, binning=binning))
    if pos.dtype == torch.float32 and tiled is not None and bool(_lib.lib().ast_fft64_supported(n)):
        # fp32 particles on a grid without fp32 tile passes (128^3, 2048^3): the grid holds rho - mean (only the discarded DC
        # mode differs), the transform runs in double straight from the fp32 grid
        # (side 2048 takes its fp32 passes.  Their z rows could fold the halo records on load - power_sums_fused64(halo=) -
        # but TWO kernels read every row there, the transform and the low-k sums, and both would fetch the 9 GB of records:
        # 86.6 ms against 83.5 with the paint's own fold kernel, so the grid is folded here)
        grid = paint(pos, mass, n, paint_box, window, scale=scale, method=tiled, offset="mean")
        return finish_power(*power_sums_fused64(grid, boxsize, binning=binning, mean=0.0))
    fast64 = pos.dtype == torch.float64 and tiled is not None and bool(_lib.lib().ast_fft64_supported(n))
    if fast64 and defer_fold64:               # float64: the halo fold inside the double z pass (26.2 vs 26.5 ms at 1024^3)
        grid, halo = paint(pos, mass, n, paint_box, window, scale=scale, method=tiled, defer_fold=True)
        return finish_power(*power_sums_fused64(grid, boxsize, halo=halo, binning=binning))
    return fftpower_1d(paint(pos, mass, n, paint_box, window, scale=scale), boxsize, binning=binning)


def fftpower_1d(field1, boxsize, field2=None, fused=True, binning=None):
    """``FFTPower(first, mode="1d", kmin=2*pi/L[, second])`` for in-memory grids
    (nbodykit call sites: power_spectrum_3d.py:189-224, stats_subfind.py:142-150)."""
    n = field1.shape[0]
    assert tuple(field1.shape) == (n, n, n) and n % 2 == 0
    if fused and field2 is None and fused_power_supported(field1):
        # the grid's mean is removed as the z pass loads the cells (it only feeds the discarded DC mode): an fp32
        # transform of an O(1) mean would leave its round-off on every shell
        mean = total_mass(field1.reshape(-1), 0) / float(field1.numel())
        return finish_power(*power_sums_fused(field1, boxsize, mean=mean, binning=binning))
    if fused and field2 is None and fused_power64_supported(field1, allow_f32=True):
        # float64 cubes of side 128 ... 2048 - and fp32 cubes of the sides the fp32 tile passes do not cover (128, 2048),
        # transformed in double without a float64 copy of the grid
        mean = total_mass(field1.reshape(-1), 0) / float(field1.numel()) if field1.dtype == torch.float32 else None
        return finish_power(*power_sums_fused64(field1, boxsize, binning=binning, mean=mean))
    if field1.dtype == torch.float32:
        # fp32 grids that do not take the fused path above - cross spectra, sizes the tile FFT does not cover: an fp32
        # transform would carry the O(1) mean's round-off into the low shells (2e-6 and worse: only the fused path
        # removes the mean and patches the lowest shells); the transforms run in double instead
        field1 = field1.double()
        field2 = None if field2 is None else field2.double()
    s1 = r2c(field1)
    s2 = None if field2 is None else r2c(field2)
    return finish_power(*power_bin_1d(s1, s2, n, boxsize, binning=binning))


# ------------------------------------------ catalogue meshes: interlacing + compensation
def interlace_compensate(c1, c2, nmesh, window, compensated=True, i0=None, i1=None):
    """In place on c1 (half spectrum of the plain paint): combine with c2 (spectrum of the paint shifted by half
    a cell, or None) and divide by the mass-assignment window - nbodykit CatalogMesh, interlaced / compensated."""
    n = int(nmesh)
    i0 = (0, n) if i0 is None else tuple(i0)
    i1 = (0, n) if i1 is None else tuple(i1)
    assert c1.is_cuda and c1.is_contiguous() and c1.numel() == i0[1] * i1[1] * (n // 2 + 1)
    assert c2 is None or (c2.dtype == c1.dtype and c2.numel() == c1.numel() and c2.is_contiguous())
    if c2 is None and not compensated:
        return c1
    check(_lib.lib().ast_interlace_compensate(ptr(c1), ptr(c2), _CPLX[c1.dtype], n, _lib.WIN[window.lower()],
                                              int(bool(compensated)), int(i0[0]), int(i0[1]), int(i1[0]), int(i1[1]),
                                              stream()), "ast_interlace_compensate")
    return c1


def catalog_mesh_complex(pos, mass, nmesh, boxsize, window="tsc", interlaced=True, compensated=True):
    """delta_k of a particle catalogue like nbodykit ``CatalogMesh(..., Nmesh, BoxSize, window=, interlaced=,
    compensated=).compute(mode="complex")``: the painted field is normalised to 1 + delta (mean weight per cell 1),
    transformed with pmesh's 1/Ng convention, interlaced with a second paint shifted by half a cell and divided
    by the window.  Returns (half spectrum, shotnoise = L^3 sum w^2 / (sum w)^2)."""
    n = int(nmesh)
    wsum = total_mass(mass, pos.shape[0])
    scale = float(n) ** 3 / wsum
    c1 = r2c(paint(pos, mass, n, boxsize, window, scale=scale))
    c2 = r2c(paint(pos, mass, n, boxsize, window, scale=scale, shift=0.5)) if interlaced else None
    interlace_compensate(c1, c2, n, window, compensated)
    if mass is None:
        w2 = float(pos.shape[0])
    else:
        w2 = float(triple_product_sum(mass, mass, torch.ones_like(mass)).item())
    return c1, float(boxsize) ** 3 * w2 / wsum ** 2


def catalog_power_1d(pos1, mass1, nmesh, boxsize, window="tsc", interlaced=True, compensated=True, pos2=None, mass2=None):
    """``FFTPower(CatalogMesh(cat1, ...), mode="1d", kmin=2 pi / L[, second=CatalogMesh(cat2, ...)])``: dict(k, power,
    modes, shotnoise); the shot noise is reported, not subtracted (auto: L^3 sum w^2 / (sum w)^2; cross: 0)."""
    n = int(nmesh)
    c1, sn = catalog_mesh_complex(pos1, mass1, n, boxsize, window, interlaced, compensated)
    c2 = None
    if pos2 is not None:
        c2, _ = catalog_mesh_complex(pos2, mass2, n, boxsize, window, interlaced, compensated)
        sn = 0.0
    res = finish_power(*power_bin_1d(c1, c2, n, boxsize))
    res["shotnoise"] = sn
    return res


# ----------------------------------------------------------------- bispectrum
def c2r(spec, shape, out=None):
    """Unnormalised inverse of :func:`r2c`'s layout: sum_k spec_k e^{ikx} (rocFFT C2R;
    the input spectrum is used as scratch)."""
    n0, n1, n2 = shape
    code = _CPLX[spec.dtype]
    if out is None:
        out = torch.empty(shape, dtype=_TO_REAL[spec.dtype], device=spec.device)
    fft_plan(_lib.FFT_C2R, code, (n0, n1, n2), 1, 1.0, False).execute(spec, out)
    return out


def c2r_tile(spec, work=None, out=None, m_lo=0, m_hi=0, scale=1.0):
    """Unnormalised inverse of :func:`r2c`'s layout through the hand-written tile passes (complex64 cubes of side
    256/512/1024), optionally restricted to the shell m_lo <= |m| < m_hi (the bispectrum estimator's masked inverse
    transform in one call).  ``spec`` is left intact; ``work`` (same shape) is scratch."""
    n = spec.shape[0]
    assert spec.is_cuda and spec.is_contiguous() and spec.dtype == torch.complex64 and tuple(spec.shape) == (n, n, n // 2 + 1)
    work = torch.empty_like(spec) if work is None else work
    out = torch.empty((n, n, n), dtype=torch.float32, device=spec.device) if out is None else out
    check(_lib.lib().ast_fft_tile_c2r_3d(ptr(spec), ptr(work), ptr(out), F32, n, int(m_lo), int(m_hi), float(scale), stream()),
          "ast_fft_tile_c2r_3d")
    return out


def tile_work_pitch(n):
    """Row pitch (complex elements) of the scratch spectra of :func:`c2r_tile_batch`: n/2+1 rounded up to 16 elements, so
    that the 128-byte row pieces of the x / y passes' 16-column tiles are whole lines (at n/2+1 every piece straddles two
    lines, both shared with the neighbouring tiles: 19 % more bytes written and re-read, profiles/r04_bispec_pruning.txt)."""
    return (int(n) // 2 + 1 + 15) // 16 * 16


def c2r_tile_batch(spec, shells, works, outs=None, scale=1.0, xy_batch=None):
    """:func:`c2r_tile` for up to 8 shells ``[(m_lo, m_hi), ...]`` of one spectrum (the shell is the launches' second grid
    dimension).  ``works``: one scratch spectrum per shell, shaped like ``spec`` or (n, n, pitch) with any pitch >= n/2+1
    (:func:`tile_work_pitch`).  xy_batch: shells per launch of the x and y passes (default 1:
    shell by shell, so that a shell's y pass reads its x pass's output out of the Infinity Cache; measured at 512^3:
    eight per launch 8.4 ms for the 31 shells' x / y passes against 6.7); the z passes always go in one launch.
    Returns the real fields."""
    n = spec.shape[0]
    k = len(shells)
    assert 1 <= k <= 8 and len(works) >= k
    pitch = int(works[0].shape[-1])
    for w in works[:k]:
        assert w.is_cuda and w.is_contiguous() and w.dtype == torch.complex64 and tuple(w.shape) == (n, n, pitch) and pitch >= n // 2 + 1
    assert spec.is_cuda and spec.is_contiguous() and spec.dtype == torch.complex64 and tuple(spec.shape) == (n, n, n // 2 + 1)
    if outs is None:
        outs = [torch.empty((n, n, n), dtype=torch.float32, device=spec.device) for _ in range(k)]
    wp = (ct.c_void_p * k)(*[w.data_ptr() for w in works[:k]])
    op = (ct.c_void_p * k)(*[o.data_ptr() for o in outs[:k]])
    lo = (ct.c_int * k)(*[int(s[0]) for s in shells])
    hi = (ct.c_int * k)(*[int(s[1]) for s in shells])
    L = _lib.lib()
    if xy_batch is None:
        import os
        xy_batch = int(os.environ.get("ASTRILD_BISPEC_XY_BATCH", "1"))
    xy_batch = max(1, min(int(xy_batch), k))
    if xy_batch >= k:
        check(L.ast_fft_tile_c2r_3d_batch(ptr(spec), wp, op, F32, n, lo, hi, k, float(scale), 3, pitch, stream()), "ast_fft_tile_c2r_3d_batch")
        return list(outs[:k])
    for b0 in range(0, k, xy_batch):
        kb = min(xy_batch, k - b0)
        sub = lambda arr, typ: (typ * kb)(*arr[b0:b0 + kb])
        check(L.ast_fft_tile_c2r_3d_batch(ptr(spec), sub(wp, ct.c_void_p), sub(op, ct.c_void_p), F32, n, sub(lo, ct.c_int),
                                          sub(hi, ct.c_int), kb, float(scale), 1, pitch, stream()), "ast_fft_tile_c2r_3d_batch")
    check(L.ast_fft_tile_c2r_3d_batch(ptr(spec), wp, op, F32, n, lo, hi, k, float(scale), 2, pitch, stream()), "ast_fft_tile_c2r_3d_batch")
    return list(outs[:k])


def shell_filter(spec, nmesh, m_lo, m_hi, out=None, i0=None, i1=None, dtype=None):
    """out = spec * 1[m_lo <= |m| < m_hi]; spec=None writes the bare indicator."""
    n = int(nmesh)
    i0 = (0, n) if i0 is None else tuple(i0)
    i1 = (0, n) if i1 is None else tuple(i1)
    if out is None:
        cd = spec.dtype if spec is not None else dtype
        out = torch.empty((i0[1], i1[1], n // 2 + 1), dtype=cd, device=device())
    check(_lib.lib().ast_shell_filter(ptr(spec), ptr(out), _CPLX[out.dtype], n, int(m_lo), int(m_hi),
                                      int(i0[0]), int(i0[1]), int(i1[0]), int(i1[1]), stream()),
          "ast_shell_filter")
    return out


def triple_product_sum(a, b, c):
    out = torch.zeros(1, dtype=torch.float64, device=a.device)
    check(_lib.lib().ast_triple_product_sum(ptr(a), ptr(b), ptr(c), real_code(a), a.numel(), ptr(out), stream()),
          "ast_triple_product_sum")
    return out


def triple_product_sums(fields, triangles):
    """sum over cells of f_i f_j f_l for every (i, j, l) in ``triangles`` (keys of the dict ``fields``), every field read
    once per batch of <= 256 triangles (ast_triple_product_sums).  Returns a float64 tensor, one entry per triangle."""
    L = _lib.lib()
    triangles = [tuple(t) for t in triangles]
    out = torch.zeros(len(triangles), dtype=torch.float64, device=device())
    scratch = torch.empty(L.ast_triple_product_sums_scratch_bytes() // 8, dtype=torch.float64, device=device())
    for b0 in range(0, len(triangles), 256):
        batch = triangles[b0:b0 + 256]
        used = sorted({s for t in batch for s in t})
        first = fields[used[0]]
        if len(used) * 257 * first.element_size() > 160 * 1024:          # more shells than one LDS chunk holds
            for n, (i, j, l) in enumerate(batch):
                out[b0 + n:b0 + n + 1] = triple_product_sum(fields[i], fields[j], fields[l])
            continue
        for s in used:
            f = fields[s]
            assert f.is_contiguous() and f.dtype == first.dtype and f.numel() == first.numel()
        slot = {s: n for n, s in enumerate(used)}
        ptrs = torch.tensor([fields[s].data_ptr() for s in used], dtype=torch.int64).to(device())
        tri = torch.tensor([[slot[s] for s in t] for t in batch], dtype=torch.int32).to(device())
        check(L.ast_triple_product_sums(ptr(ptrs), len(used), real_code(first), first.numel(), ptr(tri), len(batch),
                                        ptr(scratch), ptr(out[b0:]), stream()), "ast_triple_product_sums")
    return out


_tri_cache = {}


def bispectrum(field, boxsize, edges, triangles):
    """FFT (Scoccimarro) bispectrum estimator on integer-|m| shells [edges[i], edges[i+1]).

    Returns dict(B, ntri, k) with one entry per (i, j, l) in ``triangles``; ntri is
    the exact integer count of closed triangles.  The reference's Bispectrum3D has
    no bispectrum arithmetic (bispectrum_3d.py:165-215 computes P(k)); this is the
    estimator its docstring cites (:42-44).
    """
    n = field.shape[0]
    assert tuple(field.shape) == (n, n, n) and n % 2 == 0
    edges = [int(e) for e in edges]
    triangles = [tuple(int(v) for v in t) for t in triangles]
    used = sorted({s for t in triangles for s in t})
    spec = r2c(field)
    tile = spec.dtype == torch.complex64 and bool(_lib.lib().ast_fft_tile_supported(F32, n))
    # (the tile passes' scratch spectrum has line-aligned rows)
    scratch = torch.empty((n, n, tile_work_pitch(n)), dtype=spec.dtype, device=spec.device) if tile else torch.empty_like(spec)
    key = (torch.cuda.current_device(), n, tuple(edges), tuple(triangles))
    ntri = _tri_cache.get(key)
    if ntri is None:
        # Triangle counts depend on the geometry only; they are computed ONCE per (N, edges, triangles), always in
        # float64: I_s(0) equals the shell's mode count (1e6 at 512^3), and the fp32 round-off of that one cell
        # alone would move sum I_i I_j I_l / Ng by thousands.  In double the sum is within ~1e-3 of the integer.
        iscratch = torch.empty(spec.shape, dtype=torch.complex128, device=spec.device)
        ifields = {}
        L = _lib.lib()
        forward_only = bool(L.ast_fft64_supported(n))
        for s in used:
            if forward_only:
                # the indicator is real and even: its inverse transform is its forward one (hand-written double
                # passes, no rocFFT), real up to round-off, unfolded from the half lattice onto the full one
                mask = torch.empty((n, n, n), dtype=torch.float64, device=spec.device)
                check(L.ast_shell_mask_real(ptr(mask), n, edges[s], edges[s + 1], stream()), "ast_shell_mask_real")
                check(L.ast_fft64_r2c_3d(ptr(mask), ptr(iscratch), n, 1.0, stream()), "ast_fft64_r2c_3d")
                check(L.ast_half_real_to_full(ptr(iscratch), ptr(mask), n, stream()), "ast_half_real_to_full")
                ifields[s] = mask
                continue
            shell_filter(None, n, edges[s], edges[s + 1], out=iscratch)
            ifields[s] = c2r(iscratch, (n, n, n))
        dens = triple_product_sums(ifields, triangles).cpu().numpy()
        del ifields, iscratch
        exact = dens / float(n) ** 3
        ntri = np.rint(exact).astype(np.int64)
        worst = float(np.abs(exact - ntri).max()) if len(exact) else 0.0
        if worst > 0.05:
            raise _lib.AstrildHipError(f"triangle counts are not integers to 0.05 (worst {worst:.3g})")
        _tri_cache[key] = ntri
        _tri_cache[key + ("residual",)] = worst
    # The triangle sums form f_i f_j f_l of fp32 fields in fp32 (only the running sums are double): a field in physical units
    # (a mass-weighted grid in Msun/h per cell) would overflow the product above |D| ~ 7e12.  The shell fields are therefore
    # built from the spectrum divided by A = max |field| - values of order one - and the sums multiplied back by A^3.
    amp = 1.0
    if field.dtype == torch.float32 and field.numel():
        lo_hi = torch.empty(2, dtype=torch.float64, device=field.device)
        check(_lib.lib().ast_minmax(ptr(field), real_code(field), field.numel(), ptr(lo_hi), stream()), "ast_minmax")
        amp = float(lo_hi.abs().max()) or 1.0
        if not np.isfinite(amp):
            raise _lib.AstrildHipError("bispectrum: the field holds inf / NaN")
    # FUSED tail (default where it applies: fp32 tile sizes, <= 32 shells): every shell's masked x and y passes into its OWN
    # scratch spectrum, then ONE kernel that runs the z passes of all shells row by row and forms the triangle sums from
    # LDS (ast_fft_tile_c2r_triangles) - the 31 real cubes (0.5 GB each at 512^3, written once and read back once) never
    # exist.  ASTRILD_BISPEC_FUSED=0: the cubes and ast_triple_product_sums, as before.
    import os
    if tile and len(used) <= 32 and os.environ.get("ASTRILD_BISPEC_FUSED", "1") != "0":
        L = _lib.lib()
        pitch = tile_work_pitch(n)
        works = [scratch] + [torch.empty_like(scratch) for _ in range(len(used) - 1)]
        wp = (ct.c_void_p * len(used))(*[w.data_ptr() for w in works])
        hi = (ct.c_int * len(used))(*[edges[s + 1] for s in used])
        for i, s_ in enumerate(used):           # shell by shell: a shell's y pass finds its x pass's output in the Infinity Cache
            one_w = (ct.c_void_p * 1)(works[i].data_ptr())
            one_o = (ct.c_void_p * 1)(None)                          # (no output: passes = 1 stops before the z pass)
            check(L.ast_fft_tile_c2r_3d_batch(ptr(spec), one_w, one_o, F32, n, (ct.c_int * 1)(edges[s_]), (ct.c_int * 1)(edges[s_ + 1]),
                                              1, 1.0, 1, pitch, stream()), "ast_fft_tile_c2r_3d_batch")
        slot = {s_: i for i, s_ in enumerate(used)}
        tri_d = torch.tensor([[slot[v] for v in t] for t in triangles], dtype=torch.int32).to(spec.device)
        out = torch.empty(len(triangles), dtype=torch.float64, device=spec.device)
        fscr = torch.empty(int(L.ast_fft_tile_c2r_triangles_scratch_bytes()) // 8, dtype=torch.float64, device=spec.device)
        for b0 in range(0, len(triangles), 512):
            nb_ = min(512, len(triangles) - b0)
            check(L.ast_fft_tile_c2r_triangles(wp, hi, len(used), F32, n, pitch, 1.0 / amp, ptr(tri_d[b0:]), nb_, ptr(fscr), ptr(out[b0:]),
                                               stream()), "ast_fft_tile_c2r_triangles")
        num = out.cpu().numpy() * amp ** 3
        del works
        kf = 2.0 * np.pi / boxsize
        kmid = np.array([[kf * 0.5 * (edges[s] + edges[s + 1]) for s in t] for t in triangles])
        with np.errstate(invalid="ignore", divide="ignore"):
            b = float(boxsize) ** 6 * num / (ntri * float(n) ** 3)
        return {"B": b, "ntri": ntri, "k": kmid, "ntri_residual": _tri_cache[key + ("residual",)]}
    dfields = {}
    if tile:
        # the masked, pruned inverse tile passes (shell mask fused into the first pass's loads), shell by shell through ONE
        # scratch spectrum.  ASTRILD_BISPEC_BATCH=k runs k shells per launch (ast_fft_tile_c2r_3d_batch) - measured at
        # 512^3 / 31 shells: the z passes gain (5.9 -> 5.1 ms, the small shells fill the large ones' tails) but the x / y
        # passes lose more (6.9 -> 8.0 ms: k scratch spectra instead of one that stays in the Infinity Cache)
        import os
        batch = max(1, min(8, int(os.environ.get("ASTRILD_BISPEC_BATCH", "1")))) if len(used) > 1 else 1
        works = [scratch] + [torch.empty_like(scratch) for _ in range(min(batch, len(used)) - 1)]
        for b0 in range(0, len(used), batch):
            group = used[b0:b0 + batch]
            fields = c2r_tile_batch(spec, [(edges[s], edges[s + 1]) for s in group], works, scale=1.0 / amp)
            dfields.update(zip(group, fields))
        del works
    if not tile and amp != 1.0:
        sr = torch.view_as_real(spec)
        check(_lib.lib().ast_divide(ptr(sr), real_code(sr), sr.numel(), amp, stream()), "ast_divide")
    for s in ([] if tile else used):
        shell_filter(spec, n, edges[s], edges[s + 1], out=scratch)
        dfields[s] = c2r(scratch, (n, n, n))
    num = triple_product_sums(dfields, triangles).cpu().numpy() * amp ** 3
    kf = 2.0 * np.pi / boxsize
    kmid = np.array([[kf * 0.5 * (edges[s] + edges[s + 1]) for s in t] for t in triangles])
    with np.errstate(invalid="ignore", divide="ignore"):
        b = float(boxsize) ** 6 * num / (ntri * float(n) ** 3)
    return {"B": b, "ntri": ntri, "k": kmid, "ntri_residual": _tri_cache[key + ("residual",)]}
