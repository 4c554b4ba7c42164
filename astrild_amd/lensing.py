"""HBM-resident kappa-map operations.

Thin wrappers over the C-ABI: plane stack with lensing-kernel re-weighting,
unit conversion, Gaussian smoothing, kappa -> deflection / potential, PDF
histogram.  fp64 like the reference; torch tensors are memory holders only.
"""
import ctypes as ct
import time

import numpy as np
import torch

from . import _lib
from ._lib import F64, check
from .device import as_device, device, ptr, real_code, stream


# ------------------------------------------------------------------ a-7 stack
def kernel_function(x, x_s):
    """g(x, x_s) = (x_s - x) * x / x_s  (rayramses.py:315-326, simcoll.py:432-443)."""
    return (x_s - x) * x / x_s


def translate_redshift_weights(x_near, x_far, x_src, x_src_shift):
    """(numerator, denominator) of the re-weighting of rayramses.py:269-312:
    quantity * g(x_mid, x_s') / g(x_mid, x_s) with x_mid = (x_far + x_near)/2 and the
    clamp x_s' = x_far when the next snapshot lies beyond the shifted source."""
    x_near, x_far = np.asarray(x_near, dtype=np.float64), np.asarray(x_far, dtype=np.float64)
    x_shift = np.where(x_far > x_src_shift, x_far, x_src_shift)
    x_mid = 0.5 * (x_far + x_near)
    return kernel_function(x_mid, x_shift), kernel_function(x_mid, x_src)


def kappa_stack(planes, wnum=None, wden=None, out=None):
    """out = sum_p planes[p] * wnum[p] / wden[p] in plane order (first = copy, then
    running +=), like rayramses.py:224-232 / simcoll.py:322-336.

    planes: list of equal-shape CUDA tensors (or one (P, ...) tensor)."""
    if isinstance(planes, torch.Tensor):
        planes = [planes[p] for p in range(planes.shape[0])]
    assert len(planes) >= 1
    first = planes[0]
    code = real_code(first)
    for t in planes:
        assert t.is_cuda and t.is_contiguous() and t.dtype == first.dtype and t.shape == first.shape
    if out is None:
        out = torch.empty_like(first)
    # pointer table and weights travel in ONE non-blocking copy from pinned memory: a pageable host-to-device copy is
    # stream ordered AND blocks the host, i.e. the host would sit out everything still queued (the previous map)
    P = len(planes)
    host = torch.empty(3 * P, dtype=torch.int64, pin_memory=True)
    host[:P] = torch.tensor([t.data_ptr() for t in planes], dtype=torch.int64)
    if wnum is not None:
        wn_h, wd_h = np.asarray(wnum, dtype=np.float64), np.asarray(wden, dtype=np.float64)
        assert wn_h.size == P == wd_h.size
        host[P:2 * P] = torch.from_numpy(np.ascontiguousarray(wn_h)).view(torch.int64)
        host[2 * P:] = torch.from_numpy(np.ascontiguousarray(wd_h)).view(torch.int64)
    table = host.to(first.device, non_blocking=True)
    wn = table[P:2 * P].view(torch.float64) if wnum is not None else None
    wd = table[2 * P:].view(torch.float64) if wnum is not None else None
    aligned = all(t.data_ptr() % 16 == 0 for t in planes) and out.data_ptr() % 16 == 0
    check(_lib.lib().ast_kappa_stack(ptr(table), ptr(wn), ptr(wd), P, first.numel(), code, ptr(out),
                                     int(aligned), stream()), "ast_kappa_stack")
    return out


# ------------------------------------------------------------- a-8 per-map ops
C_LIGHT_KMS = 299792.458       # astropy.constants.c.to("km/s").value (sky_utils.py:17)


def convert_code_to_phy_units(quantity, t):
    """In place: / c^2 for shear/deflt/kappa_2, / c^3 for isw_rs (sky_utils.py:318-339)."""
    if quantity in ["shear_x", "shear_y", "deflt_x", "deflt_y", "kappa_2"]:
        div = C_LIGHT_KMS ** 2
    elif quantity in ["isw_rs"]:
        div = C_LIGHT_KMS ** 3
    else:
        return t
    check(_lib.lib().ast_divide(ptr(t), real_code(t), t.numel(), div, stream()), "ast_divide")
    return t


class _Plan:
    _create = _destroy = None

    def __init__(self, *args):
        self.handle = ct.c_void_p()
        check(getattr(_lib.lib(), self._create)(ct.byref(self.handle), *args), self._create)

    def __del__(self):
        try:
            if self.handle:
                getattr(_lib.lib(), self._destroy)(self.handle)
                self.handle = ct.c_void_p()
        except Exception:
            pass


class LensPlan(_Plan):
    """kappa -> (alpha1, alpha2) / phi for an nc x nc map of side bsz [rad]; caches
    the kernel spectra of lensing_funcs.c:45-83,117-148 on the device."""
    _create, _destroy = "ast_lens_plan_create", "ast_lens_plan_destroy"

    def __init__(self, nc, bsz):
        super().__init__(int(nc), float(bsz))
        self.nc = int(nc)

    def alphas(self, kappa):
        assert kappa.is_cuda and kappa.dtype == torch.float64 and kappa.numel() == self.nc ** 2
        kappa = kappa.contiguous()
        a1, a2 = torch.empty_like(kappa), torch.empty_like(kappa)
        check(_lib.lib().ast_kappa_to_alphas(self.handle, ptr(kappa), ptr(a1), ptr(a2), stream()),
              "ast_kappa_to_alphas")
        return a1, a2

    def phi(self, kappa):
        assert kappa.is_cuda and kappa.dtype == torch.float64 and kappa.numel() == self.nc ** 2
        kappa = kappa.contiguous()
        out = torch.empty_like(kappa)
        check(_lib.lib().ast_kappa_to_phi(self.handle, ptr(kappa), ptr(out), stream()), "ast_kappa_to_phi")
        return out


class SmoothPlan(_Plan):
    _create, _destroy = "ast_smooth_plan_create", "ast_smooth_plan_destroy"

    def __init__(self, npix):
        super().__init__(int(npix))
        self.npix = int(npix)

    def gaussian(self, img, sigma_px, kind="gaussianFFT"):
        """In place.  kind: "gaussianFFT" (periodic) or "gaussian" (real space,
        reflect, truncate 4) — lenstools ConvergenceMap.smooth's two branches; "gaussian_mirror": the real-space
        kernel with scipy's "mirror" boundary (the prefilter of skimage.transform.resize)."""
        assert img.is_cuda and img.dtype == torch.float64 and img.numel() == self.npix ** 2 and img.is_contiguous()
        mode = {"gaussianFFT": 0, "gaussian": 1, "gaussian_mirror": 2}[kind]
        check(_lib.lib().ast_gaussian_smooth(self.handle, ptr(img), float(sigma_px), mode, stream()),
              "ast_gaussian_smooth")
        return img


_lens_plans, _smooth_plans = {}, {}


def lens_plan(nc, bsz):
    key = (torch.cuda.current_device(), int(nc), float(bsz))
    if key not in _lens_plans:
        _lens_plans.clear()          # one resident plan: the spectra are 3 x 0.5 GB at nc = 4096
        _lens_plans[key] = LensPlan(nc, bsz)
    return _lens_plans[key]


def smooth_plan(npix):
    key = (torch.cuda.current_device(), int(npix))
    if key not in _smooth_plans:
        _smooth_plans[key] = SmoothPlan(npix)
    return _smooth_plans[key]


def resize_antialiased(img, npix):
    """``skimage.transform.resize(img, (npix, npix), anti_aliasing=True)`` of a square fp64 map for npix <= len(img), the
    call behind ``SkyArray.resize`` (sky_array.py:475-496).  scikit-image is not in the reference's lock file; this is
    the algorithm of scikit-image >= 0.19: ``scipy.ndimage.gaussian_filter`` (truncate 4) with sigma =
    (len(img) / npix - 1) / 2, then ``scipy.ndimage.zoom(order=1, grid_mode=True)`` - both on the device, both in
    ndimage's "mirror" boundary: resize's default ``mode="reflect"`` is numpy.pad's naming, which skimage's
    ``_to_ndimage_mode`` translates to ndimage "mirror" (edge pixel not repeated).  For npix <= len(img) the zoom's
    sample points stay inside the map, so only the prefilter sees the boundary.  Returns a device tensor; the input
    is left alone."""
    npix = int(npix)
    if torch.is_tensor(img):
        t = img.to(device="cuda", dtype=torch.float64).contiguous().clone()
    else:
        from .device import as_device
        t = as_device(np.ascontiguousarray(img, dtype=np.float64))
    if t.dim() != 2 or t.shape[0] != t.shape[1]:
        raise ValueError("resize_antialiased: a square 2-D map is expected")
    nin = int(t.shape[0])
    if not 1 <= npix <= nin:
        raise NotImplementedError("resize_antialiased lowers the pixel count (sky_array.py:484): 1 <= npix <= len(img)")
    sigma = max(0.0, (nin / npix - 1.0) / 2.0)
    if sigma > 0.0:
        smooth_plan(nin).gaussian(t, sigma, "gaussian_mirror")
    out = torch.empty((npix, npix), dtype=torch.float64, device=t.device)
    check(_lib.lib().ast_zoom_linear(ptr(t), nin, ptr(out), npix, stream()), "ast_zoom_linear")
    return out


def deflection_to_shear(alpha1, alpha2, h):
    """(gamma1, gamma2) of SkyUtils.convert_deflection_to_shear (sky_utils.py:342-362): np.gradient of both deflection
    components at uniform spacing h (the pixel size in the unit of alpha) and the reference's combinations of them."""
    a1, a2 = (t if torch.is_tensor(t) else None for t in (alpha1, alpha2))
    from .device import as_device
    a1 = as_device(np.ascontiguousarray(alpha1, dtype=np.float64)) if a1 is None else a1.to(device="cuda", dtype=torch.float64).contiguous()
    a2 = as_device(np.ascontiguousarray(alpha2, dtype=np.float64)) if a2 is None else a2.to(device="cuda", dtype=torch.float64).contiguous()
    if a1.dim() != 2 or a1.shape[0] != a1.shape[1] or a1.shape != a2.shape:
        raise ValueError("deflection_to_shear: two square maps of equal size are expected")
    g1, g2 = torch.empty_like(a1), torch.empty_like(a1)
    check(_lib.lib().ast_deflection_to_shear(ptr(a1), ptr(a2), int(a1.shape[0]), float(h), ptr(g1), ptr(g2), stream()),
          "ast_deflection_to_shear")
    return g1, g2


def minmax(t):
    out = torch.empty(2, dtype=torch.float64, device=t.device)
    check(_lib.lib().ast_minmax(ptr(t), real_code(t), t.numel(), ptr(out), stream()), "ast_minmax")
    lo, hi = out.cpu().numpy()
    return float(lo), float(hi)


class PendingHistogram:
    """np.histogram(t, bins=nbins, range=None, density=) in flight: the min/max pass, the counting pass and the copy of
    (counts, range) into pinned host memory are queued on the stream; ``result()`` waits for THAT copy only - kernels
    queued after it keep running - and returns (values, bin_edges) as numpy."""

    def __init__(self, t, nbins, density=False):
        self.nbins, self.density = int(nbins), density
        buf = torch.zeros(self.nbins + 2, dtype=torch.int64, device=t.device)
        check(_lib.lib().ast_histogram_auto(ptr(t), real_code(t), t.numel(), self.nbins, ptr(buf), ptr(buf[self.nbins:]),
                                            stream()), "ast_histogram_auto")
        self.host = torch.empty(self.nbins + 2, dtype=torch.int64, pin_memory=True)
        self.host.copy_(buf, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        self._keep = buf

    def result(self):
        self.event.synchronize()
        host = self.host.numpy()
        counts = host[:self.nbins].copy()
        lo, hi = (float(v) for v in host[self.nbins:].view(np.float64))
        if lo == hi:                  # numpy widens a degenerate range by +-0.5
            lo, hi = lo - 0.5, hi + 0.5
        edges = np.linspace(lo, hi, self.nbins + 1)
        if self.density:
            return counts / np.diff(edges) / counts.sum(), edges
        return counts, edges


def histogram(t, nbins, range=None, density=False):
    """np.histogram(t, bins=nbins, range=, density=) -> (values, bin_edges) as numpy."""
    nbins = int(nbins)
    if range is None:
        # min/max and counts in one stream-ordered call; ONE transfer brings back both (counts, then the range's bits)
        return PendingHistogram(t, nbins, density).result()
    lo, hi = float(range[0]), float(range[1])
    if lo == hi:
        lo, hi = lo - 0.5, hi + 0.5
    counts = torch.zeros(nbins, dtype=torch.int64, device=t.device)
    check(_lib.lib().ast_histogram(ptr(t), real_code(t), t.numel(), lo, hi, nbins, ptr(counts), stream()),
          "ast_histogram")
    counts = counts.cpu().numpy()
    edges = np.linspace(lo, hi, nbins + 1)
    if density:
        return counts / np.diff(edges) / counts.sum(), edges
    return counts, edges


def order_statistics(t, ks):
    """Exact k-th smallest elements (0-based) of a device tensor - radix select on the GPU (ast_order_statistics)."""
    ks = [int(k) for k in ks]
    karr = (ct.c_size_t * len(ks))(*ks)
    out = (ct.c_double * len(ks))()
    scratch = torch.empty(2048, dtype=torch.int64, device=t.device)
    check(_lib.lib().ast_order_statistics(ptr(t), real_code(t), t.numel(), karr, len(ks), out, ptr(scratch), stream()),
          "ast_order_statistics")
    return [float(v) for v in out]


def percentile(t, qs):
    """np.percentile(t, q) (default "linear" method) for each q, from exact order statistics:
    numpy's virtual index (n - 1) * q, gamma = its fractional part and numpy's two-sided lerp."""
    n = t.numel()
    res = []
    for q in qs:
        qf = np.true_divide(q, 100)
        vi = (n - 1) * qf
        prev = int(np.floor(vi))
        gamma = vi - prev
        prev = min(max(prev, 0), n - 1)
        a, b = order_statistics(t, [prev, min(prev + 1, n - 1)])
        diff = b - a
        v = a + diff * gamma
        if gamma >= 0.5:
            v = b - diff * (1 - gamma)
        res.append(float(v))
    return res


def peak_find(t, lo=-np.inf, hi=np.inf):
    """Heights and flat pixel indices of the strict 8-neighbour local maxima of the interior of a square map with
    lo <= height < hi, in row-major scan order (lenstools ConvergenceMap.locatePeaks; numpy arrays)."""
    assert t.dim() == 2 and t.shape[0] == t.shape[1] and t.is_contiguous()
    npix = t.shape[0]
    cap = max(1024, t.numel() // 8)
    while True:
        values = torch.empty(cap, dtype=t.dtype, device=t.device)
        index = torch.empty(cap, dtype=torch.int64, device=t.device)
        count = torch.zeros(1, dtype=torch.int64, device=t.device)
        check(_lib.lib().ast_peak_find(ptr(t), real_code(t), npix, float(lo), float(hi), cap, ptr(values), ptr(index),
                                       ptr(count), stream()), "ast_peak_find")
        n = int(count.item())
        if n <= cap:
            break
        cap = n                                   # a strict maximum excludes its neighbours: n <= numel / 4 in any case
    values, index = values[:n].cpu().numpy(), index[:n].cpu().numpy()
    order = np.argsort(index, kind="stable")
    return values[order], index[order]


# --------------------------------------------- f-3 flat-sky spectra (lenstools restated)
def _rfft2(t):
    from . import device as dev
    n = t.shape[0]
    out = torch.empty((n, n // 2 + 1), dtype=torch.complex128, device=t.device)
    dev.fft_plan(_lib.FFT_R2C, F64, (n, n), 1, 1.0, False).execute(t, out)
    return out


def flat_power_spectrum(img, angle_deg, l_edges, img2=None):
    """lenstools ``ConvergenceMap.powerSpectrum(l_edges)`` (angular_power_spectrum.py:38-53): (l, P_l) with l the bin
    centres; bins (l_k, l_k+1], mean of |rfft2|^2 over the bin's half-plane pixels times (angle / npix^2)^2; empty
    bins are 0 like lenstools' (it only divides where there are hits)."""
    l_edges = np.asarray(l_edges, dtype=np.float64)
    nb = len(l_edges) - 1
    t = as_device(np.ascontiguousarray(img, dtype=np.float64)) if not isinstance(img, torch.Tensor) else img
    n = t.shape[0]
    assert t.dim() == 2 and t.shape[1] == n and t.dtype == torch.float64
    ft1 = _rfft2(t.contiguous())
    ft2 = None if img2 is None else _rfft2(as_device(np.ascontiguousarray(img2, dtype=np.float64)))
    angle = np.deg2rad(angle_deg)
    edges = as_device(l_edges)
    psum = torch.zeros(nb, dtype=torch.float64, device=t.device)
    hits = torch.zeros(nb, dtype=torch.int64, device=t.device)
    check(_lib.lib().ast_flat_power_bin(ptr(ft1), ptr(ft2), n, float(angle), ptr(edges), nb, ptr(psum), ptr(hits),
                                        stream()), "ast_flat_power_bin")
    psum, hits = psum.cpu().numpy(), hits.cpu().numpy()
    power = np.where(hits > 0, psum / np.maximum(hits, 1), psum) * (angle / float(n) ** 2) ** 2
    return 0.5 * (l_edges[:-1] + l_edges[1:]), power


def flat_bispectrum_equilateral(img, angle_deg, l_edges):
    """lenstools ``ConvergenceMap.bispectrum(l_edges, configuration="equilateral")`` (bispectrum_2d.py:33-50) by the
    FFT estimator: per bin ring-filter the spectrum, transform back, sum the cube; triangle counts from the bare
    rings.  B = angle^4 / npix^6 * <ft ft ft> over closed triangles with all sides in the bin; bins without a
    closed triangle are 0.  Returns (l, B_l, integer triangle counts)."""
    from . import device as dev
    l_edges = np.asarray(l_edges, dtype=np.float64)
    nb = len(l_edges) - 1
    t = as_device(np.ascontiguousarray(img, dtype=np.float64)) if not isinstance(img, torch.Tensor) else img
    n = t.shape[0]
    ft = _rfft2(t.contiguous())
    ring = torch.empty_like(ft)
    field = torch.empty((n, n), dtype=torch.float64, device=t.device)
    angle = float(np.deg2rad(angle_deg))
    c2r = dev.fft_plan(_lib.FFT_C2R, F64, (n, n), 1, 1.0, False)
    num, den = np.zeros(nb), np.zeros(nb)
    for k in range(nb):
        for src, acc in ((ft, num), (None, den)):
            check(_lib.lib().ast_ring_filter_2d(ptr(src), ptr(ring), n, angle, float(l_edges[k]), float(l_edges[k + 1]),
                                                stream()), "ast_ring_filter_2d")
            c2r.execute(ring, field)
            acc[k] = float(dev.triple_product_sum(field, field, field).item())
    npx = float(n) ** 2
    ntri = np.rint(den / npx).astype(np.int64)
    with np.errstate(invalid="ignore", divide="ignore"):
        b = np.where(ntri > 0, (num / npx) / np.maximum(ntri, 1), 0.0) * angle ** 4 / float(n) ** 6
    return 0.5 * (l_edges[:-1] + l_edges[1:]), b, ntri


def add(a, b, out=None):
    out = torch.empty_like(a) if out is None else out
    check(_lib.lib().ast_add(ptr(a), ptr(b), ptr(out), real_code(a), a.numel(), stream()), "ast_add")
    return out


# --------------------------------------------------- f-2 NFW halo stamps
def nfw_paint(halo_cat, extent, direction, suppress, suppression_R, npix, signal, out=None):
    """Add the NFW deflection-angle ("alpha") or moving-lens temperature ("dT") stamp of
    every halo of ``halo_cat`` (dict of equal-length sequences with astrild's keys) onto an
    npix x npix map — SkyUtils.analytic_Halo_signal_to_SkyArray (sky_utils.py:79-137)."""
    nh = len(halo_cat["m200"])
    f64 = lambda key: as_device(np.ascontiguousarray(np.asarray(halo_cat[key], dtype=np.float64)))
    i32 = lambda arr: as_device(np.ascontiguousarray(np.asarray(arr, dtype=np.int32)))
    if out is None:
        out = torch.zeros((npix, npix), dtype=torch.float64, device=device())
    if nh == 0:
        return out
    sig = {"alpha": 0, "dT": 1}[signal]
    mask = sum(1 << int(d) for d in set(int(d) for d in direction))
    if sig == 0 and int(np.sum(list(direction))) > 1:
        raise AssertionError("Only 0 and 1 are valid direction indications.")
    r200_pix = np.asarray(halo_cat["r200_pix"])
    stamp = np.array([int(2 * r * extent) + 1 for r in r200_pix])       # sky_utils.py:112,130
    dist = as_device(np.ascontiguousarray(np.asarray(halo_cat["Dc"], dtype=np.float64) * 0.6774))
    vx = f64("theta1_tv") if sig == 1 else None
    vy = f64("theta2_tv") if sig == 1 else None
    keep = [f64("r200_deg"), f64("m200"), f64("c_NFW"), dist, vx, vy, i32(stamp), i32(halo_cat["theta1_pix"]),
            i32(halo_cat["theta2_pix"])]
    for start in range(0, nh, 65535):                                  # grid.y limit
        n = min(65535, nh - start)
        sl = [None if t is None else t[start:start + n] for t in keep]
        check(_lib.lib().ast_nfw_paint(*[ptr(t) for t in sl], n, float(extent), mask, int(bool(suppress)),
                                       float(suppression_R), sig, ptr(out), int(npix), stream()), "ast_nfw_paint")
    return out


def add_patch(limg, simg, cen_pix):
    """limg[y, x] += simg, clipped at the boundary (SkyUtils.add_patch_to_map, sky_utils.py:140-173)."""
    check(_lib.lib().ast_add_patch(ptr(limg), limg.shape[0], ptr(simg), simg.shape[0], int(cen_pix[0]),
                                   int(cen_pix[1]), stream()), "ast_add_patch")
    return limg


# ------------------------------------------------- synthetic planes + bench leg
def synth_kappa_plane(p, npix, seed0=4242, rms=0.01, dtype=torch.float64):
    """Plane p of the synthetic stack: a Gaussian random field with P(l) ~ (l + l0)^-2,
    seeded by seed0 + p.  Generated with torch's FFT as bench plumbing (inputs are not part of the path)."""
    fy = torch.fft.fftfreq(npix, device="cuda", dtype=torch.float64)[:, None]
    fx = torch.fft.rfftfreq(npix, device="cuda", dtype=torch.float64)[None, :]
    amp = 1.0 / (torch.sqrt(fx * fx + fy * fy) * npix + 10.0)
    g = torch.Generator(device="cuda").manual_seed(seed0 + p)
    white = torch.randn((npix, npix), generator=g, device="cuda", dtype=torch.float64)
    f = torch.fft.irfft2(torch.fft.rfft2(white) * amp, s=(npix, npix))
    return (f * (rms / f.std())).to(dtype).contiguous()


def synth_kappa_planes(nplanes, npix, seed0=4242, rms=0.01, dtype=torch.float64, ids=None):
    """The planes `ids` (default: all) of the synthetic stack, as views into one allocation (device.upload_planes: the stack
    reads one pixel of every plane at a time and runs 8-10 % faster over one large allocation than over separate blocks)."""
    ids = list(range(nplanes) if ids is None else ids)
    slab = torch.empty((len(ids), npix, npix), dtype=dtype, device="cuda")
    for i, p in enumerate(ids):
        slab[i].copy_(synth_kappa_plane(p, npix, seed0, rms, dtype))
    return [slab[i] for i in range(len(ids))]


def synth_plane_weights(nplanes):
    """(wnum, wden) of the synthetic stack: a 1000 Mpc/h box cut into nplanes slabs, source moved from 1100 to 1000."""
    mid = (np.arange(nplanes) + 0.5) * (1000.0 / nplanes)
    half = 0.5 * 1000.0 / nplanes
    return translate_redshift_weights(mid - half, mid + half, 1100.0, 1000.0)


def kappa_map_tail(lp, sp, sigma_px, keep=None):
    """The per-map stages behind the stack (sky_utils.py:318-339, filters.py:181-225, sky_array.py:428-433, 780-817) as one
    callable for kappa_shard.MapStream: / c^2 -> Gaussian FFT smoothing -> 100-bin PDF (queued) -> kappa -> (alpha1, alpha2).
    Nothing in it waits for the GPU; the PDFs are collected from the returned PendingHistogram afterwards."""
    def tail(m, kappa_flat):
        npix = lp.nc
        img = kappa_flat.view(npix, npix)
        convert_code_to_phy_units("kappa_2", img)
        sp.gaussian(img, sigma_px, "gaussianFFT")
        pdf = PendingHistogram(img, 100, density=True)
        a1, a2 = lp.alphas(img)
        if keep is not None:
            keep[m] = (img.clone(), a1, a2)
        return pdf
    return tail


def bench_kappa_pipeline(nplanes=64, npix=4096, steps=3, warmup=1, theta_deg=20.0, sigma_arcmin=1.0, group=None):
    """kappa-maps/s at npix^2 for one full map: stack nplanes planes (weighted) ->
    / c^2 -> Gaussian FFT smoothing -> kappa -> (alpha1, alpha2) -> 100-bin PDF.
    With a process group of P > 1 ranks the planes are sharded (plane p on rank p mod P) and the maps are a STREAM
    (kappa_shard.MapStream; the loops of simcoll.py:267-336 / rayramses.py:186-232 produce one stacked map per
    iteration): map m is reduced onto rank m mod P, which runs its single-map stages on a second stream while all ranks
    stack map m + 1 - a step is then P maps (every rank is root once); the time is the slowest rank's, between barriers."""
    from . import device as dev
    world, rank = 1, 0
    if group is not None:
        import torch.distributed as dist
        from . import kappa_shard
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    ids = list(range(rank, nplanes, world))
    planes = synth_kappa_planes(nplanes, npix, ids=ids)
    wnum, wden = synth_plane_weights(nplanes)
    wnum, wden = wnum[ids], wden[ids]
    bsz = np.deg2rad(theta_deg)
    sigma_px = sigma_arcmin / 60.0 * npix / theta_deg
    lp = lens_plan(npix, bsz)
    sp = smooth_plan(npix)
    out = torch.empty((npix, npix), dtype=torch.float64, device="cuda")
    maps_per_step = world if world > 1 else 1
    stream = kappa_shard.MapStream(npix * npix, group) if world > 1 else None
    tail = kappa_map_tail(lp, sp, sigma_px)

    def step():
        if world > 1:
            # the PDFs are collected after ALL maps have been queued (run(), below): waiting for one here would hold back
            # this rank's part of the next maps' collectives
            for _ in range(maps_per_step):
                stream.push(planes, wnum, wden, tail)
            return []
        kappa_stack(planes, wnum, wden, out=out)
        convert_code_to_phy_units("kappa_2", out)
        sp.gaussian(out, sigma_px, "gaussianFFT")
        pdf = PendingHistogram(out, 100, density=True)          # queued; the host fetches it ONE MAP LATER (below), after the
        a1, a2 = lp.alphas(out)                                  # next map has been enqueued: it never waits on an idle GPU
        return [(pdf, a1, a2)]

    def barrier():
        if world > 1:
            dist.barrier(group)
        torch.cuda.synchronize()

    def collect(item):
        (item[0] if isinstance(item, tuple) else item).result()

    def run(count):
        # one GPU: map m's PDF is collected after map m + 1 has been enqueued (the loops of simcoll.py:267-336 produce a
        # stream of maps; with the fetch right behind its own map the step time depended on how fast the host could
        # refill the queue: 3.4 - 5.2 ms per map from box to box on the same kernels).  P ranks: after all maps.
        pend = []
        for _ in range(count):
            pend += step()
            while world == 1 and len(pend) > 1:
                collect(pend.pop(0))
        if world > 1:
            pend = list(stream.finish().values())
            stream.results = {}
        for item in pend:
            collect(item)

    run(warmup)
    barrier()
    dev.profile_enable(True)
    t0 = time.perf_counter()
    run(steps)
    barrier()
    dt = (time.perf_counter() - t0) / (steps * maps_per_step)
    prof = dev.profile_report()
    dev.profile_enable(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
        dt = float(tmax.item())
    stack_ms = prof.get("kappa_stack", (0, 0.0))[1] / (steps * maps_per_step)
    nloc = len(ids)
    stack_bytes = (nloc + 1) * npix * npix * 8 + (0 if world == 1 else 2 * npix * npix * 8)     # + chunk sums
    # SURVEY.md §8(d) algorithmic bytes of the whole per-map pipeline: stack (P + 1) maps, FFT smoothing 8 map
    # passes, kappa -> alpha 12 passes over the padded (2 npix)^2 array + pad/crop
    px = npix * npix * 8
    pipeline_bytes = (nplanes + 1) * px + 8 * px + (12 * 4 * px + 6 * px)
    res = {
        "metric": f"kappa-maps/s at {npix}^2 ({nplanes}-plane weighted stack + smoothing + kappa->alpha + PDF, fp64)",
        "value": 1.0 / dt, "unit": "maps/s", "ms_per_map": dt * 1e3, "n_gpus": world,
        "planes_per_rank": nloc, "maps_per_step": maps_per_step,
        "mode": "one GPU" if world == 1 else f"stream of maps over {world} ranks: rotating reduce root (map m on rank m mod P), "
                                             "per-map stages on a second stream beside the next maps' stacks",
        "stack": {"ms": round(stack_ms, 4), "alg_GB": round(stack_bytes / 1e9, 3),
                  "GBps": round(stack_bytes / stack_ms / 1e6, 1) if stack_ms else None,
                  "frac": round(stack_bytes / stack_ms / 1e6 / 8000.0, 4) if stack_ms else None},
        "roofline": {"bound": "hbm", "scope": "whole per-map pipeline, all GPUs", "alg_GB": round(pipeline_bytes / 1e9, 3),
                     "achieved": round(pipeline_bytes / dt / 1e9, 1), "peak": 8000.0 * world, "unit": "GB/s",
                     "frac": round(pipeline_bytes / dt / 1e9 / (8000.0 * world), 4)},
        "kernels_ms": {k: round(v[1] / (steps * maps_per_step), 4) for k, v in prof.items()},
    }
    return res


def bench_kappa_api(nplanes=64, npix=4096, nz=8, one_by_one=2):
    """The stack THROUGH THE REFERENCE-SHAPED API: SimulationCollection.sum_raytracing_snapshots over `nplanes` host arrays
    (the .npy branch of simcoll.py:267-336; the loader hands over arrays already in host memory, so disk is not timed) for
    `nz` source redshifts.  As a sequence (one call: every plane uploaded once, resident in HBM, one stacking pass per source
    redshift) and, for `one_by_one` of them, as the reference's loop does it (one call per source redshift: every call
    uploads all planes again).  maps/s = maps handed back as numpy arrays per second of wall time."""
    import pandas as pd
    from .simcoll import SimulationCollection
    host = [p.cpu().numpy() for p in synth_kappa_planes(nplanes, npix)]
    torch.cuda.empty_cache()
    idx = pd.MultiIndex.from_tuples([(1, r + 1) for r in range(nplanes)], names=["box_nr", "snap_nr"])
    table = pd.DataFrame({"redshift": [(r + 0.5) * (1.0 / nplanes) for r in range(nplanes)]}, index=idx)

    class Flat:
        def comoving_distance(self, z):
            return 3000.0 * z

    class MemCollection(SimulationCollection):
        def _load_ray_map(self, ray_file):
            return host[int(ray_file) - 1]

    import types
    sims = {"box1": types.SimpleNamespace(dirs={"sim": ""}, file_dsc={"root": "kappa", "extension": "npy"})}
    import glob as _glob

    def make():
        sc = MemCollection(table.copy(), sims, cosmology=Flat())
        return sc
    # (the collection finds its files with glob: answered from memory here)
    real_glob = _glob.glob
    from . import simcoll as _sc
    _sc.glob.glob = lambda pat: [pat.split("*")[1].split(".")[0]]
    try:
        zs = [0.55 + 0.05 * i for i in range(nz)]
        rng = {"z": [], "box": [0], "ray": [0]}
        make().sum_raytracing_snapshots(None, ["kappa_2"], ["kappa_2"], rng, z_src=1.1, z_src_shift=zs[:2], reweight=True)    # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        maps = make().sum_raytracing_snapshots(None, ["kappa_2"], ["kappa_2"], rng, z_src=1.1, z_src_shift=zs, reweight=True)
        torch.cuda.synchronize()
        t_many = time.perf_counter() - t0
        assert len(maps) == nz and all(m.shape == (npix, npix) for m in maps)
        t0 = time.perf_counter()
        for z in zs[:one_by_one]:
            one = make().sum_raytracing_snapshots(None, ["kappa_2"], ["kappa_2"], rng, z_src=1.1, z_src_shift=z, reweight=True)
        torch.cuda.synchronize()
        t_one = (time.perf_counter() - t0) / max(1, one_by_one)
        same = bool(np.array_equal(one, maps[one_by_one - 1])) if one_by_one else None
    finally:
        _sc.glob.glob = real_glob
    return {"metric": f"stacked kappa maps/s through SimulationCollection.sum_raytracing_snapshots: {nplanes} host planes x {npix}^2 fp64, "
                      f"{nz} source redshifts", "value": nz / t_many, "unit": "maps/s",
            "ms_total": round(t_many * 1e3, 2), "ms_per_map": round(t_many / nz * 1e3, 2),
            "one_call_per_source_redshift": {"ms_per_map": round(t_one * 1e3, 2), "maps_per_s": round(1.0 / t_one, 2),
                                             "note": "the reference's loop shape: every call uploads all planes again"},
            "bit_identical_to_single_calls": same,
            "h2d_GB_once": round(nplanes * npix * npix * 8 / 1e9, 2),
            "note": "z_src_shift as a sequence: planes uploaded once and resident in HBM, one pass of ast_kappa_stack per "
                    "source redshift, results copied back as numpy arrays; the loader returns arrays already in host memory"}



def bench_skyarray_chain(npix=4096, reps=3):
    """One map through the reference-shaped per-map API (rays/skys/sky_array.py): SkyArray.from_array(host map) ->
    filter(Gaussian, 1 arcmin) -> convert_convergence_to_deflection(rtn=False) -> pdf -> wl_peak_counts -> the two deflection
    maps read back as arrays.  Timed as the methods leave it (what they produce stays in HBM, rays/_resident.MapStore) and
    with every intermediate map fetched between the methods (the reference's shape: numpy arrays in ``self.data``)."""
    from .rays.skys import SkyArray
    host = synth_kappa_planes(1, npix)[0].cpu().numpy()

    def chain(fetch):
        sky = SkyArray.from_array(host, opening_angle=20.0, quantity="kappa_2", dir_in="")
        sky.filter({"gaussian": {"theta_i": 1.0, "abbrev": "g"}}, on="orig")
        if fetch:
            sky.data["orig_g"]
        sky.convert_convergence_to_deflection(on="orig_g", rtn=False)
        if fetch:
            sky.data["defltx"], sky.data["deflty"]
        pdf = sky.pdf(100, of="orig_g")
        peaks = sky.wl_peak_counts(30, "", of="orig_g")
        return sky.data["defltx"], sky.data["deflty"], pdf, peaks

    out = {}
    for name, fetch in (("resident", False), ("fetched_between_methods", True)):
        chain(fetch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = chain(fetch)
        torch.cuda.synchronize()
        out[name] = {"ms_per_map": round((time.perf_counter() - t0) / reps * 1e3, 2)}
        out[name + "_result"] = res
    same = all(np.array_equal(a, b) for a, b in zip(out.pop("resident_result")[:2], out.pop("fetched_between_methods_result")[:2]))
    return {"metric": f"one {npix}^2 float64 map through SkyArray: filter -> kappa->alpha -> pdf -> peak counts, host array in, deflection maps out",
            "value": round(1e3 / out["resident"]["ms_per_map"], 2), "unit": "maps/s", **out, "bit_identical": bool(same),
            "pcie_MB": {"resident": round(3 * npix * npix * 8 / 1e6), "fetched_between_methods": round(8 * npix * npix * 8 / 1e6)}}
