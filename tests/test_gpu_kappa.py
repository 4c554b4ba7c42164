"""GPU parity for the kappa-map path through the
C-ABI, against the oracle and the reference's own known-answer values.

Tolerances: plane stack and unit conversion fp64 bit-exact (sequential IEEE ops);
histogram counts bit-exact; FFT-based results 1e-6 relative to the map's peak
(north_star tolerance), in practice ~1e-13.
"""
import ctypes as ct
import json
import os

import numpy as np
import numpy.testing as npt
import pytest

from oracle import kappa as ok

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


@pytest.fixture(scope="module")
def lens(hip):
    from astrild_amd import lensing
    torch.cuda.set_device(0)
    return lensing


@pytest.fixture(scope="module")
def dev(hip):
    from astrild_amd import device
    return device


def test_stack_unweighted_bit_exact(lens, dev):
    rng = np.random.default_rng(0)
    planes = [rng.standard_normal((96, 96)) * 10.0 ** rng.integers(-3, 3) for _ in range(17)]
    got = lens.kappa_stack([dev.as_device(p) for p in planes]).cpu().numpy()
    assert np.array_equal(got, ok.kappa_stack(planes))


def test_stack_weighted_bit_exact_with_source_clamp(lens, dev):
    rng = np.random.default_rng(1)
    P = 12
    planes = [rng.standard_normal((64, 64)) for _ in range(P)]
    x_near = np.arange(P) * 83.3
    x_far = x_near + 83.3
    x_src, x_shift = 1100.0, 700.0          # planes beyond 700 get the x_far clamp
    wn, wd = lens.translate_redshift_weights(x_near, x_far, x_src, x_shift)
    got = lens.kappa_stack([dev.as_device(p) for p in planes], wn, wd).cpu().numpy()
    ref = ok.kappa_stack(planes, x_near, x_far, x_src, x_shift)
    assert np.array_equal(got, ref)


def test_stack_fp32_and_single_plane(lens, dev):
    rng = np.random.default_rng(2)
    planes = [rng.standard_normal((32, 32)).astype(np.float32) for _ in range(5)]
    got = lens.kappa_stack([dev.as_device(p) for p in planes]).cpu().numpy()
    ref = planes[0].copy()
    for p in planes[1:]:
        ref = ref + p
    assert np.array_equal(got, ref)
    one = lens.kappa_stack([dev.as_device(planes[0])]).cpu().numpy()
    assert np.array_equal(one, planes[0])


def test_unit_conversion_bit_exact_and_reference_values(lens, dev):
    g = GOLD["unit_conversion"]
    for case in g["cases"]:
        t = dev.as_device(np.full(10, g["c_light_km_s"] ** case["power"]))
        assert lens.convert_code_to_phy_units(case["quantity"], t).cpu().numpy()[0] == g["expected"]
    rng = np.random.default_rng(3)
    v = rng.standard_normal(1000) * 1e9
    got = lens.convert_code_to_phy_units("kappa_2", dev.as_device(v.copy())).cpu().numpy()
    assert np.array_equal(got, ok.convert_code_to_phy_units("kappa_2", v))


def test_kappa_to_alphas_reference_known_answer_via_host_abi(hip):
    # the libglsg.so-compatible entry point, bound exactly like sky_utils.py:402-419
    g = GOLD["kappa_to_alphas"]
    gg = ok.general_gaussian(100, 1, 10)
    kappa = np.array(np.outer(gg, gg), dtype=ct.c_double)
    npix = g["npix"]
    a1 = np.zeros((npix, npix), dtype=ct.c_double)
    a2 = np.zeros((npix, npix), dtype=ct.c_double)
    # (a handle of its own, like a reference user's ct.CDLL: the package's shared handle keeps its own argtypes)
    own = ct.CDLL(hip._name)
    fn = own.kappa0_to_alphas
    fn.restype = ct.c_void_p
    fn.argtypes = [np.ctypeslib.ndpointer(dtype=ct.c_double), ct.c_int, ct.c_double,
                   np.ctypeslib.ndpointer(dtype=ct.c_double), np.ctypeslib.ndpointer(dtype=ct.c_double)]
    fn(kappa, npix, np.deg2rad(g["opening_angle_deg"]), a1, a2)
    e = g["alpha_1"]
    npt.assert_almost_equal(a1.min(), e["min"], decimal=e["decimal"])
    npt.assert_almost_equal(a1.mean(), e["mean"], decimal=e["decimal"])
    npt.assert_almost_equal(a1.max(), e["max"], decimal=e["decimal"])
    r1, r2 = ok.kappa0_to_alphas(kappa, npix, np.deg2rad(g["opening_angle_deg"]))
    npt.assert_allclose(a1, r1, rtol=0, atol=1e-6 * abs(r1).max())
    npt.assert_allclose(a2, r2, rtol=0, atol=1e-6 * abs(r2).max())
    # phi through the host ABI too
    phi = np.zeros((npix, npix), dtype=ct.c_double)
    fp = own.kappa0_to_phi
    fp.restype = ct.c_void_p
    fp.argtypes = [np.ctypeslib.ndpointer(dtype=ct.c_double), ct.c_int, ct.c_double,
                   np.ctypeslib.ndpointer(dtype=ct.c_double)]
    fp(kappa, npix, np.deg2rad(g["opening_angle_deg"]), phi)
    rp = ok.kappa0_to_phi(kappa, npix, np.deg2rad(g["opening_angle_deg"]))
    npt.assert_allclose(phi, rp, rtol=0, atol=1e-6 * abs(rp).max())


@pytest.mark.parametrize("nc", [16, 100, 256])
def test_lens_plan_device_variants_vs_oracle(lens, dev, nc):
    rng = np.random.default_rng(nc)
    kappa = rng.standard_normal((nc, nc)) * 0.01
    bsz = np.deg2rad(3.0)
    plan = lens.LensPlan(nc, bsz)
    kd = dev.as_device(kappa)
    a1, a2 = plan.alphas(kd)
    phi = plan.phi(kd)
    a1b, _ = plan.alphas(kd)                       # cached kernel spectra: second call identical
    r1, r2 = ok.kappa0_to_alphas(kappa, nc, bsz)
    rp = ok.kappa0_to_phi(kappa, nc, bsz)
    npt.assert_allclose(a1.cpu().numpy(), r1, rtol=0, atol=1e-10 * abs(r1).max())
    npt.assert_allclose(a2.cpu().numpy(), r2, rtol=0, atol=1e-10 * abs(r2).max())
    npt.assert_allclose(phi.cpu().numpy(), rp, rtol=0, atol=1e-10 * abs(rp).max())
    assert torch.equal(a1, a1b)


def test_gaussian_real_space_reference_known_answer(lens, dev):
    g = GOLD["nfw_halo"]
    halo = {k: np.array(v) for k, v in g["halo"].items()}
    dt = ok.analytic_halo_signal_map(halo, g["extent"], g["direction"], g["suppress"], g["suppression_R"],
                                     g["npix"], "dT")
    gs = GOLD["gaussian_smoothing"]
    plan = lens.SmoothPlan(g["npix"])
    for case in gs["cases"]:
        sigma_px = ok.fwhm_to_sigma(case["fwhm_arcmin"]) / 60.0 * g["npix"] / gs["theta_deg"]
        img = dev.as_device(dt.copy())
        plan.gaussian(img, sigma_px, "gaussian")
        got = img.cpu().numpy()
        npt.assert_almost_equal(got.max() * 1e8, case["max_times_1e8"], decimal=case["decimal"])
        ref = ok.gaussian_smooth(dt, gs["theta_deg"], ok.fwhm_to_sigma(case["fwhm_arcmin"]), kind="gaussian")
        npt.assert_allclose(got, ref, rtol=0, atol=1e-12 * abs(ref).max())


@pytest.mark.parametrize("npix", [64, 501, 512])
def test_gaussian_fft_branch_vs_oracle(lens, dev, npix):
    rng = np.random.default_rng(npix)
    img = rng.standard_normal((npix, npix))
    theta, sig = 5.0, 1.7
    sigma_px = sig / 60.0 * npix / theta
    t = dev.as_device(img.copy())
    lens.SmoothPlan(npix).gaussian(t, sigma_px, "gaussianFFT")
    ref = ok.gaussian_smooth(img, theta, sig, kind="gaussianFFT")
    npt.assert_allclose(t.cpu().numpy(), ref, rtol=0, atol=1e-12 * abs(ref).max())


def test_gaussian_real_space_small_map_large_kernel(lens, dev):
    # kernel radius larger than the map: exercises repeated reflection
    rng = np.random.default_rng(7)
    img = rng.standard_normal((24, 24))
    t = dev.as_device(img.copy())
    lens.SmoothPlan(24).gaussian(t, 9.0, "gaussian")
    from scipy import ndimage
    npt.assert_allclose(t.cpu().numpy(), ndimage.gaussian_filter(img, 9.0), rtol=0, atol=1e-13)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_histogram_counts_bit_exact_and_pdf(lens, dev, dtype):
    rng = np.random.default_rng(5)
    img = (rng.standard_normal((300, 300)) * 0.02).astype(dtype)
    t = dev.as_device(img)
    lo, hi = lens.minmax(t)
    assert lo == img.min() and hi == img.max()
    for nbins in (1, 7, 100, 1000):
        counts, edges = lens.histogram(t, nbins)
        rc, re = np.histogram(img.astype(np.float64), bins=nbins)
        assert np.array_equal(counts, rc)
        npt.assert_allclose(edges, re, rtol=1e-15, atol=0)
    vals, _ = lens.histogram(t, 50, density=True)
    rv, _ = ok.pdf(img.astype(np.float64), 50)
    npt.assert_allclose(vals, rv, rtol=1e-13)
    c2, _ = lens.histogram(t, 10, range=(-0.01, 0.03))
    assert np.array_equal(c2, np.histogram(img.astype(np.float64), bins=10, range=(-0.01, 0.03))[0])


def test_histogram_of_a_constant_map_and_ragged_minmax(lens, dev):
    """range=None on a constant map: numpy widens the degenerate range by +-0.5 (done on the device here); min/max over
    a length that is no multiple of the unrolled trip."""
    img = np.full(1000, 0.25)
    counts, edges = lens.histogram(dev.as_device(img), 8)
    rc, re = np.histogram(img, bins=8)
    assert np.array_equal(counts, rc)
    npt.assert_allclose(edges, re, rtol=1e-15)
    rng = np.random.default_rng(11)
    for n in (1, 63, 4 * 2048 * 256 + 5, 5_000_003):
        v = rng.standard_normal(n)
        lo, hi = lens.minmax(dev.as_device(v))
        assert lo == v.min() and hi == v.max()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_histogram_wide_loads_replicas_and_unaligned_views(lens, dev, dtype):
    """The counting kernel's paths: 16-byte loads with a ragged tail, an unaligned view (scalar loads), 1 .. 32 LDS
    replicas of the bins (nbins 4096 .. 64), more workgroups than the grid cap - counts equal numpy's, bin for bin."""
    rng = np.random.default_rng(23)
    v = (rng.standard_normal(3_000_001) * 0.02).astype(dtype)
    v[::7] = np.round(v[::7], 2)                              # many values on or near edges
    t = dev.as_device(v)
    for view, ref in ((t, v), (t[1:], v[1:]), (t[: 4 * 1024 * 512 + 3], v[: 4 * 1024 * 512 + 3])):
        for nbins in (64, 100, 2048, 4096):
            counts, edges = lens.histogram(view, nbins)
            rc, re = np.histogram(ref.astype(np.float64), bins=nbins)
            assert np.array_equal(counts, rc), (nbins, ref.size)
            npt.assert_allclose(edges, re, rtol=1e-15, atol=0)
        c2, _ = lens.histogram(view, 33, range=(-0.01, 0.015))
        assert np.array_equal(c2, np.histogram(ref.astype(np.float64), bins=33, range=(-0.01, 0.015))[0])


def test_histogram_values_on_bin_edges(lens, dev):
    # values exactly on edges: left-closed bins, right-most bin closed
    img = np.array([0.0, 0.1, 0.2, 0.3, 0.5, 0.7, 1.0, 1.0, 0.9999999999999999, 0.30000000000000004])
    counts, _ = lens.histogram(dev.as_device(img), 10)
    assert np.array_equal(counts, np.histogram(img, bins=10)[0])


def test_add_galaxy_shape_noise(lens, dev):
    kap = np.random.default_rng(1).standard_normal((64, 64)) * 0.01
    gsn = ok.galaxy_shape_noise(64, 34077)
    got = lens.add(dev.as_device(kap), dev.as_device(gsn)).cpu().numpy()
    assert np.array_equal(got, kap + gsn)


@pytest.mark.parametrize("world,nplanes,npix", [(2, 9, 512), (4, 64, 4096)])
def test_sharded_stack_ranks_share_one_gpu(hip, tmp_path, world, nplanes, npix):
    _sharded_stack_ranks_share_one_gpu(tmp_path, world, nplanes, npix, False)


@pytest.mark.parametrize("world,nplanes,npix", [(3, 10, 512), (4, 16, 2048)])
def test_map_stream_ranks_share_one_gpu(hip, tmp_path, world, nplanes, npix):
    """kappa_shard.MapStream with the real ops: 2 P + 1 maps, map m reduced onto rank m mod P, which smooths it, takes
    its PDF and its deflection field on a second stream; against the single-GPU pipeline on the sequential stack."""
    _sharded_stack_ranks_share_one_gpu(tmp_path, world, nplanes, npix, True)


def test_resident_planes_several_maps_sharded(hip, tmp_path):
    """The sharded form of `z_src_shift` as a sequence (PlaneStacker._stack_many under a process group): 3 ranks on one GPU,
    5 maps over the same resident planes, map m on rank m mod 3; each against the single-GPU stack with its weights."""
    import torch
    from astrild_amd import lensing
    world, nplanes, npix = 3, 10, 512
    _sharded_stack_ranks_share_one_gpu(tmp_path, world, nplanes, npix, "api")
    out = str(tmp_path / "stack.npy")
    planes = lensing.synth_kappa_planes(nplanes, npix)
    wnum, wden = lensing.synth_plane_weights(nplanes)
    res = [np.load(out + f".api{r}.npz") for r in range(world)]
    for m in range(world + 2):
        ref = lensing.kappa_stack(planes, None if m == 1 else wnum * (1.0 + 0.5 * m), None if m == 1 else wden).cpu().numpy()
        got = res[m % world][f"map{m}"].reshape(npix, npix)
        npt.assert_allclose(got, ref, rtol=0, atol=4e-15 * np.abs(ref).max())


def _sharded_stack_ranks_share_one_gpu(tmp_path, world, nplanes, npix, stream):
    """Config D rehearsal: `world` processes on cuda:0 over gloo, plane p on rank p mod P, HipStackOps + the
    all-to-all chunk exchange of kappa_shard, against the single-GPU sequential stack (64 planes x 4096^2 fp64
    at the stated size).  Re-association only: <= P ulp of the sum of |planes|."""
    import socket
    import subprocess
    import sys
    import torch
    from astrild_amd import lensing
    torch.cuda.set_device(0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "stack.npy")
    worker = os.path.join(os.path.dirname(__file__), "kappa_gpu_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(nplanes), str(npix), out] +
                              (["api"] if stream == "api" else ["stream"] if stream else [])) for r in range(world)]
    assert [p.wait(timeout=600) for p in procs] == [0] * world
    got = np.load(out).reshape(npix, npix)
    planes = lensing.synth_kappa_planes(nplanes, npix)
    wnum, wden = lensing.synth_plane_weights(nplanes)
    seq = lensing.kappa_stack(planes, wnum, wden).cpu().numpy()
    bound = sum(abs(float(wnum[p] / wden[p])) * float(planes[p].abs().max()) for p in range(nplanes))
    npt.assert_allclose(got, seq, rtol=0, atol=world * 2.0 ** -52 * bound)
    assert np.abs(got - seq).max() <= 1e-15 * np.abs(seq).max() * 4
    if not stream or stream == "api":
        return
    lp, sp = lensing.lens_plan(npix, np.deg2rad(20.0)), lensing.smooth_plan(npix)
    res = [np.load(out + f".stream{r}.npz") for r in range(world)]
    for m in range(2 * world + 1):
        r = m % world
        assert f"map{m}" in res[r].files and all(f"map{m}" not in res[q].files for q in range(world) if q != r)
        ref = lensing.kappa_stack(planes, wnum * (1.0 + 0.25 * m), wden)
        lensing.convert_code_to_phy_units("kappa_2", ref)
        sp.gaussian(ref, 3.4, "gaussianFFT")
        counts, _ = lensing.histogram(ref, 100, density=True)
        a1, _ = lp.alphas(ref)
        scale = float(ref.abs().max())
        npt.assert_allclose(res[r][f"map{m}"], ref.cpu().numpy(), rtol=0, atol=1e-13 * scale)
        npt.assert_allclose(res[r][f"a1_{m}"], a1.cpu().numpy(), rtol=0, atol=1e-12 * float(a1.abs().max()))
        # the PDF of a map that differs by 1e-16 may move a pixel across a bin edge
        assert np.abs(res[r][f"pdf{m}"] - counts).sum() <= 4e-6 * counts.sum()


@pytest.mark.parametrize("npix,conv,limits", [(64, "", None), (257, "normalize", None), (512, "", (-0.5, 1.5))])
def test_wl_peak_counts_bit_exact(lens, dev, npix, conv, limits):
    """SkyArray.wl_peak_counts: peak set, percentile bounds and histogram counts identical to the oracle."""
    import torch
    from astrild_amd.rays import SkyMap
    rng = np.random.default_rng(npix)
    img = ok.gaussian_smooth(rng.standard_normal((npix, npix)), 10.0, 2.0, kind="gaussianFFT")
    img[5, 7] = img[5, 8] = img.max() + 1.0                       # a plateau: not a peak
    t = dev.as_device(img)
    for q in (5, 95, 50, 0, 100, 33.3):
        assert lens.percentile(t, [q])[0] == np.percentile(img, q)
    ks = [0, 1, npix * npix // 3, npix * npix - 1]
    assert lens.order_statistics(t, ks) == np.sort(img.ravel())[ks].tolist()
    assert lens.order_statistics(dev.as_device(img.astype(np.float32)), ks) == \
        np.sort(img.astype(np.float32).ravel())[ks].astype(np.float64).tolist()
    vals, idx = lens.peak_find(t)
    rv, rp = ok.locate_peaks(img, np.array([-np.inf, np.inf]))
    assert np.array_equal(vals, rv) and np.array_equal(idx, rp[:, 0] * npix + rp[:, 1])
    sky = SkyMap.from_array(img, npix, 10.0, "kappa_2", "/tmp/")
    df = sky.wl_peak_counts(12, conv, limits=limits)
    centres, counts = ok.wl_peak_counts(img, 12, conv, limits)
    assert np.array_equal(df["counts"].values, counts)            # integer counts: exact
    assert np.array_equal(df["kappa"].values, centres)


def test_flat_sky_power_and_bispectrum_vs_oracle(lens, dev):
    """AngularPowerSpectrum.from_array and Bispectrum2D.from_skymap (lenstools restated) -
    annulus membership and pixel counts exact, P_l to 1e-12; the FFT-estimator bispectrum against the
    brute-force triangle enumeration, triangle counts exact."""
    from astrild_amd.rays import SkyMap
    from astrild_amd.power_spectra import AngularPowerSpectrum
    from astrild_amd.bispectra import Bispectrum2D
    rng = np.random.default_rng(12)
    npix, theta = 256, 10.0
    img = ok.gaussian_smooth(rng.standard_normal((npix, npix)), theta, 6.0, kind="gaussianFFT")
    img = img + 0.5 * img ** 2
    sky = SkyMap.from_array(img, npix, theta, "kappa_2", "/tmp/")
    lf = 2 * np.pi / np.deg2rad(theta)
    edges = np.concatenate([lf * np.arange(0.5, 40.0, 3.0), [lf * 190.0, lf * 400.0]])   # incl. integer-radius edges
    edges[1] = lf * 3.0                                                                  # an edge ON a lattice radius
    aps = AngularPowerSpectrum.from_array(sky, "orig", edges)
    l, p = ok.flat_power_spectrum(img, theta, edges)
    assert np.array_equal(aps.ell, l)
    npt.assert_allclose(aps.P, p, rtol=1e-12, atol=1e-14 * p.max())
    assert aps.P[-1] == 0.0                                                              # beyond the corner: empty
    # cross power of two maps
    img2 = np.roll(img, 3, axis=0)
    l2, p2 = lens.flat_power_spectrum(img, theta, edges, img2=img2)
    npt.assert_allclose(p2, ok.flat_power_spectrum(img, theta, edges, img2=img2)[1], rtol=1e-11, atol=1e-13 * p.max())
    # bispectrum on a small map, brute force
    n2 = 32
    small = img[:n2, :n2].copy()
    sky2 = SkyMap.from_array(small, n2, 2.0, "kappa_2", "/tmp/")
    lf2 = 2 * np.pi / np.deg2rad(2.0)
    e2 = lf2 * np.array([1.5, 3.5, 6.0, 9.2, 14.0])
    bs = Bispectrum2D.from_skymap(sky2, "orig", e2)
    lb, bb, nb = ok.flat_bispectrum_equilateral_brute(small, 2.0, e2)
    assert np.array_equal(bs.ntri, nb)
    npt.assert_allclose(bs.B, bb, rtol=1e-9, atol=1e-12 * np.abs(bb).max())


_LENS_SPLIT = {16384: (128, 128), 8192: (64, 128), 4096: (64, 64), 2048: (32, 64), 1024: (32, 32), 512: (16, 32), 256: (16, 16)}


@pytest.mark.parametrize("length", [256, 512, 1024, 2048, 8192, 16384])
def test_lens_column_transforms_against_numpy(hip, dev, length):
    """The two-pass column transforms of the padded lens convolution (lens_fft.hip): forward = np.fft.fft along axis 0
    with frequency k1 + N1 k2 at row N2 k1 + k2; rows >= nonzero_rows are never read (NaN there); the inverse undoes it
    (times `mul`, unnormalised) and writes only the rows that are kept."""
    from astrild_amd import _lib
    n1, n2 = _LENS_SPLIT[length]
    pitch, ncols = 40, 37                                  # ragged: 37 of 40 columns, not a multiple of the 16-column tile
    rng = np.random.default_rng(length)
    x = rng.standard_normal((length, pitch)) + 1j * rng.standard_normal((length, pitch))
    perm = np.array([(r // n2) + n1 * (r % n2) for r in range(length)])      # row r holds frequency perm[r]
    ref = np.fft.fft(x[:, :ncols], axis=0)
    d = dev.as_device(x.copy())
    _lib.check(hip.ast_lens_cols_forward(dev.ptr(d), length, pitch, ncols, length, dev.stream()))
    got = d.cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(got[:, :ncols] - ref[perm]).max() < 1e-13 * scale
    assert np.array_equal(got[:, ncols:], x[:, ncols:])                      # columns past ncols untouched
    # zero-padded input: the lower half holds NaN and is never read
    xz = x.copy()
    xz[length // 2:] = 0.0
    refz = np.fft.fft(xz[:, :ncols], axis=0)
    xn = x.copy()
    xn[length // 2:] = np.nan
    dz = dev.as_device(xn)
    _lib.check(hip.ast_lens_cols_forward(dev.ptr(dz), length, pitch, ncols, length // 2, dev.stream()))
    gotz = dz.cpu().numpy()
    assert np.isfinite(gotz[:, :ncols]).all()
    assert np.abs(gotz[:, :ncols] - refz[perm]).max() < 1e-13 * scale
    # inverse of spec * mul, all rows
    m = rng.standard_normal((length, pitch)) + 1j * rng.standard_normal((length, pitch))
    spec_nat = ref * 1.0
    mul_nat = m[:, :ncols]
    want = np.fft.ifft(spec_nat * mul_nat, axis=0) * length
    spec_perm = np.zeros((length, pitch), dtype=np.complex128)
    mul_perm = np.zeros((length, pitch), dtype=np.complex128)
    spec_perm[:, :ncols] = spec_nat[perm]
    mul_perm[:, :ncols] = mul_nat[perm]
    sd, md = dev.as_device(spec_perm), dev.as_device(mul_perm)
    out = dev.as_device(np.full((length, pitch), np.nan + 0j))
    _lib.check(hip.ast_lens_cols_inverse(dev.ptr(sd), dev.ptr(md), dev.ptr(out), length, pitch, ncols, length, dev.stream()))
    o = out.cpu().numpy()
    assert np.abs(o[:, :ncols] - want).max() < 1e-13 * np.abs(want).max()
    assert np.array_equal(sd.cpu().numpy(), spec_perm)                       # inputs intact
    # only the upper half kept
    out2 = dev.as_device(np.full((length, pitch), np.nan + 0j))
    _lib.check(hip.ast_lens_cols_inverse(dev.ptr(sd), dev.ptr(md), dev.ptr(out2), length, pitch, ncols, length // 2, dev.stream()))
    o2 = out2.cpu().numpy()
    assert np.abs(o2[:length // 2, :ncols] - want[:length // 2]).max() < 1e-13 * np.abs(want).max()
    # in place without a multiplier: round trip
    _lib.check(hip.ast_lens_cols_inverse(dev.ptr(d), None, dev.ptr(d), length, pitch, ncols, length, dev.stream()))
    back = d.cpu().numpy()
    assert np.abs(back[:, :ncols] / length - x[:, :ncols]).max() < 1e-13 * np.abs(x).max()
    assert hip.ast_lens_cols_forward(dev.ptr(d), 3000, pitch, ncols, 3000, dev.stream()) < 0
    assert hip.ast_lens_cols_supported(8192) == 1 and hip.ast_lens_cols_supported(100) == 0


@pytest.mark.parametrize("length,nmul", [(256, 2), (512, 1), (1024, 2), (2048, 2), (4096, 1), (8192, 2), (16384, 2)])
def test_lens_column_convolution_in_three_passes(hip, dev, length, nmul):
    """ast_lens_cols_convolve: out_m = ifft(fft(x) * mul_m) along axis 0 (unnormalised, the rows that are kept), with the
    spectrum of x never stored - against numpy, and against the forward + inverse calls it replaces."""
    import ctypes as ct
    from astrild_amd import _lib
    n1, n2 = _LENS_SPLIT[length]
    pitch, ncols = 40, 37
    rng = np.random.default_rng(7 * length + nmul)
    x = rng.standard_normal((length, pitch)) + 1j * rng.standard_normal((length, pitch))
    perm = np.array([(r // n2) + n1 * (r % n2) for r in range(length)])
    muls_nat = [rng.standard_normal((length, ncols)) + 1j * rng.standard_normal((length, ncols)) for _ in range(nmul)]
    mds = []
    for m in muls_nat:
        mp = np.zeros((length, pitch), dtype=np.complex128)
        mp[:, :ncols] = m[perm]
        mds.append(dev.as_device(mp))
    for nonzero, keep in [(length, length), (length // 2, length // 2)]:
        xz = x.copy()
        xz[nonzero:] = 0.0
        spec = np.fft.fft(xz[:, :ncols], axis=0)
        xin = x.copy()
        xin[nonzero:] = np.nan                               # never read
        d = dev.as_device(xin)
        outs = [dev.as_device(np.full((length, pitch), np.nan + 0j)) for _ in range(nmul)]
        marr = (ct.c_void_p * nmul)(*[dev.ptr(t) for t in mds])
        oarr = (ct.c_void_p * nmul)(*[dev.ptr(t) for t in outs])
        _lib.check(hip.ast_lens_cols_convolve(dev.ptr(d), length, pitch, ncols, nonzero, marr, oarr, nmul, keep, dev.stream()))
        # the calls it replaces, on the same input
        d2 = dev.as_device(xin)
        _lib.check(hip.ast_lens_cols_forward(dev.ptr(d2), length, pitch, ncols, nonzero, dev.stream()))
        for m, md, out in zip(muls_nat, mds, outs):
            want = np.fft.ifft(spec * m, axis=0) * length
            o = out.cpu().numpy()
            assert np.abs(o[:keep, :ncols] - want[:keep]).max() < 2e-13 * np.abs(want).max()
            assert np.isnan(o[:, ncols:]).all()              # columns past ncols untouched
            old = dev.as_device(np.full((length, pitch), np.nan + 0j))
            _lib.check(hip.ast_lens_cols_inverse(dev.ptr(d2), dev.ptr(md), dev.ptr(old), length, pitch, ncols, keep, dev.stream()))
            assert np.abs(o[:keep, :ncols] - old.cpu().numpy()[:keep, :ncols]).max() < 2e-13 * np.abs(want).max()
    bad = (ct.c_void_p * 1)(dev.ptr(d))
    assert hip.ast_lens_cols_convolve(dev.ptr(d), length, pitch, ncols, length, marr, bad, 1, length, dev.stream()) < 0   # out == data
    assert hip.ast_lens_cols_convolve(dev.ptr(d), length, pitch, ncols, length, marr, oarr, 3, length, dev.stream()) < 0


@pytest.mark.parametrize("npix,sigma_px", [(1000, 2.5), (777, 4.3), (2048, 7.5), (64, 3.0), (1500, 18.8), (256, 12.0)])
def test_gaussian_fft_smoothing_real_space_route_equals_fft_route(lens, dev, npix, sigma_px, monkeypatch):
    """For 2.5 <= sigma_px <= 18.8 "gaussianFFT" runs as two periodic real-space passes; the FFT route (forced through
    AST_SMOOTH_FFT) gives the same map to 1e-13 of its peak, on ragged sizes too."""
    rng = np.random.default_rng(npix)
    img = rng.standard_normal((npix, npix))
    plan = lens.SmoothPlan(npix)
    a = dev.as_device(img.copy())
    plan.gaussian(a, sigma_px, "gaussianFFT")
    monkeypatch.setenv("AST_SMOOTH_FFT", "1")
    b = dev.as_device(img.copy())
    plan.gaussian(b, sigma_px, "gaussianFFT")
    monkeypatch.delenv("AST_SMOOTH_FFT")
    a, b = a.cpu().numpy(), b.cpu().numpy()
    assert not np.array_equal(a, b)                               # two different routes did run
    npt.assert_allclose(a, b, rtol=0, atol=2e-13 * np.abs(b).max())
    npt.assert_allclose(a.mean(), img.mean(), rtol=0, atol=1e-13)  # the periodic kernel sums to one


@pytest.mark.parametrize("nc", [128, 256, 512, 1024, 2048, 4096, 8192])
def test_lens_row_transforms_against_numpy(hip, dev, nc):
    """Hand-written row transforms of the padded convolution: forward = np.fft.rfft of (row, nc zeros); inverse =
    scale * first nc samples of the unnormalised irfft."""
    from astrild_amd import _lib
    rng = np.random.default_rng(nc)
    rows = 6
    nh = nc + 1
    kap = rng.standard_normal((nc, nc))
    spec = dev.as_device(np.full((nc, nh), np.nan + 0j))
    kd = dev.as_device(kap)
    _lib.check(hip.ast_lens_rows_forward(dev.ptr(kd), nc, dev.ptr(spec), nh, dev.stream()))
    got = spec.cpu().numpy()
    sel = [0, 1, 2, nc // 2, nc - 2, nc - 1][:rows]
    ref = np.fft.rfft(np.concatenate([kap[sel], np.zeros((len(sel), nc))], axis=1), axis=1)
    assert np.abs(got[sel] - ref).max() < 1e-12 * np.abs(ref).max()
    assert np.isfinite(got).all()
    # inverse on an arbitrary Hermitian-consistent spectrum: the spectrum of real rows of length 2 nc
    full = rng.standard_normal((nc, 2 * nc))
    sp = np.fft.rfft(full, axis=1)
    sd = dev.as_device(sp)
    out = dev.as_device(np.full((nc, nc), np.nan))
    scale = 0.37
    _lib.check(hip.ast_lens_rows_inverse(dev.ptr(sd), nh, nc, scale, dev.ptr(out), dev.stream()))
    o = out.cpu().numpy()
    want = scale * 2 * nc * full[:, :nc]                   # unnormalised C2R = length * irfft
    assert np.abs(o - want).max() < 1e-12 * np.abs(want).max()
    assert hip.ast_lens_rows_supported(4096) == 1 and hip.ast_lens_rows_supported(8192) == 1 and hip.ast_lens_rows_supported(100) == 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_minmax_wide_loads_tails_and_unaligned_views(dtype):
    """ast_minmax (np.histogram's range=None pass; the paint's mass bound): 16-byte loads, one pair of atomics per
    workgroup - against numpy for lengths that are not multiples of the vector, tiny inputs and a view that starts off a
    16-byte boundary (scalar loads)."""
    from astrild_amd import device as dev, lensing
    rng = np.random.default_rng(12)
    for n in (1, 3, 255, 4097, 1_000_003, 4096 * 4096):
        a = rng.standard_normal(n).astype(dtype)
        a[rng.integers(n)] = -77.5
        a[rng.integers(n)] = 99.25 if n > 1 else a[0]
        t = dev.as_device(a)
        lo, hi = lensing.minmax(t)
        assert (lo, hi) == (float(a.min()), float(a.max()))
        if n > 4:
            lo, hi = lensing.minmax(t[1:])
            assert (lo, hi) == (float(a[1:].min()), float(a[1:].max()))


def test_lens_plan_at_the_reference_default_npix_8192(lens, dev, monkeypatch):
    """nc = 8192 is the reference's halo-map default (sky_array.py:266,348): the hand-written row (32 x 16 x 16 points) and
    column (128 x 128) transforms carry the plan there too; against the rocFFT 2-D route on the same map, and no rocFFT
    kernel runs in the hand-written one (the profile sites are the lens.* ones only)."""
    nc, bsz = 8192, np.deg2rad(10.0)
    g = torch.Generator(device="cuda").manual_seed(8192)
    kd = torch.randn((nc, nc), generator=g, device="cuda", dtype=torch.float64) * 0.01
    plan = lens.LensPlan(nc, bsz)
    dev.profile_enable(True)
    a1, a2 = plan.alphas(kd)
    torch.cuda.synchronize()
    sites = set(dev.profile_report())
    dev.profile_enable(False)
    assert sites and all(s.startswith("lens.") for s in sites), sites
    del plan
    monkeypatch.setenv("AST_LENS_ROCFFT_2D", "1")
    ref = lens.LensPlan(nc, bsz)
    r1, r2 = ref.alphas(kd)
    monkeypatch.delenv("AST_LENS_ROCFFT_2D")
    for a, r in ((a1, r1), (a2, r2)):
        assert float((a - r).abs().max()) <= 1e-11 * float(r.abs().max())
    assert float(a1.abs().max()) > 0.0


@pytest.mark.parametrize("npix", [2, 37, 512])
def test_deflection_to_shear_bit_exact(lens, dev, npix):
    """ast_deflection_to_shear (SkyUtils.convert_deflection_to_shear, sky_utils.py:342-362): np.gradient with edge_order 1
    and the reference's combinations, term by term - equal to the numpy restatement bit for bit, edges included;
    SkyUtils / SkyArray wrappers return the reference's (gamma_2, gamma_1) order."""
    rng = np.random.default_rng(npix)
    a1, a2 = rng.standard_normal((npix, npix)) * 1e-3, rng.standard_normal((npix, npix)) * 1e-3
    h = np.deg2rad(10.0) / npix
    w1, w2 = ok.deflection_to_shear(a1, a2, h)
    g1, g2 = lens.deflection_to_shear(a1, a2, h)
    assert np.array_equal(g1.cpu().numpy(), w1) and np.array_equal(g2.cpu().numpy(), w2)
    from astrild_amd.rays.skys import SkyArray, SkyUtils
    s1, s2 = SkyUtils.convert_deflection_to_shear(a1, a2, npix, 10.0)
    assert np.array_equal(s1, w1) and np.array_equal(s2, w2)
    sky = SkyArray.from_array(a1.copy(), opening_angle=10.0, quantity="alpha", dir_in="")
    r2, r1 = sky.convert_deflection_to_shear(img=(a1, a2), rtn=True)
    assert np.array_equal(r1, w1) and np.array_equal(r2, w2)
    sky.data["deflty"], sky.data["defltx"] = a1, a2
    sky.convert_deflection_to_shear()
    assert np.array_equal(sky.data["gammay"], w1) and np.array_equal(sky.data["gammax"], w2)
    with pytest.raises(Exception):
        lens.deflection_to_shear(a1, a2[:-1], h)


@pytest.mark.parametrize("nc", [37, 100, 129, 1000, 3000])
def test_lens_plan_of_any_size_is_embedded_in_a_power_of_two(lens, dev, monkeypatch, nc):
    """Map sizes the hand-written passes do not cover (the reference's own test map is 100^2, test_skyutils.py:113-125;
    SkyArray.convert_convergence_to_deflection takes any npix): the convolution only pairs pixel offsets |d| < nc, so it
    is computed on the next power-of-two grid with the map's pixel size and cut-off - against the oracle (nc <= 129),
    against the exact-size rocFFT 2-D route (AST_LENS_NO_EMBED=1), and without a rocFFT kernel in the profile."""
    bsz = np.deg2rad(7.0)
    g = torch.Generator(device="cuda").manual_seed(nc)
    kd = torch.randn((nc, nc), generator=g, device="cuda", dtype=torch.float64) * 0.01
    monkeypatch.delenv("AST_LENS_NO_EMBED", raising=False)
    plan = lens.LensPlan(nc, bsz)
    dev.profile_enable(True)
    a1, a2 = plan.alphas(kd)
    phi = plan.phi(kd)
    torch.cuda.synchronize()
    sites = set(dev.profile_report())
    dev.profile_enable(False)
    assert sites and all(k.startswith("lens.") for k in sites), sites
    assert a1.shape == (nc, nc) and phi.shape == (nc, nc)
    monkeypatch.setenv("AST_LENS_NO_EMBED", "1")
    ref = lens.LensPlan(nc, bsz)
    r1, r2 = ref.alphas(kd)
    rp = ref.phi(kd)
    monkeypatch.delenv("AST_LENS_NO_EMBED")
    for got, want in ((a1, r1), (a2, r2), (phi, rp)):
        assert float((got - want).abs().max()) <= 1e-11 * float(want.abs().max())
    if nc <= 129:
        k = kd.cpu().numpy()
        o1, o2 = ok.kappa0_to_alphas(k, nc, bsz)
        npt.assert_allclose(a1.cpu().numpy(), o1, rtol=0, atol=1e-10 * abs(o1).max())
        npt.assert_allclose(a2.cpu().numpy(), o2, rtol=0, atol=1e-10 * abs(o2).max())
        npt.assert_allclose(phi.cpu().numpy(), ok.kappa0_to_phi(k, nc, bsz), rtol=0, atol=1e-10 * abs(rp.cpu().numpy()).max())
    a1b, _ = plan.alphas(kd)
    assert torch.equal(a1, a1b)
    del plan, ref


def test_large_map_just_above_a_power_of_two_keeps_the_exact_size_plan(lens, dev, monkeypatch):
    """nc = 2100 would embed in 4096 - 3.8 x the transform area and plan buffers: the exact-size rocFFT plan stays
    (embedding only up to 1.5 x the side, or for maps <= 2048 px); same deflection field as the embedded plan."""
    nc, bsz = 2100, np.deg2rad(5.0)
    g = torch.Generator(device="cuda").manual_seed(nc)
    kd = torch.randn((nc, nc), generator=g, device="cuda", dtype=torch.float64) * 0.01
    plan = lens.LensPlan(nc, bsz)
    dev.profile_enable(True)
    a1, a2 = plan.alphas(kd)
    torch.cuda.synchronize()
    sites = set(dev.profile_report())
    dev.profile_enable(False)
    assert any(not k.startswith("lens.") for k in sites), sites        # a rocFFT plan ran
    monkeypatch.setenv("AST_LENS_EMBED_ALWAYS", "1")
    emb = lens.LensPlan(nc, bsz)
    e1, e2 = emb.alphas(kd)
    monkeypatch.delenv("AST_LENS_EMBED_ALWAYS")
    for got, want in ((a1, e1), (a2, e2)):
        assert float((got - want).abs().max()) <= 1e-11 * float(want.abs().max())
