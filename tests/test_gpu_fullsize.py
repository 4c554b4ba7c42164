"""GPU, BASELINE.json sizes: size-independent properties of the hot path where the
oracle is too slow — mass conservation, Parseval, direct-vs-tiled and
natural-vs-shuffled agreement, fp32-vs-fp64 agreement, linearity of the stack,
mean preservation of the periodic smoothing, histogram totals."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev(hip):
    from astrild_amd import device
    torch.cuda.set_device(0)
    return device


@pytest.mark.parametrize("n", [512, 1024])
def test_cic_pk_pipeline_properties_at_baseline_size(dev, n):
    L = 1000.0
    pos = dev.synth_lattice_particles(n, n, L, seed=20240601, dtype=torch.float32)
    npart = pos.shape[0]
    grid = dev.paint(pos, None, n, L, "cic", method="tiled")
    # (1) mass conservation: sum(grid) == Np (fp64 reduction of an fp32 grid)
    total = float(grid.sum(dtype=torch.float64))
    assert total == pytest.approx(npart, rel=1e-7)
    assert float(grid.min()) >= 0.0
    # (2) Parseval: sum_modes w |delta_k|^2 == <f^2>
    spec = dev.r2c(grid)
    w = torch.full((n // 2 + 1,), 2.0, dtype=torch.float64, device="cuda")
    w[0] = w[-1] = 1.0
    lhs = float(((spec.real.double() ** 2 + spec.imag.double() ** 2) * w).sum())
    rhs = float((grid.double() ** 2).mean())
    assert lhs == pytest.approx(rhs, rel=2e-6)
    # (3) shell sums: every mode of the half lattice inside the Nyquist sphere is counted once
    ks, ps, nm = dev.power_bin_1d(spec, None, n, L)
    res = dev.finish_power(ks, ps, nm)
    assert int(nm.sum()) > 0 and np.isfinite(res["power"]).all() and (res["power"] > 0).all()
    kf = 2 * np.pi / L
    assert np.all(res["k"] >= kf * np.arange(1, n // 2)) and np.all(res["k"] < kf * np.arange(2, n // 2 + 1))
    # DC excluded: total shell power < total power
    assert float(ps.sum()) / L ** 3 < lhs
    del spec
    # (4) direct (global atomics) and tiled (LDS) paints agree
    if n == 512:
        g2 = dev.paint(pos, None, n, L, "cic", method="direct")
        assert float((g2 - grid).abs().max()) < 5e-6
        del g2
    # (5) shuffled order gives the same grid (same multiset of particles)
    pos_s = dev.synth_lattice_particles(n, n, L, seed=20240601, dtype=torch.float32, shuffle=True)
    g3 = dev.paint(pos_s, None, n, L, "cic", method="tiled")
    assert float((g3 - grid).abs().max()) < 5e-6
    del g3, pos_s
    # (6) fp32 pipeline against the fp64 pipeline on the same fp32 positions
    if n == 512:
        g64 = dev.paint(pos.double(), None, n, L, "cic", method="tiled")
        r64 = dev.fftpower_1d(g64, L)
        r32 = dev.finish_power(ks, ps, nm)
        assert np.array_equal(r64["modes"], r32["modes"])
        # high-k shells (white, well above fp32 FFT round-off): north_star tolerance
        np.testing.assert_allclose(r32["power"][n // 8:], r64["power"][n // 8:], rtol=1e-6)
        # cold-lattice low-k shells carry ~1e-5 of the peak power: compare against the peak
        np.testing.assert_allclose(r32["power"], r64["power"], rtol=0, atol=1e-6 * r64["power"].max())


def test_tsc_mass_conservation_and_translation_at_512(dev):
    n, L = 512, 1000.0
    pos = dev.synth_lattice_particles(n, n, L, seed=7, dtype=torch.float32)
    g = dev.paint(pos, None, n, L, "tsc", method="tiled")
    assert float(g.sum(dtype=torch.float64)) == pytest.approx(pos.shape[0], rel=1e-7)
    # integer-cell translation (exact in fp32 for this box: 3 cells = 5.859375)
    shift = torch.tensor([3 * L / n, 5 * L / n, 17 * L / n], dtype=torch.float32, device="cuda")
    moved = torch.remainder(pos + shift, L)
    g2 = dev.paint(moved.contiguous(), None, n, L, "tsc", method="tiled")
    ref = torch.roll(g, shifts=(3, 5, 17), dims=(0, 1, 2))
    # positions were re-rounded to fp32 after the shift: agreement at the input's own resolution
    assert float((g2 - ref).abs().max()) < 2e-3
    assert float((g2 - ref).abs().mean()) < 2e-5


def test_kappa_stack_and_pipeline_properties_at_4096(hip, dev):
    from astrild_amd import lensing
    npix, P = 4096, 64
    planes = lensing.synth_kappa_planes(P, npix)
    tot = lensing.kappa_stack(planes)
    # stack == torch's own sequential sum, bit for bit (fp64 IEEE adds in plane order)
    ref = planes[0].clone()
    for p in planes[1:]:
        ref += p
    assert torch.equal(tot, ref)
    # linearity in the weights: stack(2*w) == 2*stack(w) exactly (power-of-two scaling)
    wn, wd = np.linspace(0.5, 1.5, P), np.full(P, 0.75)
    a = lensing.kappa_stack(planes, wn, wd)
    b = lensing.kappa_stack(planes, 2 * wn, wd)
    assert torch.equal(b, 2 * a)
    del planes, ref, b
    # periodic Gaussian smoothing preserves the mean and reduces the variance
    m0, v0 = float(a.mean()), float(a.var())
    sm = a.clone()
    lensing.smooth_plan(npix).gaussian(sm, 3.4, "gaussianFFT")
    assert float(sm.mean()) == pytest.approx(m0, rel=1e-9, abs=1e-15)
    assert float(sm.var()) < v0
    # histogram totals
    counts, edges = lensing.histogram(sm, 200)
    assert counts.sum() == npix * npix and len(edges) == 201
    # kappa -> alpha is linear and vanishes for a zero map
    plan = lensing.lens_plan(npix, np.deg2rad(20.0))
    a1, a2 = plan.alphas(sm)
    b1, b2 = plan.alphas(2 * sm)
    scale = float(a1.abs().max())
    assert float((b1 - 2 * a1).abs().max()) < 1e-12 * scale
    z1, z2 = plan.alphas(torch.zeros_like(sm))
    assert float(z1.abs().max()) == 0.0 and float(z2.abs().max()) == 0.0


def _triangles_brute_force(n, lo_hi_i, lo_hi_j, lo_hi_l):
    """Exact integer count of closed triangles q1 + q2 + q3 = 0 with |q1| in shell i, |q2| in j, |q3| in l on the
    FULL integer lattice (independent of any FFT): enumerate q1, q2 and test -(q1+q2), with the DFT's aliasing."""
    def shell_vectors(lo, hi):
        r = np.arange(-hi, hi + 1)
        g = np.stack(np.meshgrid(r, r, r, indexing="ij"), axis=-1).reshape(-1, 3)
        m2 = (g ** 2).sum(axis=1)
        return g[(m2 >= lo * lo) & (m2 < hi * hi)]
    a, b = shell_vectors(*lo_hi_i), shell_vectors(*lo_hi_j)
    lo, hi = lo_hi_l
    total = 0
    for chunk in np.array_split(a, max(1, len(a) * len(b) // 4_000_000)):
        q3 = -(chunk[:, None, :] + b[None, :, :])
        q3 = (q3 + n // 2) % n - n // 2          # alias into [-n/2, n/2)
        q3 = np.where(q3 == -n // 2, n // 2, q3)   # the DFT keeps index n/2 (|m| is what matters)
        m2 = (q3.astype(np.int64) ** 2).sum(axis=2)
        total += int(((m2 >= lo * lo) & (m2 < hi * hi)).sum())
    return total


def test_config_e_bispectrum_512(dev):
    """BASELINE.json configs[4] at its stated size: FFT triangle counting on a 512^3 grid.
    N_tri: within 0.02 of an integer for every bin (fp64 I-fields), and the lowest bins equal an
    independent integer enumeration exactly; B: fp32 pipeline against the fp64 pipeline."""
    n, L, width = 512, 1000.0, 8
    pos = dev.synth_lattice_particles(n, n, L, seed=20240601, dtype=torch.float32)
    grid = dev.paint(pos, None, n, L, "cic")
    edges = list(range(1, n // 2 + 1, width))
    nsh = len(edges) - 1
    tri = [(i, i, i) for i in range(nsh)] + [(0, i, i) for i in range(1, nsh)] + \
          [(i, i, min(nsh - 1, 2 * i)) for i in range(1, nsh // 2)]
    dev._tri_cache.clear()
    r32 = dev.bispectrum(grid, L, edges, tri)
    assert r32["ntri_residual"] < 0.02                # sum I I I / Ng sits this close to an integer (counts up to 1e12)
    assert (r32["ntri"] >= 0).all() and r32["ntri"][:nsh].min() > 0
    # exact integer check, no FFT involved: equilateral (0,0,0), squeezed (0,1,1), isosceles (1,1,2)
    for t in [(0, 0, 0), (0, 1, 1), (1, 1, 2)]:
        want = _triangles_brute_force(n, *[(edges[s], edges[s + 1]) for s in t])
        assert int(r32["ntri"][tri.index(t)]) == want, t
    g64 = dev.paint(pos.double(), None, n, L, "cic")
    del pos
    r64 = dev.bispectrum(g64, L, edges, tri)
    assert np.array_equal(r64["ntri"], r32["ntri"])
    ok = r64["ntri"] > 0
    scale = np.abs(r64["B"][ok]).max()
    # fp32 shell fields (1e-7 per cell, 1.3e8 cells, three factors): 2e-4 of each bin, with a floor at 1e-5 of the
    # largest |B| for the bins near a zero crossing
    np.testing.assert_allclose(r32["B"][ok], r64["B"][ok], rtol=2e-4, atol=1e-5 * scale)
    dev._tri_cache.clear()
    dev.clear_plan_cache()


@pytest.mark.parametrize("hint", [None, "xsorted"])
def test_bench_path_at_1024_against_the_float64_pipeline(dev, hint):
    """What bench.py times, at the size it times it (hint=None is bench.py's own call for natural order: two big
    launches, tflags 6; "xsorted" the chunked pipeline): fp32 paint(defer_fold, offset=mean) +
    power_sums_fused(halo=) (rows_r2c with the halo fold and the low-k z sums, the x pass fused with the shell
    binning) against the float64 pipeline on the SAME fp32 positions: mode counts equal, every shell within 1e-6
    (north_star), and the shell sums of the fused path equal to those of its own folded grid transformed separately."""
    n, L = 1024, 1000.0
    pos = dev.synth_lattice_particles(n, n, L, seed=20240601, dtype=torch.float32)
    mean = pos.shape[0] / float(n) ** 3
    grid, halo = dev.paint(pos, None, n, L, "cic", method="tiled", accumulate=False, defer_fold=True, offset=mean,
                           hint=hint, check_dropped=False)
    r32 = dev.finish_power(*dev.power_sums_fused(grid, L, halo=halo))
    del grid, halo
    r64 = dev.paint_power_1d(pos.double(), None, n, L, "cic")
    assert np.array_equal(r32["modes"], r64["modes"])
    np.testing.assert_allclose(r32["k"], r64["k"], rtol=1e-12)
    np.testing.assert_allclose(r32["power"], r64["power"], rtol=1e-6)
    # Parseval on the same grid with the fold done by the paint: sum over ALL modes of w |delta_k|^2 == <delta^2>,
    # and the shells of the fused path add up to the part of it inside the Nyquist sphere
    g2 = dev.paint(pos, None, n, L, "cic", method="tiled", accumulate=False, offset=mean)
    del pos
    rhs = float((g2.double() ** 2).mean())
    spec = dev.r2c(g2)
    del g2
    w = torch.full((n // 2 + 1,), 2.0, dtype=torch.float64, device="cuda")
    w[0] = w[-1] = 1.0
    p3 = spec.real.double() ** 2
    p3 += spec.imag.double() ** 2
    lhs = float((p3 * w).sum())
    del p3
    assert lhs == pytest.approx(rhs, rel=2e-6)
    ks, ps, nm = dev.power_bin_1d(spec, None, n, L)
    inside = float(ps.sum()) / L ** 3
    fused = float(np.nansum(r32["power"] * r32["modes"])) / L ** 3
    assert fused == pytest.approx(inside, rel=2e-6) and inside < lhs


def test_shuffled_tsc_at_512_against_the_float64_pipeline(dev):
    """Unordered particles + TSC (what stats_subfind.py:125-131 feeds the paint): the scattered fp32 path with the
    fused transform against the float64 pipeline on the same positions."""
    n, L = 512, 1000.0
    pos = dev.synth_lattice_particles(n, n, L, seed=7, dtype=torch.float32, shuffle=True)
    mean = pos.shape[0] / float(n) ** 3
    grid, halo = dev.paint(pos, None, n, L, "tsc", method="tiled", accumulate=False, defer_fold=True, offset=mean,
                           hint="scattered", check_dropped=False)
    r32 = dev.finish_power(*dev.power_sums_fused(grid, L, halo=halo))
    del grid, halo
    r64 = dev.paint_power_1d(pos.double(), None, n, L, "tsc")
    assert np.array_equal(r32["modes"], r64["modes"])
    np.testing.assert_allclose(r32["power"], r64["power"], rtol=1e-6)


@pytest.mark.parametrize("window,shuffle", [("cic", False), ("tsc", True)])
def test_config_b_512_against_the_oracle(dev, window, shuffle):
    """BASELINE.json configs[1] at its stated size against the ORACLE itself (not another HIP pipeline): 512^3
    device-generated fp32 positions -> host -> oracle paint + fftpower_1d in float64 (power_spectrum_3d.py:183-224,
    stats_subfind.py:130-150), against paint_power_1d on the same positions in float64 (modes equal, k 1e-12, every
    shell 1e-9) and in fp32 (the fused path bench.py times; every shell 1e-6, north_star).  The second case is the
    unordered TSC input that stats_subfind.py:125-131 feeds the paint (scattered path)."""
    from oracle import mesh as omesh, fftpower as offt
    n, L = 512, 1000.0
    pos = dev.synth_lattice_particles(n, n, L, seed=20240601, dtype=torch.float32, shuffle=shuffle)
    host = pos.cpu().numpy()
    ref = offt.fftpower_1d(omesh.paint(host, None, n, L, window), L)
    del host
    r64 = dev.paint_power_1d(pos.double(), None, n, L, window)
    np.testing.assert_array_equal(r64["modes"], ref["modes"])
    np.testing.assert_allclose(r64["k"], ref["k"], rtol=1e-12)
    np.testing.assert_allclose(r64["power"], ref["power"].real, rtol=1e-9)
    mean = pos.shape[0] / float(n) ** 3
    grid, halo = dev.paint(pos, None, n, L, window, method="tiled", accumulate=False, defer_fold=True, offset=mean,
                           hint="scattered" if shuffle else None, check_dropped=False)
    r32 = dev.finish_power(*dev.power_sums_fused(grid, L, halo=halo))
    np.testing.assert_array_equal(r32["modes"], ref["modes"])
    np.testing.assert_allclose(r32["power"], ref["power"].real, rtol=1e-6)


def test_side_2048_single_precision_passes_against_the_double_passes(dev, monkeypatch):
    """1024^3 lattice particles on a 2048^3 fp32 grid (the largest cube one GPU holds; domain_level is arbitrary in the
    reference, power_spectrum_3d.py:183-188): the three-stage single-precision passes + double-precision low-k patch
    (ast_fft32_big_power_3d) against the double passes over the same fp32 grid (ast_fft64_power_3d_f32) - every shell
    within 1e-6 (north_star), mode counts equal - and through paint_power_1d."""
    n, npside, L = 2048, 1024, 1000.0
    pos = dev.synth_lattice_particles(npside, n, L, seed=20240601, dtype=torch.float32)
    grid = dev.paint(pos, None, n, L, "cic", method="tiled", offset="mean", check_dropped=False)
    fast = dev.finish_power(*dev.power_sums_fused64(grid, L, mean=0.0))
    torch.cuda.synchronize()
    dev._power_scratch.clear()
    torch.cuda.empty_cache()
    monkeypatch.setenv("ASTRILD_FFT32_BIG_OFF", "1")
    ref = dev.finish_power(*dev.power_sums_fused64(grid, L, mean=0.0))
    monkeypatch.delenv("ASTRILD_FFT32_BIG_OFF")
    dev._power_scratch.clear()
    del grid
    torch.cuda.empty_cache()
    assert np.array_equal(fast["modes"], ref["modes"])
    rel = np.abs(fast["power"] / ref["power"] - 1.0)
    assert rel.max() < 1e-6, (int(rel.argmax()), float(rel.max()))
    whole = dev.paint_power_1d(pos, None, n, L, "cic")
    np.testing.assert_allclose(whole["power"], ref["power"], rtol=1e-6)
    dev._power_scratch.clear()
    torch.cuda.empty_cache()
