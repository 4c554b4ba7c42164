"""GPU: the hand-written LDS-tiled FFT passes against numpy (float64) and rocFFT.
fp32 tolerance: 1e-6 of the spectrum's rms per mode (north_star float tolerance)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev(hip):
    from astrild_amd import device
    torch.cuda.set_device(0)
    return device


@pytest.mark.parametrize("n", [256, 512, 1024])
@pytest.mark.parametrize("ncols,batch", [(16, 1), (37, 3), (513, 2)])
def test_strided_c2c_pass(dev, hip, n, ncols, batch):
    from astrild_amd import _lib
    rng = np.random.default_rng(n + ncols)
    pitch = ncols + 5                                   # element stride larger than the tile row
    a = (rng.standard_normal((batch, n, pitch)) + 1j * rng.standard_normal((batch, n, pitch))).astype(np.complex64)
    t = dev.as_device(a.copy())
    _lib.check(hip.ast_fft_tile_c2c(dev.ptr(t), 0, n, pitch, ncols, batch, n * pitch, 0.5, dev.stream()))
    got = t.cpu().numpy()
    ref = 0.5 * np.fft.fft(a.astype(np.complex128), axis=1)
    rms = np.sqrt(np.mean(np.abs(ref[:, :, :ncols]) ** 2))
    np.testing.assert_allclose(got[:, :, :ncols], ref[:, :, :ncols], rtol=0, atol=2e-6 * rms)
    assert np.array_equal(got[:, :, ncols:], a[:, :, ncols:])           # padding columns untouched


@pytest.mark.parametrize("n", [256, 512, 1024])
@pytest.mark.parametrize("nrows", [1, 16, 50])
def test_rows_r2c_pass(dev, hip, n, nrows):
    from astrild_amd import _lib
    rng = np.random.default_rng(n + nrows)
    x = rng.standard_normal((nrows, n)).astype(np.float32)
    t = dev.as_device(x)
    out = torch.zeros((nrows, n // 2 + 1), dtype=torch.complex64, device="cuda")
    _lib.check(hip.ast_fft_tile_rows_r2c(dev.ptr(t), dev.ptr(out), 0, n, nrows, n, n // 2 + 1, 1.0, dev.stream()))
    ref = np.fft.rfft(x.astype(np.float64), axis=1)
    rms = np.sqrt(np.mean(np.abs(ref) ** 2))
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=2e-6 * rms)


@pytest.mark.parametrize("n", [256, 512])
def test_r2c_3d_tile_vs_rocfft_and_numpy(dev, n):
    rng = np.random.default_rng(n)
    f = (1.0 + rng.standard_normal((n, n, n))).astype(np.float32)
    t = dev.as_device(f)
    a = dev.r2c(t, engine="tile").cpu().numpy().astype(np.complex128)
    b = dev.r2c(t, engine="rocfft").cpu().numpy().astype(np.complex128)
    ref = np.fft.rfftn(f.astype(np.float64)) / f.size
    norm = np.sqrt(np.sum(np.abs(ref) ** 2))
    err_tile = np.sqrt(np.sum(np.abs(a - ref) ** 2)) / norm
    err_roc = np.sqrt(np.sum(np.abs(b - ref) ** 2)) / norm
    # relative L2 error of an fp32 FFT is a few 1e-7; north_star float tolerance is 1e-6
    assert err_tile < 1e-6
    assert err_tile < 2 * err_roc + 1e-8                      # no less accurate than rocFFT
    assert abs(a[0, 0, 0] - f.astype(np.float64).mean()) < 3e-7
    # per-mode worst case (set by the O(1) mean riding through the passes in fp32): on par with rocFFT
    ref[0, 0, 0] = a[0, 0, 0] = b[0, 0, 0] = 0
    assert np.abs(a - ref).max() < 3 * np.abs(b - ref).max() + 1e-12
    assert np.abs(a - ref).max() < 2e-7            # a few fp32 ulp of the mean


def test_tile_engine_rejects_unsupported(dev):
    with pytest.raises(Exception):
        dev.r2c(torch.zeros((64, 64, 64), dtype=torch.float32, device="cuda"), engine="tile")
    with pytest.raises(Exception):
        dev.r2c(torch.zeros((64, 64, 64), dtype=torch.float64, device="cuda"), engine="tile")
    out64 = dev.r2c(torch.ones((256, 256, 256), dtype=torch.float64, device="cuda"), engine="tile")      # double passes
    assert abs(complex(out64[0, 0, 0]) - 1.0) < 1e-13 and float(out64.abs().sum()) - 1.0 < 1e-9
    out = dev.r2c(torch.ones((64, 64, 64), dtype=torch.float32, device="cuda"))          # auto -> rocFFT
    assert abs(complex(out[0, 0, 0]) - 1.0) < 1e-6


@pytest.mark.parametrize("n", [256, 512])
def test_fused_fft_power_matches_unfused_and_oracle(dev, n):
    from oracle import fftpower as offt
    rng = np.random.default_rng(n + 1)
    f = (1.0 + 0.5 * rng.standard_normal((n, n, n))).astype(np.float32)
    t = dev.as_device(f)
    L = 750.0
    fused = dev.fftpower_1d(t, L)                     # fused tile path
    plain = dev.fftpower_1d(t, L, fused=False)        # spectrum to HBM + separate binning
    assert np.array_equal(fused["modes"], plain["modes"])
    np.testing.assert_allclose(fused["k"], plain["k"], rtol=0, atol=0)
    # two fp32 pipelines; the fused one takes its five lowest shells from the double-precision low-k channel, the
    # plain one carries the O(1) mean's round-off there
    np.testing.assert_allclose(fused["power"][5:], plain["power"][5:], rtol=4e-6)
    np.testing.assert_allclose(fused["power"], plain["power"], rtol=3e-5)
    if n == 256:
        ref = offt.fftpower_1d(f, L)
        assert np.array_equal(fused["modes"], ref["modes"])
        # fp32 grid + fp32 FFT (+ low-k channel) against the float64 oracle: 1e-6 on every shell
        np.testing.assert_allclose(fused["power"], ref["power"].real, rtol=1e-6)


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("n", [256, 512])
def test_forward_pruning_to_the_nyquist_disc_changes_no_bit(dev, n, dtype):
    """FFTPower(mode="1d", kmin=k_F) drops |m| >= N/2 (power_spectrum_3d.py:189-195): the y pass does not store the rows with
    k_y^2 + k_z0^2 > (N/2)^2 and the binning pass leaves their tiles out.  Same shell sums as the unpruned passes
    (AST_FFT_NO_PRUNE) - bit for bit in fp32 - under both edge rules (a vector of norm exactly N/2 may fall into the last shell)."""
    import os
    g = torch.Generator(device="cuda").manual_seed(n + 5)
    t = torch.randn((n, n, n), dtype=torch.float32 if dtype == "f32" else torch.float64, device="cuda", generator=g)
    fn = dev.power_sums_fused if dtype == "f32" else dev.power_sums_fused64
    for L in (1000.0, 750.0):                          # the float64 rule's edge falls depend on L
        for rule in ("float64", "integer"):
            _, pruned, _ = fn(t, L, binning=rule)
            os.environ["AST_FFT_NO_PRUNE"] = "1"
            try:
                _, full, _ = fn(t, L, binning=rule)
            finally:
                del os.environ["AST_FFT_NO_PRUNE"]
            if dtype == "f32":
                assert torch.equal(pruned, full), (L, rule)
            else:      # double products added by LDS atomics in arrival order: the last bits move from run to run either way
                np.testing.assert_allclose(pruned.cpu().numpy(), full.cpu().numpy(), rtol=1e-12)
            assert float(pruned[-1]) > 0.0


def test_mean_subtraction_recovers_cold_low_k_shells_in_fp32(dev):
    # lattice + small jitter: low-k power is ~1e-5 of the peak; the fp32 FFT round-off of the
    # O(1) mean density swamps it unless the mean is removed on load
    n, L = 256, 1000.0
    pos = dev.synth_lattice_particles(n, n, L, seed=3, dtype=torch.float32)
    g32 = dev.paint(pos, None, n, L, "cic")
    ref = dev.fftpower_1d(dev.paint(pos.double(), None, n, L, "cic"), L)          # fp64 pipeline
    plain = dev.finish_power(*dev.power_sums_fused(g32, L, lowk=False))
    centred = dev.finish_power(*dev.power_sums_fused(g32, L, mean=1.0, lowk=False))
    err_plain = np.abs(plain["power"] / ref["power"] - 1)
    err_centred = np.abs(centred["power"] / ref["power"] - 1)
    assert err_centred.max() < 2e-5                     # fp32 grid cells still carry 6e-8 of the mean
    assert err_centred[:8].max() < 0.2 * err_plain[:8].max() + 1e-7
    np.testing.assert_allclose(centred["power"][n // 8:], ref["power"][n // 8:], rtol=1e-6)


@pytest.mark.parametrize("window", ["cic", "tsc"])
@pytest.mark.parametrize("n", [256, 512])
def test_deferred_fold_in_the_z_pass_is_bit_identical(hip, window, n):
    """paint(defer_fold=True) + power_sums_fused(halo=) against the paint that folds its halo records
    itself: the z pass adds the same record lines in the same order, so the shell sums are the same bits."""
    from astrild_amd import device as dev
    torch.cuda.set_device(0)
    L = 1000.0
    pos = dev.synth_lattice_particles(n, n, L, seed=3, dtype=torch.float32)
    mean = 1.0
    grid = dev.paint(pos, None, n, L, window, method="tiled")
    _, ref, _ = dev.power_sums_fused(grid, L, mean=mean)
    grid2, halo = dev.paint(pos, None, n, L, window, method="tiled", defer_fold=True)
    _, got, _ = dev.power_sums_fused(grid2, L, mean=mean, halo=halo)
    assert torch.equal(got, ref)
    assert not torch.equal(grid2, grid)            # the deferred grid alone is incomplete


def test_paint_power_pipeline_with_mass_matches_separate_calls(hip):
    """device.paint_power_1d (SubFind.power_spectrum's pipeline: TSC paint with masses / dx^3, then FFTPower)
    on its fused fp32 path against paint + fftpower_1d."""
    from astrild_amd import device as dev
    torch.cuda.set_device(0)
    rng = np.random.default_rng(8)
    n, L, npart = 256, 500.0, 300000
    pos = dev.as_device(rng.uniform(0, L, size=(npart, 3)).astype(np.float32))
    mass = dev.as_device(rng.uniform(0.5, 2.0, size=npart).astype(np.float32))
    dx = L / n
    got = dev.paint_power_1d(pos, mass, n, L, "tsc", scale=1.0 / dx ** 3)
    ref = dev.fftpower_1d(dev.paint(pos, mass, n, L, "tsc", scale=1.0 / dx ** 3), L)
    assert np.array_equal(got["modes"], ref["modes"])
    # the pipeline's grid holds rho - mean (rounded once), the separate paint rho: both fp32, same spectrum
    np.testing.assert_allclose(got["power"], ref["power"], rtol=1e-6)
    from oracle import mesh as omesh, fftpower as offt
    ref64 = offt.fftpower_1d(omesh.paint(pos.cpu().numpy(), mass.cpu().numpy(), n, L, "tsc") / dx ** 3, L)
    np.testing.assert_allclose(got["power"], ref64["power"].real, rtol=1e-6)


def test_shell_lookup_isqrt_is_exact_for_every_mode_norm(hip):
    """The fused binning takes floor(sqrt(|m|^2)) from one hardware sqrt of |m|^2 + 1/2: exact for all
    |m|^2 up to 3 * 512^2 (side 1024), checked against integer arithmetic."""
    import math
    from astrild_amd import _lib
    from astrild_amd.device import ptr, stream
    torch.cuda.set_device(0)
    count = 3 * 512 * 512 + 1
    out = torch.empty(count, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().ast_fft_tile_isqrt_table(ptr(out), count, stream()), "ast_fft_tile_isqrt_table")
    v = np.arange(count, dtype=np.int64)
    ref = np.floor(np.sqrt(v.astype(np.float64))).astype(np.int64)
    ref -= ref * ref > v
    ref += (ref + 1) * (ref + 1) <= v
    assert np.array_equal(out.cpu().numpy().astype(np.int64), ref)
    assert math.isqrt(count - 1) == int(ref[-1])


@pytest.mark.parametrize("n,window", [(256, "cic"), (512, "tsc")])
def test_lowk_double_precision_channel_meets_1e6_on_every_shell(hip, n, window):
    """The cold lattice's lowest shells hold 1e-5 of the peak power; the fp32 transform's round-off floor leaves them
    ~2e-6 / |m|^2 off.  With the low-k channel (modes |m_i| <= 5 as DFT sums in double, halo records folded the
    same way) every shell of the fp32 pipeline agrees with the fp64 pipeline on the same positions to 1e-6."""
    from astrild_amd import device as dev
    torch.cuda.set_device(0)
    L = 1000.0
    pos = dev.synth_lattice_particles(n, n, L, seed=20240601, dtype=torch.float32)
    ref = dev.fftpower_1d(dev.paint(pos.double(), None, n, L, window, method="tiled"), L)          # fp64 grid, rocFFT fp64
    grid, halo = dev.paint(pos, None, n, L, window, method="tiled", defer_fold=True, offset="mean")
    with_lowk = dev.finish_power(*dev.power_sums_fused(grid, L, halo=halo, lowk=True))
    without = dev.finish_power(*dev.power_sums_fused(grid, L, halo=halo, lowk=False))
    assert np.array_equal(with_lowk["modes"], ref["modes"])
    rel = np.abs(with_lowk["power"] / ref["power"] - 1.0)
    rel0 = np.abs(without["power"] / ref["power"] - 1.0)
    assert rel.max() < 1e-6, rel[:8]
    assert rel[:5].max() < 6e-7                              # the patched shells: only the fp32 CELL rounding is left
    assert np.array_equal(with_lowk["power"][5:], without["power"][5:])      # the other shells are untouched
    assert rel0[:5].max() > rel[:5].max()
    # the plain (folded) grid path takes the same channel
    grid2 = dev.paint(pos, None, n, L, window, method="tiled")
    plain = dev.finish_power(*dev.power_sums_fused(grid2, L, mean=1.0, lowk=True))
    assert np.abs(plain["power"] / ref["power"] - 1.0).max() < 1e-6


@pytest.mark.parametrize("n", [256, 512])
def test_tile_c2r_3d_and_fused_shell_mask(hip, n):
    """The inverse tile passes (x, y, z with the conjugation trick, rows C2R): unnormalised inverse of the forward
    transform, and the same restricted to a shell against ast_shell_filter + the rocFFT C2R plan."""
    from astrild_amd import device as dev
    torch.cuda.set_device(0)
    rng = np.random.default_rng(n)
    f = rng.standard_normal((n, n, n)).astype(np.float32)
    t = dev.as_device(f)
    spec = dev.r2c(t)                                              # rfftn / Ng
    keep = spec.clone()
    back = dev.c2r_tile(spec)                                      # sum_k spec_k e^{ikx} = f
    assert torch.equal(spec, keep)                                 # the spectrum is left intact
    err = (back - t).abs().max().item()
    assert err < 5e-6 * np.abs(f).max(), err
    # the masked transform is pruned (tiles and rows a shell leaves zero are neither written nor read): the work
    # buffer starts as NaN, so a read of anything the earlier pass skipped would show
    for lo, hi in [(0, 1), (1, 2), (1, 9), (15, 17), (40, 48), (n // 2 - 8, n // 2), (n // 2, n // 2 + 40), (1, n)]:
        masked = dev.shell_filter(spec, n, lo, hi)
        ref = dev.c2r(masked, (n, n, n))
        work = torch.full_like(spec, float("nan"))
        got = dev.c2r_tile(spec, work=work, m_lo=lo, m_hi=hi)
        scale = ref.abs().max().item()
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max().item() < 5e-6 * scale, (lo, hi)
    assert hip.ast_fft_tile_c2r_3d(dev.ptr(spec), dev.ptr(spec), dev.ptr(back), 0, n, 0, 0, 1.0, dev.stream()) < 0   # work == spec


@pytest.mark.parametrize("n", [256, 512])
def test_batched_inverse_passes_equal_the_single_shell_calls(hip, n):
    """ast_fft_tile_c2r_3d_batch (up to 8 shells per launch, the shell = the launches' second grid dimension) leaves, for
    every shell, exactly the field the single-shell call leaves - small and large shells in one batch, work arrays
    poisoned with NaN (nothing a shell's pruning skips is read), a batch of one, and bad arguments refused."""
    from astrild_amd import device as dev
    torch.cuda.set_device(0)
    g = torch.Generator(device="cuda").manual_seed(n)
    spec = dev.r2c(torch.randn((n, n, n), generator=g, device="cuda", dtype=torch.float32))
    keep = spec.clone()
    shells = [(1, 9), (n // 2 - 8, n // 2), (0, 1), (33, 41), (9, 17), (1, n), (n // 4, n // 4 + 8), (100, 101)]
    works = [torch.full_like(spec, float("nan")) for _ in shells]
    got = dev.c2r_tile_batch(spec, shells, works, xy_batch=8)
    assert torch.equal(spec, keep)
    for xy in (1, 3):                                   # x / y passes shell by shell or three at a time, z passes in one launch
        works2 = [torch.full_like(spec, float("nan")) for _ in shells]
        got2 = dev.c2r_tile_batch(spec, shells, works2, xy_batch=xy)
        assert all(torch.equal(a, b) for a, b in zip(got, got2)), xy
    del works2, got2
    for (lo, hi), field in zip(shells, got):
        ref = dev.c2r_tile(spec, work=torch.full_like(spec, float("nan")), m_lo=lo, m_hi=hi)
        assert torch.isfinite(field).all() and torch.equal(field, ref), (lo, hi)
    one = dev.c2r_tile_batch(spec, shells[3:4], works)
    assert torch.equal(one[0], got[3])
    # scratch spectra with line-aligned rows (what device.bispectrum uses): the same fields, bit for bit
    pitch = dev.tile_work_pitch(n)
    assert pitch % 16 == 0 and n // 2 + 1 <= pitch < n // 2 + 17
    worksp = [torch.full((n, n, pitch), float("nan"), dtype=spec.dtype, device="cuda") for _ in shells]
    for xy in (1, 8):
        gotp = dev.c2r_tile_batch(spec, shells, worksp, xy_batch=xy)
        assert all(torch.equal(a, b) for a, b in zip(got, gotp)), xy
    del worksp, gotp
    import ctypes as ct
    wp = (ct.c_void_p * 2)(works[0].data_ptr(), works[0].data_ptr())        # the same work array twice
    op = (ct.c_void_p * 2)(got[0].data_ptr(), got[1].data_ptr())
    lo, hi = (ct.c_int * 2)(1, 9), (ct.c_int * 2)(9, 17)
    assert hip.ast_fft_tile_c2r_3d_batch(dev.ptr(spec), wp, op, 0, n, lo, hi, 2, 1.0, 3, 0, dev.stream()) < 0
    assert hip.ast_fft_tile_c2r_3d_batch(dev.ptr(spec), wp, op, 0, n, lo, hi, 9, 1.0, 3, 0, dev.stream()) < 0
    assert hip.ast_fft_tile_c2r_3d_batch(dev.ptr(spec), wp, op, 0, n, lo, hi, 1, 1.0, 0, 0, dev.stream()) < 0
    assert hip.ast_fft_tile_c2r_3d_batch(dev.ptr(spec), wp, op, 0, n, lo, hi, 1, 1.0, 3, n // 2, dev.stream()) < 0      # pitch < n/2+1


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_triple_product_sums_one_pass(hip, dtype):
    """All triangle sums in one pass over the fields (ast_triple_product_sums) against float64 numpy; ragged cell
    count (not a multiple of the chunk), repeated shells, more triangles than one batch, run-to-run identical."""
    from astrild_amd import device as dev
    torch.cuda.set_device(0)
    rng = np.random.default_rng(5)
    count, nf = 3 * 256 * 41 + 77, 7
    host = {2 * s + 1: rng.standard_normal(count).astype(np.float32) for s in range(nf)}     # keys need not be 0..nf-1
    fields = {k: dev.as_device(v).to(dtype) for k, v in host.items()}
    keys = sorted(host)
    tri = [(a, b, c) for a in keys for b in keys for c in keys if a <= b <= c]               # 84 triangles
    tri = tri + [(keys[0], keys[1], keys[2])] * 200                                          # 284 > 256: two batches
    got = dev.triple_product_sums(fields, tri)
    again = dev.triple_product_sums(fields, tri)
    assert torch.equal(got, again)
    ref = np.array([np.sum(host[a].astype(np.float64) * host[b].astype(np.float64) * host[c].astype(np.float64))
                    for a, b, c in tri])
    scale = np.array([np.sum(np.abs(host[a].astype(np.float64) * host[b] * host[c])) for a, b, c in tri])
    # float64 fields: exact products, sums in double.  float32 fields: the two products of a term are formed in the
    # fields' own precision (two roundings of 2^-24 each, unbiased), the sum in double
    tol = 1e-13 if dtype == torch.float64 else 1.5e-7
    assert np.all(np.abs(got.cpu().numpy() - ref) <= tol * scale)
    if dtype == torch.float32:                          # ... and the rounding errors do not add up coherently
        assert np.median(np.abs(got.cpu().numpy() - ref) / scale) < 2e-9
    one = dev.triple_product_sum(fields[keys[0]], fields[keys[1]], fields[keys[2]]).item()
    assert abs(one - ref[-1]) <= tol * scale[-1]


@pytest.mark.parametrize("n", [128, 256, 512, 1024])
def test_double_precision_fused_power_against_rocfft_route(hip, n):
    """ast_fft64_power_3d (hand-written double passes, binning fused into the x pass) against rocFFT R2C + ast_power_bin_1d
    on the same float64 grid, both shell rules; and against the numpy oracle at 256^3."""
    from astrild_amd import device as dev
    from oracle import fftpower as offt
    torch.cuda.set_device(0)
    if n == 1024:                                                   # generated on the device: 8.6 GB
        g = torch.Generator(device="cuda").manual_seed(n)
        t = torch.randn((n, n, n), dtype=torch.float64, device="cuda", generator=g) + 3.0
        f = None
    else:
        rng = np.random.default_rng(n)
        f = rng.standard_normal((n, n, n)) + 3.0
        t = dev.as_device(f)
    keep = t.clone()
    for rule in (("float64",) if n == 1024 else ("float64", "integer")):
        ks, ps, nm = dev.power_sums_fused64(t, 1000.0, binning=rule)
        spec = dev.r2c(t, engine="rocfft")
        ks2, ps2, nm2 = dev.power_bin_1d(spec, None, n, 1000.0, binning=rule)
        assert torch.equal(nm, nm2)
        np.testing.assert_allclose(ps.cpu().numpy(), ps2.cpu().numpy(), rtol=1e-11)
    assert torch.equal(t, keep)                                     # the grid is left intact
    res = dev.fftpower_1d(t, 1000.0)                                # default route for float64 cubes of this size
    if n <= 256:
        ref = offt.fftpower_1d(f, 1000.0)
        assert np.array_equal(res["modes"], ref["modes"])
        np.testing.assert_allclose(res["power"], ref["power"].real, rtol=1e-9)


def test_config_a_size_runs_on_the_hand_written_passes(hip):
    """128^3 (BASELINE config A's own size): float64 grids through ast_fft64_power_3d, fp32 grids through the same double
    passes widened on load (ast_fft64_power_3d_f32) - both against the oracle; the fp32 pipeline (paint rho - mean, double
    transform) within 1e-6 of the oracle on the same particles."""
    from astrild_amd import device as dev
    from oracle import fftpower as offt, mesh as omesh
    torch.cuda.set_device(0)
    n, L = 128, 1000.0
    assert hip.ast_fft64_supported(128) and hip.ast_fft64_supported(2048) and not hip.ast_fft64_supported(4096)
    rng = np.random.default_rng(128)
    f32 = (rng.standard_normal((n, n, n)) + 3.0).astype(np.float32)
    ref = offt.fftpower_1d(f32.astype(np.float64), L)
    for rule in ("float64", "integer"):
        got = dev.fftpower_1d(dev.as_device(f32), L, binning=rule)            # fp32 grid, transformed in double
        want = ref if rule == "float64" else offt.fftpower_1d(f32.astype(np.float64), L, binning="integer")
        assert np.array_equal(got["modes"], want["modes"])
        np.testing.assert_allclose(got["power"], want["power"].real, rtol=1e-10)
    pos = omesh.lattice_particles(n, n, L, seed=20240601)
    oracle = offt.fftpower_1d(omesh.paint(pos, None, n, L, "cic"), L)
    for dt, rtol in ((np.float64, 1e-9), (np.float32, 1e-6)):
        res = dev.paint_power_1d(dev.as_device(pos.astype(dt)), None, n, L, "cic")
        assert np.array_equal(res["modes"], oracle["modes"])
        if dt == np.float32:        # the same fp32-rounded positions through the oracle
            oracle32 = offt.fftpower_1d(omesh.paint(pos.astype(np.float32).astype(np.float64), None, n, L, "cic"), L)
            np.testing.assert_allclose(res["power"], oracle32["power"].real, rtol=rtol)
        else:
            np.testing.assert_allclose(res["power"], oracle["power"].real, rtol=rtol)


@pytest.mark.parametrize("window", ["cic", "tsc"])
def test_double_precision_pipeline_with_deferred_fold(hip, window):
    """paint(defer_fold) + ast_fft64_power_3d_halo (the halo records folded by the z pass) == paint + FFTPower on the
    folded grid, for float64 particles (the reference's dtype)."""
    from astrild_amd import device as dev
    torch.cuda.set_device(0)
    n, L = 256, 1000.0
    pos = dev.synth_lattice_particles(n, n, L, seed=3, dtype=torch.float64)
    got = dev.paint_power_1d(pos, None, n, L, window)             # default: the halo fold inside the z pass
    plain = dev.paint_power_1d(pos, None, n, L, window, defer_fold64=False)     # folded grid, then the fused passes
    np.testing.assert_allclose(plain["power"], got["power"], rtol=1e-12)
    grid = dev.paint(pos, None, n, L, window)
    ref = dev.fftpower_1d(grid, L, fused=False)
    assert np.array_equal(got["modes"], ref["modes"])
    np.testing.assert_allclose(got["power"], ref["power"], rtol=1e-10)


@pytest.mark.parametrize("n", [128, 256, 512])
def test_double_precision_r2c_against_rocfft_and_numpy(hip, n):
    """ast_fft64_r2c_3d (device.r2c's route for float64 cubes) against the rocFFT plan; against numpy at 256^3."""
    from astrild_amd import device as dev
    torch.cuda.set_device(0)
    rng = np.random.default_rng(n + 1)
    f = rng.standard_normal((n, n, n))
    t = dev.as_device(f)
    got = dev.r2c(t)
    ref = dev.r2c(t, engine="rocfft")
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 1e-13 * scale * np.sqrt(n)
    if n <= 256:
        want = np.fft.rfftn(f) / n ** 3
        assert np.abs(got.cpu().numpy() - want).max() < 1e-13 * np.abs(want).max() * np.sqrt(n)


# ------------------------------------------------------------------ disc layout (the slab transpose's wire format)
def _disc(hip, n, parts):
    from astrild_amd import slab
    return slab.disc_layout(n, parts, 32 if n == 1024 else 16, 16)


@pytest.mark.parametrize("n,parts,nplanes", [(256, 1, 3), (256, 4, 5), (512, 8, 2), (1024, 8, 2)])
def test_ky_pass_storing_in_the_disc_layout(dev, hip, n, parts, nplanes):
    """ast_fft_tile_c2c_disc against numpy: every element of every part's planes (only rows inside the Nyquist disc exist),
    the rank's own part written to its own place, the rest of the send buffer left alone."""
    from astrild_amd import _lib
    lay = _disc(hip, n, parts)
    nz, pitch = n // 2 + 1, (n // 2 + 1 + 15) // 16 * 16
    rng = np.random.default_rng(n + parts)
    a = (rng.standard_normal((nplanes, n, pitch)) + 1j * rng.standard_normal((nplanes, n, pitch))).astype(np.complex64)
    t = dev.as_device(a.copy())
    me = parts - 1
    packed = torch.full((nplanes * lay["total"],), float("nan"), dtype=torch.complex64, device="cuda")
    mine = torch.full((nplanes, lay["S"][me]), float("nan"), dtype=torch.complex64, device="cuda")
    _lib.check(hip.ast_fft_tile_c2c_disc(dev.ptr(t), dev.ptr(packed), 0, n, pitch, nplanes, parts, me, dev.ptr(mine), 0.5, dev.stream()))
    ref = 0.5 * np.fft.fft(a.astype(np.complex128), axis=1)
    rms = np.sqrt(np.mean(np.abs(ref[:, :, :nz]) ** 2))
    got_all = packed.cpu().numpy()
    for q in range(parts):
        rows, cols = lay["rows"][q], lay["cols"][q]
        if q == me:
            got = mine.cpu().numpy()
            assert np.isnan(got_all[nplanes * lay["cumS"][q]: nplanes * (lay["cumS"][q] + lay["S"][q])].real).all()      # its slot in the send buffer stays unwritten
        else:
            got = got_all[nplanes * lay["cumS"][q]: nplanes * (lay["cumS"][q] + lay["S"][q])].reshape(nplanes, -1)
        ok = cols < nz                                     # (columns past n/2 of the last tile: copies, never read)
        np.testing.assert_allclose(got[:, ok], ref[:, rows[ok], cols[ok]], rtol=0, atol=2e-6 * rms)
    assert torch.equal(t.cpu(), torch.from_numpy(a))       # the planes are left intact


@pytest.mark.parametrize("n,parts", [(256, 4), (512, 8)])
def test_last_pass_over_disc_blocks_adds_up_to_the_single_gpu_sums(dev, hip, n, parts):
    """One GPU plays every rank: z rows + k_y pass of ALL planes in the disc layout gives each part's whole block; the
    last pass + binning of the parts' blocks adds up to the shell sums of the single-GPU pipeline (same modes, same
    arithmetic per mode; the order of the additions differs)."""
    from astrild_amd import _lib
    lay = _disc(hip, n, parts)
    g = torch.Generator(device="cuda").manual_seed(n + 11)
    grid = torch.randn((n, n, n), dtype=torch.float32, device="cuda", generator=g)
    nz, pitch = n // 2 + 1, (n // 2 + 1 + 15) // 16 * 16
    spec = torch.empty((n, n, pitch), dtype=torch.complex64, device="cuda")
    _lib.check(hip.ast_fft_tile_rows_r2c(dev.ptr(grid), dev.ptr(spec), 0, n, n * n, n, pitch, 1.0, dev.stream()))
    packed = torch.empty((n * lay["total"],), dtype=torch.complex64, device="cuda")
    _lib.check(hip.ast_fft_tile_c2c_disc(dev.ptr(spec), dev.ptr(packed), 0, n, pitch, n, parts, -1, None, 1.0, dev.stream()))
    scratch = torch.empty(int(hip.ast_fft_tile_disc_power_scratch_bytes(n, parts)), dtype=torch.uint8, device="cuda")
    for rule in ("float64", "integer"):
        total = torch.zeros(n // 2 - 1, dtype=torch.float64, device="cuda")
        for q in range(parts):
            block = packed[n * lay["cumS"][q]: n * (lay["cumS"][q] + lay["S"][q])].clone()
            psum = torch.zeros(n // 2 - 1, dtype=torch.float64, device="cuda")
            _lib.check(hip.ast_fft_tile_disc_block_power(dev.ptr(block), dev.ptr(scratch), scratch.numel(), 0, n, parts, q,
                                                         1.0 / float(n) ** 3, 1000.0, 0, _lib.BIN[rule], dev.ptr(psum), dev.stream()))
            assert float(psum.sum()) > 0.0
            total += psum
        _, ref, _ = dev.power_sums_fused(grid, 1000.0, lowk=False, binning=rule)
        np.testing.assert_allclose(total.cpu().numpy(), ref.cpu().numpy(), rtol=1e-12)


@pytest.mark.parametrize("n", [256, 512])
def test_single_gpu_pipeline_through_the_disc_layout_changes_no_bit(dev, n):
    """AST_FFT_DISC=1: the power pipeline's k_y pass stores in the disc layout (one part) and the binning pass reads it
    from there - the same partial sums in the same order as the in-place passes."""
    import os
    g = torch.Generator(device="cuda").manual_seed(n + 13)
    t = torch.randn((n, n, n), dtype=torch.float32, device="cuda", generator=g)
    _, ref, _ = dev.power_sums_fused(t, 750.0)
    for mode in ("1", "2"):                  # 2: x-major tiles (the last pass reads N * 128 contiguous bytes per workgroup)
        os.environ["AST_FFT_DISC"] = mode
        try:
            _, got, _ = dev.power_sums_fused(t, 750.0)
        finally:
            del os.environ["AST_FFT_DISC"]
        assert torch.equal(got, ref), mode
