"""The single-precision three-stage passes for cubes of side 2048 (csrc/lens_fft.hip, ast_fft32_big_power_3d): the same
kernels at side 256 against the CPU oracle and the fp32 tile pipeline; side 2048 itself against the double-precision
passes in tests/test_gpu_fullsize.py.  Tolerance: north_star's 1e-6 relative on every shell; mode counts exact."""
import numpy as np
import pytest

from oracle import fftpower as offt

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev(hip):
    from astrild_amd import device
    torch.cuda.set_device(0)
    return device


@pytest.mark.parametrize("binning", [None, "integer"])
def test_big32_passes_at_256_against_the_oracle(dev, binning):
    from astrild_amd._lib import lib
    n, L = 256, 700.0
    assert lib().ast_fft32_big_supported(n) and lib().ast_fft32_big_supported(2048) and not lib().ast_fft32_big_supported(512)
    rng = np.random.default_rng(17)
    # an O(1) mean under a field with structure on all scales: what a painted density looks like to an fp32 transform
    host = (1.0 + 0.3 * rng.standard_normal((n, n, n)) + 0.2 * np.sin(2 * np.pi * np.arange(n) * 3 / n)[None, None, :]).astype(np.float32)
    field = dev.as_device(host)
    mean = float(host.astype(np.float64).mean())
    got = dev.finish_power(*dev.power_sums_fused64(field, L, binning=binning, mean=mean))
    ref = offt.fftpower_1d(host.astype(np.float64), L, binning=binning)
    assert np.array_equal(got["modes"], ref["modes"])
    np.testing.assert_allclose(got["k"], ref["k"], rtol=1e-12)
    np.testing.assert_allclose(got["power"], ref["power"].real, rtol=1e-6)
    # and the fp32 tile pipeline on the same grid (its own low-k channel): the two single-precision routes agree as closely
    tile = dev.finish_power(*dev.power_sums_fused(field, L, mean=mean, binning=binning))
    np.testing.assert_allclose(got["power"], tile["power"], rtol=1e-6)
    # without pruning: bit-identical shell sums (what FFTPower drops never reaches a shell)
    import os
    psum = dev.power_sums_fused64(field, L, binning=binning, mean=mean)[1].clone()
    os.environ["AST_FFT_NO_PRUNE"] = "1"
    try:
        full = dev.power_sums_fused64(field, L, binning=binning, mean=mean)[1]
    finally:
        del os.environ["AST_FFT_NO_PRUNE"]
    np.testing.assert_allclose(psum.cpu().numpy(), full.cpu().numpy(), rtol=1e-12)


def test_big32_rejects_what_it_does_not_cover(dev):
    from astrild_amd._lib import lib
    g = torch.zeros((64, 64, 64), dtype=torch.float32, device="cuda")
    ps = torch.zeros(31, dtype=torch.float64, device="cuda")
    scratch = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    assert lib().ast_fft32_big_power_3d(dev.ptr(g), dev.ptr(scratch), scratch.numel(), 64, 100.0, 0, 0.0, dev.ptr(ps), dev.stream()) != 0
    g = torch.zeros((256, 256, 256), dtype=torch.float32, device="cuda")
    ps = torch.zeros(127, dtype=torch.float64, device="cuda")
    assert lib().ast_fft32_big_power_3d(dev.ptr(g), dev.ptr(scratch), scratch.numel(), 256, 100.0, 0, 0.0, dev.ptr(ps), dev.stream()) != 0   # scratch too small


@pytest.mark.parametrize("window", ["cic", "tsc"])
def test_big32_fold_on_load_equals_the_folded_grid(dev, window):
    """A grid painted with defer_fold=True: its halo records are folded as the z rows - and the double-precision low-k sums -
    load the border rows (ast_fft32_big_power_3d_halo), in the fold kernel's own order: the same shell sums as over the grid
    the paint folded itself."""
    n, L = 256, 1000.0
    pos = dev.synth_lattice_particles(n, n, L, seed=9, dtype=torch.float32)
    folded = dev.paint(pos, None, n, L, window, method="tiled", offset="mean", check_dropped=False)
    ref = dev.power_sums_fused64(folded, L, mean=0.0)[1].clone()
    grid, halo = dev.paint(pos, None, n, L, window, method="tiled", offset="mean", defer_fold=True, check_dropped=False)
    assert not torch.equal(grid, folded)                    # the deferred grid alone is incomplete
    got = dev.power_sums_fused64(grid, L, mean=0.0, halo=halo)[1]
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-12)
