"""GPU: flat-sky window filters against the reference's known-answer values, the
oracle's numpy/scipy restatement, and through SkyArray.filter's by-name dispatch."""
import json
import os

import numpy as np
import numpy.testing as npt
import pytest

from oracle import kappa as ok

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))
RTOL = 1e-12        # fp64 stencils / sums in a different association order than numpy's at most


@pytest.fixture(scope="module", autouse=True)
def _dev(hip):
    torch.cuda.set_device(0)


@pytest.fixture(scope="module")
def dt_map():
    g = GOLD["nfw_halo"]
    halo = {k: np.array(v) for k, v in g["halo"].items()}
    return ok.analytic_halo_signal_map(halo, g["extent"], g["direction"], g["suppress"], g["suppression_R"],
                                       g["npix"], "dT")


def _close(a, b, rtol=RTOL):
    npt.assert_allclose(a, b, rtol=rtol, atol=rtol * np.abs(b).max())


def test_dgd3_reference_known_answers(dt_map):
    from astrild_amd.rays.utils import Filters
    g = GOLD["dgd3"]
    for case in g["cases"]:
        f = Filters.gaussian_third_derivative(dt_map, g["theta_deg"], g["theta_i_deg"], case["direction"])
        x_slice, y_slice = f[:, len(f) // 2], f[len(f) // 2, :]
        assert x_slice.max() == case["x_slice_max"]
        npt.assert_almost_equal(y_slice.max() * 1e7, case["y_slice_max_times_1e7"], decimal=case["decimal"])
        _close(f, ok.dgd_filter(dt_map, g["theta_deg"], g["theta_i_deg"], case["direction"], 3))


@pytest.mark.parametrize("npix", [33, 100, 257])
@pytest.mark.parametrize("order", [1, 3])
@pytest.mark.parametrize("direction", [0, 1])
def test_dgd_windows_match_oracle(npix, order, direction):
    from astrild_amd.rays.utils import Filters
    img = np.random.default_rng(npix + order).standard_normal((npix, npix))
    fct = Filters.gaussian_third_derivative if order == 3 else Filters.gaussian_first_derivative
    got = fct(img, 2.0, 0.13, direction)
    _close(got, ok.dgd_filter(img, 2.0, 0.13, direction, order))


@pytest.mark.parametrize("npix", [2, 17, 128])
def test_apodization_matches_oracle(npix):
    from astrild_amd.rays.utils import Filters
    img = np.random.default_rng(npix).standard_normal((npix, npix))
    _close(Filters.apodization(img, 1.0), ok.apodization(img), rtol=1e-13)


@pytest.mark.parametrize("direction", [1, [0, 1], [1, 0]])
@pytest.mark.parametrize("npix", [40, 101])
def test_dgd3_convolution_matches_scipy(npix, direction):
    from astrild_amd.rays.utils import Filters
    img = np.random.default_rng(npix).standard_normal((npix, npix))
    d = direction if isinstance(direction, int) else np.asarray(direction)
    got = Filters.gaussian_third_derivative_convolution(img, 1.0, 0.03, d)
    _close(got, ok.dgd3_convolution(img, 1.0, 0.03, d), rtol=1e-11)


@pytest.mark.parametrize("npix,theta_o", [(64, 0.11), (100, 0.3), (50, 0.9)])
def test_gaussian_compensated_matches_scipy(npix, theta_o):
    from astrild_amd.rays.utils import Filters
    img = np.random.default_rng(npix).standard_normal((npix, npix))
    got = Filters.gaussian_compensated(img, 1.0, theta_o / 3, theta_o)
    _close(got, ok.gaussian_compensated(img, 1.0, theta_o / 3, theta_o))


def test_convolve2d_general_window_bit_exact_small():
    """Odd/even, non-square window; scipy adds taps in the same row-major order -> identical bits."""
    from scipy import ndimage
    from astrild_amd import _lib
    from astrild_amd.device import as_device, ptr, stream
    rng = np.random.default_rng(3)
    img = rng.standard_normal((37, 37))
    for kh, kw in [(1, 1), (3, 4), (6, 5), (8, 8)]:
        w = rng.standard_normal((kh, kw))
        t, wd = as_device(img), as_device(w)
        out = torch.empty_like(t)
        _lib.check(_lib.lib().ast_convolve2d(ptr(t), ptr(wd), ptr(out), 37, kh, kw, stream()), "ast_convolve2d")
        ref = ndimage.convolve(img, w)
        _close(out.cpu().numpy(), ref, rtol=1e-14)


@pytest.mark.parametrize("npix", [64, 301])
def test_aperture_photometry_matches_oracle(npix):
    from astrild_amd.rays.utils import Filters
    img = np.random.default_rng(npix).standard_normal((npix, npix)) + 3.0
    ref = ok.aperture_photometry(img, 1.0, 0.2)
    work = img.copy()
    got = Filters.aperture_photometry(work, 1.0, 0.2)
    _close(got, ref)
    assert got is work                      # in place, like the reference


def test_skyarray_filter_dispatches_new_filters(dt_map):
    from astrild_amd.rays.skys import SkyArray
    sky = SkyArray.from_array(dt_map.copy(), opening_angle=1.0, quantity="isw_rs", dir_in="")
    out = sky.filter({"gaussian_third_derivative": {"theta_i": 0.05, "direction": 1, "abbrev": "dgd3"}},
                     on="orig", rtn=True)
    _close(out, ok.dgd_filter(dt_map, 1.0, 0.05, 1, 3))


@pytest.mark.parametrize("nin,npix", [(100, 50), (256, 100), (777, 100), (64, 64), (1024, 128), (333, 332)])
def test_resize_antialiased_against_scipy_restatement(nin, npix):
    """SkyArray.resize (sky_array.py:475-496, skimage.transform.resize(.., anti_aliasing=True)): Gaussian prefilter
    (reflect, truncate 4, sigma = (nin / npix - 1) / 2) + bilinear zoom at pixel centres, on the device, against the
    oracle's scipy.ndimage restatement - integer and ragged factors, no-op size, a factor so close to one that the
    prefilter's kernel is a single tap."""
    from astrild_amd import lensing
    rng = np.random.default_rng(nin * 1000 + npix)
    img = rng.standard_normal((nin, nin)) * 0.02 + np.linspace(0.0, 1.0, nin)[None, :]
    keep = img.copy()
    want = ok.resize_antialiased(img, npix)
    got = lensing.resize_antialiased(img, npix).cpu().numpy()
    assert got.shape == (npix, npix)
    npt.assert_allclose(got, want, rtol=0, atol=1e-13 * np.abs(want).max())
    assert np.array_equal(img, keep)                         # the input is left alone
    t = torch.as_tensor(img, device="cuda")
    got_t = lensing.resize_antialiased(t, npix).cpu().numpy()
    assert np.array_equal(got_t, got) and np.array_equal(t.cpu().numpy(), keep)
    with pytest.raises(NotImplementedError):
        lensing.resize_antialiased(img, nin + 1)


def test_resize_border_pixels_against_hand_computed_mirror_padding():
    """the device prefilter uses ndimage's "mirror" boundary (what skimage.transform.resize's default
    mode="reflect" means); checked against a convolution written out over numpy.pad(mode="reflect")."""
    from astrild_amd import lensing
    from tests.test_oracle_kappa import mirror_resize_by_hand
    rng = np.random.default_rng(12)
    for nin, npix in ((96, 24), (100, 40), (64, 7)):
        m = rng.standard_normal((nin, nin)) + np.linspace(0, 3, nin)[:, None]
        got = lensing.resize_antialiased(m, npix).cpu().numpy()
        npt.assert_allclose(got, mirror_resize_by_hand(m, npix), rtol=0, atol=1e-13 * np.abs(m).max())


def test_skyarray_resize_of_and_img():
    from astrild_amd.rays import SkyMap
    rng = np.random.default_rng(9)
    m = rng.standard_normal((128, 128))
    sky = SkyMap.from_array(m.copy(), 128, 10.0, "kappa_2", "/tmp/")
    want = ok.resize_antialiased(m, 32)
    r = sky.resize(32, img=m.copy(), rtn=True)
    npt.assert_allclose(r, want, rtol=0, atol=1e-13 * np.abs(want).max())
    sky.resize(32, of="orig")                                # stored map, in place of the original
    assert sky.data["orig"].shape == (32, 32)
    npt.assert_allclose(sky.data["orig"], want, rtol=0, atol=1e-13 * np.abs(want).max())
