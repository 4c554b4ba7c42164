"""GPU: analytic NFW halo painting against the reference's own
known-answer values and the oracle's numpy restatement."""
import json
import os
import types

import numpy as np
import numpy.testing as npt
import pandas as pd
import pytest

from oracle import kappa as ok

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


@pytest.fixture(scope="module", autouse=True)
def _dev(hip):
    torch.cuda.set_device(0)


def _halo():
    g = GOLD["nfw_halo"]
    return {k: np.array(v) for k, v in g["halo"].items()}, g


@pytest.mark.parametrize("signal", ["dT", "alpha"])
def test_reference_known_answers(signal):
    from astrild_amd.rays.skys import SkyUtils
    halo, g = _halo()
    m = SkyUtils.analytic_Halo_signal_to_SkyArray(np.array([0]), halo, g["extent"], g["direction"], g["suppress"],
                                                  g["suppression_R"], g["npix"], signal)
    e = g[signal]
    assert np.unravel_index(m.argmax(), m.shape) == tuple(e["argmax"])
    npt.assert_almost_equal(m.min(), e["min"], decimal=e["decimals"]["min"])
    npt.assert_almost_equal(m.mean(), e["mean"], decimal=e["decimals"]["mean"])
    npt.assert_almost_equal(m.max(), e["max"], decimal=e["decimals"]["max"])
    ref = ok.analytic_halo_signal_map(halo, g["extent"], g["direction"], g["suppress"], g["suppression_R"],
                                      g["npix"], signal)
    npt.assert_allclose(m, ref, rtol=0, atol=1e-9 * abs(ref).max())


def test_catalogue_of_overlapping_and_clipped_halos():
    from astrild_amd.rays.skys import SkyArray
    rng = np.random.default_rng(0)
    nh, npix = 40, 512
    cat = pd.DataFrame({
        "r200_deg": rng.uniform(0.02, 0.08, nh), "r200_pix": rng.integers(4, 20, nh).astype(float),
        "m200": 10 ** rng.uniform(13, 14.5, nh), "c_NFW": rng.uniform(2, 8, nh), "Dc": rng.uniform(500, 2000, nh),
        "theta1_pix": rng.integers(-10, npix + 10, nh), "theta2_pix": rng.integers(-10, npix + 10, nh),
        "theta1_tv": rng.normal(0, 300, nh), "theta2_tv": rng.normal(0, 300, nh)})
    for to, direction in (("dT", [0, 1]), ("dT", [1]), ("alpha", [0]), ("alpha", [1])):
        sky = SkyArray.from_halo_dataframe(cat, npix=npix, extent=3, direction=direction, suppress=True,
                                           suppression_R=2, opening_angle=10.0, to=to)
        halo = {k: cat[k].values for k in cat.columns}
        ref = ok.analytic_halo_signal_map(halo, 3, direction, True, 2, npix, to)
        npt.assert_allclose(sky.data["orig"], ref, rtol=0, atol=1e-9 * abs(ref).max())
        assert sky.npix == npix and sky.opening_angle == 10.0
    assert SkyArray.from_halo_dataframe(cat, npix=npix, extent=3, direction=[0], to="alpha").quantity == "alpha_x"
    # the Rees-Sciama map constructor (sky_array.py:341-400): the dT stamps under the isw_rs label
    for direction, label in (([0, 1], "isw_rs"), ([0], "isw_rs_x"), ([1], "isw_rs_y")):
        sky = SkyArray.from_halo_catalogue_to_temperature_perturbation_map(cat, extent=3, direction=direction, suppress=True,
                                                                            suppression_R=2, npix=npix, opening_angle=10.0)
        ref = ok.analytic_halo_signal_map({k: cat[k].values for k in cat.columns}, 3, direction, True, 2, npix, "dT")
        ref[np.isinf(ref)] = 0.0
        npt.assert_allclose(sky.data["orig"], ref, rtol=0, atol=1e-9 * abs(ref).max())
        assert sky.quantity == label and sky.opening_angle == 10.0


def test_single_stamp_maps_and_halo_series():
    from astrild_amd.rays.skys import SkyArray, SkyUtils
    a = SkyUtils.NFW_deflection_angle_map(0.05, 7e13, 2.0, 1050 * 0.6774, npix=201, extent=5, direction=[1],
                                          suppress=True, suppression_R=3)
    ref = ok.nfw_deflection_angle_map(0.05, 7e13, 2.0, 1050 * 0.6774, 201, 5, [1], True, 3)
    npt.assert_allclose(a, ref, rtol=0, atol=1e-9 * abs(ref).max())
    t = SkyUtils.NFW_temperature_perturbation_map(0.05, 7e13, 2.0, [150.0, -80.0], 1050 * 0.6774, npix=101, extent=2)
    reft = ok.nfw_temperature_perturbation_map(0.05, 7e13, 2.0, [150.0, -80.0], 1050 * 0.6774, 101, 2, [0, 1])
    npt.assert_allclose(t, reft, rtol=0, atol=1e-9 * abs(reft).max())
    halo = types.SimpleNamespace(r200_deg=0.05, m200=7e13, c_NFW=2.0, Dc=700.0, theta1_tv=100.0, theta2_tv=50.0)
    sky = SkyArray.from_halo_series(halo, npix=101, extent=2, direction=[0], to="dT")
    assert sky.quantity == "rs_x" and sky.data["orig"].shape == (101, 101)
    with pytest.raises(AssertionError):
        SkyUtils.NFW_deflection_angle_map(0.05, 7e13, 2.0, 700.0, npix=101, direction=[1, 1])


def test_add_patch_to_map_clipping():
    from astrild_amd.rays.skys import SkyUtils
    rng = np.random.default_rng(1)
    big = rng.standard_normal((50, 50))
    small = rng.standard_normal((21, 21))
    for cen in ((25, 25), (3, 48), (-5, 10), (49, 0)):
        got = SkyUtils.add_patch_to_map(big.copy(), small, cen)
        ref = ok.add_patch_to_map(big.copy(), small, cen)
        assert np.array_equal(got, ref)
