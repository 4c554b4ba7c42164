"""SkyArray keeps what its methods produce in HBM (rays/_resident.MapStore) until an array is asked for: a chain of methods
gives bit for bit what the same chain gives when every intermediate map is fetched (the reference's shape: numpy arrays in
``self.data`` between the methods, sky_array.py:119-131), and the dict still behaves like the reference's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def kappa(hip):
    torch.cuda.set_device(0)
    rng = np.random.default_rng(77)
    return rng.normal(0.0, 0.01, size=(512, 512))


def _chain(sky, fetch):
    """filter (Gaussian FFT smoothing, then the Hann window) -> shape noise -> kappa -> alpha -> pdf / peaks / C_l"""
    from astrild_amd.power_spectra.angular_power_spectrum import AngularPowerSpectrum
    look = (lambda name: sky.data[name]) if fetch else (lambda name: None)
    sky.filter({"gaussian": {"theta_i": 1.0, "abbrev": "g"}, "apodization": {"abbrev": "apo"}}, on="orig")
    look("orig_g_apo")
    sky.create_galaxy_shape_noise(0.3, 40.0, rnd_seed=5)
    sky.add_galaxy_shape_noise()
    sky.convert_convergence_to_deflection(on="orig_g_apo", rtn=False)
    look("defltx"), look("deflty")
    pdf = sky.pdf(50, of="orig_g_apo")
    peaks = sky.wl_peak_counts(20, "", of="orig_g_apo")
    sky.resize(256, of="orig_g_apo")
    look("orig_g_apo")
    return pdf, peaks


def test_resident_chain_equals_the_fetched_chain(kappa):
    from astrild_amd.rays.skys import SkyArray
    a = SkyArray.from_array(kappa.copy(), opening_angle=5.0, quantity="kappa_2", dir_in="")
    b = SkyArray.from_array(kappa.copy(), opening_angle=5.0, quantity="kappa_2", dir_in="")
    pdf_a, peaks_a = _chain(a, fetch=False)
    assert a.data.resident("orig_g_apo") and a.data.resident("defltx") and a.data.resident("deflty")
    assert not a.data.resident("orig") and not a.data.resident("orig_gsn")          # (add_galaxy_shape_noise returns the array)
    pdf_b, peaks_b = _chain(b, fetch=True)
    assert not b.data.resident("orig_g_apo") and not b.data.resident("defltx")
    assert sorted(a.data.keys()) == sorted(b.data.keys())
    for name in a.data.keys():
        got, ref = a.data[name], b.data[name]
        assert isinstance(got, np.ndarray) and got.dtype == ref.dtype and np.array_equal(got, ref), name
        assert not a.data.resident(name)                                              # handed out: the array is the map now
    assert np.array_equal(pdf_a["values"], pdf_b["values"]) and np.array_equal(pdf_a["bins"], pdf_b["bins"])
    assert peaks_a.equals(peaks_b)
    # an array that was handed out is THE map: a change made through it is what the next method sees
    a.data["orig_g_apo"][...] = 0.0
    assert float(np.abs(a.filter({"apodization": {"abbrev": "x"}}, on="orig_g_apo", rtn=True)).max()) == 0.0


def test_resident_chain_against_the_oracle(kappa):
    from oracle import kappa as ok
    from astrild_amd.rays.skys import SkyArray
    sky = SkyArray.from_array(kappa.copy(), opening_angle=5.0, quantity="kappa_2", dir_in="")
    sky.filter({"gaussian": {"theta_i": 1.0, "abbrev": "g"}}, on="orig")
    sky.convert_convergence_to_deflection(on="orig_g", rtn=False)
    assert sky.data.resident("orig_g") and sky.data.resident("defltx")
    ref = ok.gaussian_smooth(kappa, 5.0, 1.0)
    a1, a2 = ok.kappa0_to_alphas(ref, 512, np.deg2rad(5.0))
    np.testing.assert_allclose(sky.data["orig_g"], ref, rtol=0, atol=1e-13 * np.abs(ref).max())
    np.testing.assert_allclose(sky.data["deflty"], a1, rtol=0, atol=1e-12 * np.abs(a1).max())
    np.testing.assert_allclose(sky.data["defltx"], a2, rtol=0, atol=1e-12 * np.abs(a2).max())


def test_filters_take_and_return_device_maps(kappa):
    from astrild_amd.rays.utils.filters import Filters
    from astrild_amd.device import as_device
    t = as_device(kappa)
    keep = t.clone()
    for name, args in (("gaussian", dict(theta_i=1.0)), ("gaussian_high_pass", dict(theta_i=1.0)), ("apodization", {}),
                       ("gaussian_third_derivative", dict(theta_i=0.05, direction=1)),
                       ("gaussian_third_derivative_convolution", dict(theta_i=0.02, direction=1)),
                       ("gaussian_compensated", dict(theta_i=0.02, theta_o=0.04))):
        got = getattr(Filters, name)(t, 5.0, **args)
        ref = getattr(Filters, name)(kappa, 5.0, **args)
        assert isinstance(got, torch.Tensor) and got.is_cuda and isinstance(ref, np.ndarray), name
        assert torch.equal(t, keep), name                                             # the input map is left alone
        assert np.array_equal(got.cpu().numpy(), ref), name
    # aperture photometry updates its map in place, array or tensor (filters.py:40-73)
    host = kappa.copy()
    Filters.aperture_photometry(host, 5.0, 0.5)
    out = Filters.aperture_photometry(t, 5.0, 0.5)
    assert out is t and np.array_equal(t.cpu().numpy(), host)


def test_map_store_is_the_reference_dict():
    from astrild_amd.rays._resident import MapStore
    s = MapStore({"a": np.arange(4.0)})
    s["b"] = torch.arange(3.0, device="cuda", dtype=torch.float64)
    assert s.resident("b") and not s.resident("a")
    assert list(s.keys()) == ["a", "b"] and "b" in s and len(s) == 2
    assert s.device("b").is_cuda and s.resident("b")                                  # the methods' access keeps it in HBM
    assert isinstance(s["b"], np.ndarray) and not s.resident("b")                     # the caller's access fetches it, once
    assert s["b"] is s["b"] and [k for k, _ in s.items()] == ["a", "b"] and len(s.values()) == 2
    assert s.get("zz") is None and np.array_equal(s.pop("a"), np.arange(4.0)) and "a" not in s
    import copy
    s["c"] = torch.ones(2, device="cuda", dtype=torch.float64)
    d = copy.deepcopy(s)
    assert d.resident("c") and np.array_equal(d["c"], np.ones(2)) and s.resident("c")


def test_upload_planes_one_allocation_and_fallbacks(hip):
    """device.upload_planes: equal-sized host arrays become views into ONE device allocation (plane after plane, float64);
    mixed sizes, a single array and an empty list fall back to one allocation each; values and order are kept, and the
    stack over the views equals the stack over separately uploaded planes bit for bit."""
    from astrild_amd import device as dev, lensing
    torch.cuda.set_device(0)
    rng = np.random.default_rng(3)
    arrays = [rng.normal(size=(64, 64)) for _ in range(5)] + [rng.normal(size=(64, 64)).astype(np.float32)]
    planes = dev.upload_planes(arrays)
    assert len(planes) == 6 and all(p.is_cuda and p.dtype == torch.float64 and p.shape == (64 * 64,) for p in planes)
    assert all(planes[i + 1].data_ptr() - planes[i].data_ptr() == 64 * 64 * 8 for i in range(5))          # one slab
    for p, a in zip(planes, arrays):
        assert np.array_equal(p.cpu().numpy(), np.asarray(a, dtype=np.float64).reshape(-1))
    shaped = dev.upload_planes(arrays, flat=False)
    assert all(p.shape == (64, 64) for p in shaped)
    sep = [dev.as_device(np.asarray(a, dtype=np.float64)) for a in arrays]
    assert torch.equal(lensing.kappa_stack(shaped), lensing.kappa_stack(sep))
    mixed = dev.upload_planes([np.ones((4, 4)), np.ones((2, 2))])
    assert [p.numel() for p in mixed] == [16, 4]
    assert dev.upload_planes([]) == [] and dev.upload_planes([np.ones(3)])[0].shape == (3,)
    big = dev.to_numpy(torch.arange(1 << 18, dtype=torch.float64, device="cuda"))                           # >= 1 MiB: page-locked
    small = dev.to_numpy(torch.arange(8, dtype=torch.float64, device="cuda"))
    assert np.array_equal(big, np.arange(1 << 18)) and np.array_equal(small, np.arange(8)) and big.flags.writeable
