"""GPU: the astrild-named Python API (PowerSpectrum3D, Bispectrum3D, SubFind, SkyMap /
SkyArray / SkyUtils / Filters, RayRamses, SimulationCollection) against the oracle."""
import os
import types

import numpy as np
import numpy.testing as npt
import pandas as pd
import pytest

from oracle import bispectrum as ob, fftpower as offt, kappa as ok, mesh as omesh

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class FakeSimulation:
    """The attributes PowerSpectrum3D touches on astrild.simulation.Simulation."""

    def __init__(self, tmp, n, boxsize, files):
        self.boxsize, self.domain_level, self.npar = boxsize, n, n
        self.dirs = {"out": str(tmp) + "/"}
        self._files = files
        self.dir_nrs = sorted(files)

    def get_file_nrs(self, file_dsc, path, which):
        return sorted(self._files)

    def get_file_paths(self, file_dsc, path, which):
        return [self._files[k] for k in sorted(self._files)]


@pytest.fixture(scope="module", autouse=True)
def _dev(hip):
    torch.cuda.set_device(0)


def test_power_spectrum_3d_compute_auto_and_cross(tmp_path):
    from astrild_amd.power_spectra import PowerSpectrum3D
    rng = np.random.default_rng(0)
    n, L = 32, 250.0
    grids = {3: rng.standard_normal((n, n, n)) + 2.0, 7: rng.standard_normal((n, n, n))}
    files = {}
    for nr, g in grids.items():
        files[nr] = str(tmp_path / f"grid_{nr:03d}.npy")
        np.save(files[nr], g)
    sim = FakeSimulation(tmp_path, n, L, files)
    ps = PowerSpectrum3D("particles", sim)
    pk = ps.compute(["rho"], [{"path": "x", "root": "grid", "extension": "npy"}], save=False)
    for nr, g in grids.items():
        k, p = offt.power_spectrum_3d(g, L)
        npt.assert_allclose(pk["k"][f"snap_{nr}"], k, rtol=1e-12)
        npt.assert_allclose(pk["P"][f"snap_{nr}"], p, rtol=1e-10)
    # cross spectrum branch (power_spectrum_3d.py:197-222)
    k, p = ps._power_spectrum_3d(grids[3], grids[7])
    kr, pr = offt.power_spectrum_3d(grids[3], L, grids[7])
    npt.assert_allclose(p, pr, rtol=1e-9, atol=1e-12 * abs(pr).max())
    with pytest.raises(BaseException):
        ps._power_spectrum_3d(np.zeros((8, 8, 8)))


def test_snapshot_loop_is_double_buffered_and_keeps_every_snapshot_apart(tmp_path):
    """The snapshot loop (power_spectrum_3d.py:83-110) with a loader thread one snapshot ahead and rotating page-locked
    staging buffers: five snapshots (more than buffers), auto and cross - every spectrum is its own snapshot's."""
    from astrild_amd.power_spectra import PowerSpectrum3D
    rng = np.random.default_rng(3)
    n, L = 48, 250.0
    grids = {nr: rng.standard_normal((n, n, n)) + 0.1 * nr for nr in (2, 4, 5, 8, 9)}
    files = {}
    for nr, g in grids.items():
        files[nr] = str(tmp_path / f"grid_{nr:03d}.npy")
        np.save(files[nr], g)
    ps = PowerSpectrum3D("particles", FakeSimulation(tmp_path, n, L, files))
    pk = ps.compute(["rho"], [{"path": "x", "root": "grid", "extension": "npy"}], save=False)
    assert list(pk["P"]) == [f"snap_{nr}" for nr in sorted(grids)]
    for nr, g in grids.items():
        k, p = offt.power_spectrum_3d(g, L)
        npt.assert_allclose(pk["P"][f"snap_{nr}"], p, rtol=1e-10)
    nrs = sorted(grids)
    other = nrs[1:] + nrs[:1]
    cross = ps._cross_power_spectra(None, nrs, [files[a] for a in nrs], [files[b] for b in other])
    for a, b in zip(nrs, other):
        kr, pr = offt.power_spectrum_3d(grids[a], L, grids[b])
        npt.assert_allclose(cross["P"][f"snap_{a}"], pr, rtol=1e-9, atol=1e-12 * abs(pr).max())


def test_power_spectrum_3d_read_data_ngp_assign(tmp_path, monkeypatch):
    from astrild_amd.power_spectra import PowerSpectrum3D, power_spectrum_3d as mod
    rng = np.random.default_rng(1)
    n = 16
    df = pd.DataFrame({"x": rng.uniform(0, 1, 9000), "y": rng.uniform(0, 1, 9000), "z": rng.uniform(0, 1, 9000),
                       "rho": rng.standard_normal(9000)})
    monkeypatch.setattr(mod.pd, "read_hdf", lambda path, key=None: df)       # PyTables is not installed here
    sim = FakeSimulation(tmp_path, n, 100.0, {})
    grid = PowerSpectrum3D("particles", sim)._read_data("fields.h5", "rho")
    ref = omesh.ngp_assign(df.x.values, df.y.values, df.z.values, df.rho.values, n)
    assert np.array_equal(grid.cpu().numpy(), ref)


def test_power_spectrum_3d_fp32_mode(tmp_path):
    from astrild_amd.power_spectra import PowerSpectrum3D
    rng = np.random.default_rng(2)
    n, L = 64, 500.0
    g = (1.0 + 0.5 * rng.standard_normal((n, n, n))).astype(np.float32)
    ps = PowerSpectrum3D("particles", FakeSimulation(tmp_path, n, L, {}))
    ps.dtype = torch.float32
    k, p = ps._power_spectrum_3d(g)
    kr, pr = offt.power_spectrum_3d(g, L)
    npt.assert_allclose(p, pr, rtol=1e-6)


def test_subfind_power_spectrum_tsc():
    from astrild_amd.particles.hutils import SubFind
    rng = np.random.default_rng(3)
    nobj, nbins, box = 20000, 32, 62.0
    h = 0.6774
    snap = types.SimpleNamespace(
        cat={"SubhaloPos": rng.uniform(0, box * 1e3 / h, (nobj, 3)), "SubhaloMass": rng.uniform(1, 50, nobj)},
        header=types.SimpleNamespace(hubble=h, boxsize=box * 1e3),
    )
    k, p = SubFind.power_spectrum(snap, nbins=nbins, boxsize=box)
    pos = snap.cat["SubhaloPos"] * h / 1e3
    mass = snap.cat["SubhaloMass"] * h / 1e10
    grid = omesh.paint(pos, mass, nbins, box, "tsc") / (box / nbins) ** 3
    kr, pr = offt.power_spectrum_3d(grid, box)
    npt.assert_allclose(k, kr, rtol=1e-12)
    npt.assert_allclose(p, pr, rtol=1e-9)


def test_subfind_power_spectrum_clumpy_catalogue_on_the_tiles():
    """A halo-like catalogue (Gaussian clumps, masses over three decades) dense enough for the LDS tiles (24 objects per
    tile: the exact two-pass lists, `device.auto_paint_method`), through the API from raw catalogue units - the unit factors
    ride on the paint (stats_subfind.py:121-122 converts on the host first) - against the oracle on converted arrays."""
    from astrild_amd.particles.hutils import SubFind
    from astrild_amd import device as dev
    rng = np.random.default_rng(5)
    nobj, nbins, box, h = 200000, 256, 500.0, 0.6774
    assert dev.auto_paint_method(nobj, nbins, nbins, "tsc") == "tiled2"
    assert dev.auto_paint_method(nobj // 4, nbins, nbins, "tsc") == "direct" and dev.auto_paint_method(1 << 24, nbins, nbins, "cic") == "tiled"
    centres = rng.uniform(0.0, box, size=(512, 3))
    which = rng.integers(0, 512, size=nobj)
    pos = np.mod(centres[which] + rng.standard_normal((nobj, 3)) * rng.uniform(0.5, 4.0, size=512)[which][:, None], box)
    mass = 10.0 ** rng.uniform(0.0, 3.0, size=nobj)
    snap = types.SimpleNamespace(cat={"SubhaloPos": pos * 1e3 / h, "SubhaloMass": mass * 1e10 / h},
                                 header=types.SimpleNamespace(hubble=h, boxsize=box * 1e3))
    k, p = SubFind.power_spectrum(snap, nbins=nbins, boxsize=box)
    pos_c = snap.cat["SubhaloPos"] * h / 1e3
    mass_c = snap.cat["SubhaloMass"] * h / 1e10
    grid = omesh.paint(pos_c, mass_c, nbins, box, "tsc") / (box / nbins) ** 3
    kr, pr = offt.power_spectrum_3d(grid, box)
    npt.assert_allclose(k, kr, rtol=1e-12)
    npt.assert_allclose(p, pr, rtol=1e-9)


@pytest.mark.parametrize("n,width", [(16, 1), (32, 3)])
def test_bispectrum_3d_vs_oracle(tmp_path, n, width):
    from astrild_amd.bispectra import Bispectrum3D
    rng = np.random.default_rng(n)
    L = 80.0
    f = rng.standard_normal((n, n, n))
    f = f + 0.4 * f ** 2
    bs = Bispectrum3D("particles", FakeSimulation(tmp_path, n, L, {}))
    edges = ob.shell_edges(n, width=width)
    nsh = len(edges) - 1
    tri = [(i, j, l) for i in range(nsh) for j in range(i, nsh) for l in range(j, nsh)]
    res = bs._bispectrum_3d(f, shell_width=width, triangles=tri)
    b, ntri = ob.bispectrum_fft(f, L, edges, tri)
    assert np.array_equal(res["ntri"], np.rint(ntri).astype(np.int64))     # integer triangle counts: exact
    ok_ = res["ntri"] > 0
    npt.assert_allclose(res["B"][ok_], b[ok_], rtol=1e-8, atol=1e-9 * np.nanmax(abs(b)))
    # second call reuses the cached triangle counts
    res2 = bs._bispectrum_3d(f, shell_width=width, triangles=tri)
    npt.assert_allclose(res2["B"][ok_], res["B"][ok_], rtol=1e-12)
    # the inherited P(k) behaviour of the reference class is kept
    k, p = bs._power_spectrum_3d(f)
    kr, pr = offt.power_spectrum_3d(f, L)
    npt.assert_allclose(p, pr, rtol=1e-10)


def test_bispectrum_brute_force_tiny(tmp_path):
    from astrild_amd.bispectra import Bispectrum3D
    n, L = 8, 10.0
    f = np.random.default_rng(5).standard_normal((n, n, n)) ** 2
    bs = Bispectrum3D("particles", FakeSimulation(tmp_path, n, L, {}))
    edges = ob.shell_edges(n)
    tri = [(0, 0, 0), (0, 1, 1), (1, 1, 2), (0, 1, 2), (2, 2, 2)]
    res = bs._bispectrum_3d(f, triangles=tri)
    bb, nb = ob.bispectrum_brute_force(f, L, edges, tri)
    assert np.array_equal(res["ntri"], nb)
    npt.assert_allclose(res["B"][nb > 0], bb[nb > 0], rtol=1e-9)


def test_bispectrum_of_a_field_in_physical_units_does_not_overflow():
    """fp32 shell fields are multiplied in fp32 inside the triangle sums: a grid in Msun/h per cell (1e13) would overflow
    |D|^3 - the estimator normalises the fields by max |field| and scales the sums back.  B is cubic in the amplitude."""
    import torch
    from astrild_amd import device as dev
    torch.cuda.set_device(0)
    n, L = 256, 500.0
    g = torch.Generator(device="cuda").manual_seed(77)
    f = torch.randn((n, n, n), dtype=torch.float32, device="cuda", generator=g)
    f = f + 0.3 * f * f
    edges = [1, 9, 17, 33, 65]
    tri = [(0, 0, 0), (0, 1, 1), (1, 2, 2), (2, 3, 3), (1, 2, 3)]
    base = dev.bispectrum(f, L, edges, tri)
    big = dev.bispectrum(f * 1e13, L, edges, tri)
    assert np.all(np.isfinite(big["B"])) and np.array_equal(big["ntri"], base["ntri"])
    npt.assert_allclose(big["B"] / 1e39, base["B"], rtol=2e-5, atol=1e-6 * np.abs(base["B"]).max())


def test_bispectrum_fused_z_passes_and_triangle_sums(monkeypatch):
    """The estimator's tail in ONE kernel (ast_fft_tile_c2r_triangles: the z passes of all shells row by row, the triangle sums
    from LDS, no real cube ever written) against the route through the cubes (ASTRILD_BISPEC_FUSED=0) and against the oracle
    (float64 numpy) at 256^3: equilateral, squeezed, isosceles and scalene bins."""
    import torch
    from astrild_amd import device as dev
    torch.cuda.set_device(0)
    n, L = 256, 500.0
    rng = np.random.default_rng(9)
    f = rng.standard_normal((n, n, n))
    f = (f + 0.3 * f * f).astype(np.float32)
    edges = list(range(1, n // 2 + 1, 8))
    nsh = len(edges) - 1
    tri = [(i, i, i) for i in range(nsh)] + [(0, i, i) for i in range(1, nsh)] + [(1, 2, 3), (2, 5, 6), (0, 7, 7), (3, 4, 6)]
    t = dev.as_device(f)
    dev._tri_cache.clear()
    fused = dev.bispectrum(t, L, edges, tri)
    monkeypatch.setenv("ASTRILD_BISPEC_FUSED", "0")
    cubes = dev.bispectrum(t, L, edges, tri)
    monkeypatch.delenv("ASTRILD_BISPEC_FUSED")
    assert np.array_equal(fused["ntri"], cubes["ntri"])
    ok_ = fused["ntri"] > 0
    scale = np.abs(cubes["B"][ok_]).max()
    npt.assert_allclose(fused["B"][ok_], cubes["B"][ok_], rtol=1e-5, atol=1e-6 * scale)      # same fp32 fields, other summation order
    b, ntri = ob.bispectrum_fft(f.astype(np.float64), L, edges, tri)
    assert np.array_equal(fused["ntri"], np.rint(ntri).astype(np.int64))
    npt.assert_allclose(fused["B"][ok_], b[ok_], rtol=2e-4, atol=1e-5 * scale)
    again = dev.bispectrum(t, L, edges, tri)
    assert np.array_equal(again["B"], fused["B"], equal_nan=True)                            # fixed summation order
    dev._tri_cache.clear()


def _kappa_frame(npix, seed=0):
    rng = np.random.default_rng(seed)
    c2, c3 = ok.C_LIGHT_KMS ** 2, ok.C_LIGHT_KMS ** 3
    return pd.DataFrame({"kappa_2": rng.standard_normal(npix * npix) * 0.02 * c2,
                         "isw_rs": rng.standard_normal(npix * npix) * 1e-6 * c3})


def test_skymap_skyarray_pipeline():
    from astrild_amd.rays import SkyMap
    npix, theta = 128, 10.0
    df = _kappa_frame(npix)
    raw = df["kappa_2"].values.copy()
    sky = SkyMap.from_dataframe(npix, theta, "kappa_2", "/tmp/", df.copy(), "Ray_maps_zrange_0.08_0.90.h5")
    assert sky.npix == npix and sky.opening_angle == theta
    ref_map = ok.rays_to_map(ok.convert_code_to_phy_units("kappa_2", raw))
    assert np.array_equal(sky.data["orig"], ref_map)                      # unit conversion + reshape: bit-exact
    # pdf
    pdf = sky.pdf(50)
    rv, re = ok.pdf(ref_map, 50)
    npt.assert_allclose(pdf["values"], rv, rtol=1e-13)
    npt.assert_allclose(pdf["bins"], re, rtol=1e-15)
    # galaxy shape noise: same numpy PCG64 stream as the reference
    sky.create_galaxy_shape_noise(0.4, 40.0, rnd_seed=34077)
    assert np.array_equal(sky.data["gsn"], ok.galaxy_shape_noise(npix, 34077))
    assert np.array_equal(sky.add_galaxy_shape_noise(), ref_map + sky.data["gsn"])
    # filter dispatch by name, smoothing real-space branch (< 500 px)
    sky.filter({"gaussian": {"theta_i": 2.5, "abbrev": "smooth"}}, on="orig_gsn")
    ref_s = ok.gaussian_smooth(ref_map + sky.data["gsn"], theta, 2.5)
    npt.assert_allclose(sky.data["orig_gsn_smooth"], ref_s, rtol=0, atol=1e-12 * abs(ref_s).max())
    hp = sky.filter({"gaussian_high_pass": {"fwhm_i": 6.0, "abbrev": "hp"}}, on="orig", rtn=True)
    npt.assert_allclose(hp, ref_map - ok.gaussian_smooth(ref_map, theta, ok.fwhm_to_sigma(6.0)),
                        rtol=0, atol=1e-12 * abs(ref_map).max())
    # kappa -> deflection: returns (alpha_2, alpha_1) like sky_array.py:813-817
    ax, ay = sky.convert_convergence_to_deflection(on="orig")
    r1, r2 = ok.kappa0_to_alphas(ref_map, npix, np.deg2rad(theta))
    npt.assert_allclose(ax, r2, rtol=0, atol=1e-10 * abs(r2).max())
    npt.assert_allclose(ay, r1, rtol=0, atol=1e-10 * abs(r1).max())
    # crop / division / merge round trip
    tiles = sky.division(4, of="orig", rtn=True)
    assert tiles.shape == (16, 32, 32)
    assert np.array_equal(sky.merge(tiles, rtn=True), sky.data["orig"])
    with pytest.raises(BaseException):
        SkyMap.from_array(ref_map, npix, theta, "isw_rs", "/tmp/").add_galaxy_shape_noise()


def test_filters_gaussian_fft_branch_large_map():
    from astrild_amd.rays.utils import Filters
    rng = np.random.default_rng(4)
    img = rng.standard_normal((512, 512))
    got = Filters.gaussian(img, 20.0, theta_i=3.0)
    ref = ok.gaussian_smooth(img, 20.0, 3.0)             # >= 500 px -> FFT branch
    npt.assert_allclose(got, ref, rtol=0, atol=1e-12 * abs(ref).max())
    assert Filters.sigma_to_fwhm(Filters.fwhm_to_sigma(1.0)) == pytest.approx(1.0)
    with pytest.raises(ValueError):
        Filters.gaussian(img, 20.0)


class FlatCosmology:
    """comoving_distance(z) = 3000 z [Mpc]: enough to exercise the re-weighting."""

    def comoving_distance(self, z):
        return 3000.0 * z


def _ray_table():
    idx = pd.MultiIndex.from_tuples([(1, 1), (1, 2), (1, 3), (2, 1), (2, 2)], names=["box_nr", "snap_nr"])
    return pd.DataFrame({"redshift": [0.05, 0.10, 0.15, 0.20, 0.25]}, index=idx)


def test_rayramses_sum_snapshots_plain_and_reweighted():
    from astrild_amd.rays import RayRamses
    npix = 64
    frames = {(b, r): _kappa_frame(npix, seed=10 * b + r) for (b, r) in _ray_table().index}

    class MemRay(RayRamses):
        def _load_ray_map(self, ray_file):
            box = int(ray_file.split("box")[1].split("/")[0])
            ray = int(ray_file.split("output")[1].split(".")[0])
            return frames[(box, ray)].copy()

    rr = MemRay({"lc": "/lc/"}, cosmology=FlatCosmology(), ray_info_df=_ray_table())
    out = rr.sum_snapshots(None, ["kappa_2", "isw_rs"], [], {"z": [0.07, 0.22], "box": [0], "ray": [0]})
    sel = [(1, 2), (1, 3), (2, 1)]
    for col in ("kappa_2", "isw_rs"):
        assert np.array_equal(out[col].values, ok.kappa_stack([frames[s][col].values for s in sel]))
    # z_src_shift with planes at redshift <= z_src_shift: the reference raises (rayramses.py:201-205) - the default
    rr = MemRay({"lc": "/lc/"}, cosmology=FlatCosmology(), ray_info_df=_ray_table())
    with pytest.raises(BaseException, match="Redshift shift has not correct data structure"):
        rr.sum_snapshots(None, ["kappa_2", "isw_rs"], ["kappa_2"], {"z": [], "box": [0], "ray": [0]},
                         z_src=0.4, z_src_shift=0.22)
    # ... and sums unweighted when every selected plane lies beyond the shifted source
    rr = MemRay({"lc": "/lc/"}, cosmology=FlatCosmology(), ray_info_df=_ray_table())
    out = rr.sum_snapshots(None, ["kappa_2"], ["kappa_2"], {"z": [0.17, 0.3], "box": [0], "ray": [0]},
                           z_src=0.4, z_src_shift=0.16)
    assert np.array_equal(out["kappa_2"].values, ok.kappa_stack([frames[s]["kappa_2"].values for s in [(2, 1), (2, 2)]]))
    # reweight=True, whole light-cone, source moved from z=0.4 to z=0.22: what the dead block intends.  Planes with
    # z <= 0.22 are re-weighted (kappa_2 only, isw_rs never); z_next by the box rule of :207-210: (1,1) is the
    # smallest ray number of box 1 -> first output of box 2; (1,2) -> next row; (1,3), last row -> itself;
    # (2,1): no box 3 in the table -> next row; (2,2) at z = 0.25 > 0.22 keeps weight 1
    rr = MemRay({"lc": "/lc/"}, cosmology=FlatCosmology(), ray_info_df=_ray_table())
    out = rr.sum_snapshots(None, ["kappa_2", "isw_rs"], ["kappa_2"], {"z": [], "box": [0], "ray": [0]},
                           z_src=0.4, z_src_shift=0.22, reweight=True)
    order = list(_ray_table().index)
    z_near = {(1, 1): 0.05, (1, 2): 0.10, (1, 3): 0.15, (2, 1): 0.20}
    z_next = {(1, 1): 0.20, (1, 2): 0.15, (1, 3): 0.15, (2, 1): 0.25}
    ref = None
    for s_ in order:
        q = frames[s_]["kappa_2"].values.astype(np.float64)
        if s_ in z_near:
            q = ok.translate_redshift(q, 3000.0 * z_near[s_], 3000.0 * z_next[s_], 3000.0 * 0.4, 3000.0 * 0.22)
        ref = q.copy() if ref is None else ref + q
    assert np.array_equal(out["kappa_2"].values, ref)
    assert np.array_equal(out["isw_rs"].values, ok.kappa_stack([frames[s_]["isw_rs"].values for s_ in order]))


def test_simulation_collection_sum_npy_planes(tmp_path):
    from astrild_amd.simcoll import SimulationCollection
    rng = np.random.default_rng(8)
    planes, sims = {}, {}
    for box in (1, 2):
        d = tmp_path / f"box{box}"
        d.mkdir()
        sims[f"box{box}"] = types.SimpleNamespace(dirs={"sim": str(d) + "/"},
                                                  file_dsc={"root": "kappa2_maps", "extension": "npy"})
        for ray in ((1, 2, 3) if box == 1 else (1, 2)):
            planes[(box, ray)] = rng.standard_normal((48, 48))
            np.save(d / f"kappa2_maps_output0000{ray}.npy", planes[(box, ray)])
    sc = SimulationCollection(_ray_table(), sims)
    tot = sc.sum_raytracing_snapshots(None, ["kappa_2"], [], {"z": [], "box": [0], "ray": [0]},
                                      rm_ray={1: [2]})
    order = [(1, 1), (1, 3), (2, 1), (2, 2)]
    assert np.array_equal(tot, ok.kappa_stack([planes[o] for o in order]))


def test_several_source_redshifts_over_resident_planes(tmp_path):
    """z_src_shift as a sequence (the re-weighting loops of rayramses.py:205-222 / simcoll.py:302-320 once per source
    redshift): every plane is loaded once - the loader is counted - and each map equals the one-call result bit for bit."""
    from astrild_amd.rays import RayRamses
    from astrild_amd.simcoll import SimulationCollection
    npix = 64
    frames = {(b, r): _kappa_frame(npix, seed=10 * b + r) for (b, r) in _ray_table().index}
    loads = []

    class MemRay(RayRamses):
        def _load_ray_map(self, ray_file):
            loads.append(ray_file)
            box = int(ray_file.split("box")[1].split("/")[0])
            ray = int(ray_file.split("output")[1].split(".")[0])
            return frames[(box, ray)].copy()

    zs = [0.22, 0.12, 0.26, 0.22]
    rng_all = {"z": [], "box": [0], "ray": [0]}
    rr = MemRay({"lc": "/lc/"}, cosmology=FlatCosmology(), ray_info_df=_ray_table())
    many = rr.sum_snapshots(None, ["kappa_2", "isw_rs"], ["kappa_2"], rng_all, z_src=0.4, z_src_shift=zs, reweight=True)
    assert len(many) == len(zs) and len(loads) == 5                    # five planes, read once for four maps
    for z, got in zip(zs, many):
        one = MemRay({"lc": "/lc/"}, cosmology=FlatCosmology(), ray_info_df=_ray_table()).sum_snapshots(
            None, ["kappa_2", "isw_rs"], ["kappa_2"], rng_all, z_src=0.4, z_src_shift=z, reweight=True)
        assert np.array_equal(got["kappa_2"].values, one["kappa_2"].values)
        assert np.array_equal(got["isw_rs"].values, one["isw_rs"].values)
    assert not np.array_equal(many[0]["kappa_2"].values, many[1]["kappa_2"].values)
    # .npy planes through the collection
    rng = np.random.default_rng(8)
    planes, sims = {}, {}
    for box in (1, 2):
        d = tmp_path / f"box{box}"
        d.mkdir()
        sims[f"box{box}"] = types.SimpleNamespace(dirs={"sim": str(d) + "/"},
                                                  file_dsc={"root": "kappa2_maps", "extension": "npy"})
        for ray in ((1, 2, 3) if box == 1 else (1, 2)):
            planes[(box, ray)] = rng.standard_normal((48, 48))
            np.save(d / f"kappa2_maps_output0000{ray}.npy", planes[(box, ray)])
    sc = SimulationCollection(_ray_table(), sims, cosmology=FlatCosmology())
    many = sc.sum_raytracing_snapshots(None, ["kappa_2"], ["kappa_2"], rng_all, z_src=0.4, z_src_shift=[0.22, 0.12], reweight=True)
    for z, got in zip([0.22, 0.12], many):
        one = SimulationCollection(_ray_table(), sims, cosmology=FlatCosmology()).sum_raytracing_snapshots(
            None, ["kappa_2"], ["kappa_2"], rng_all, z_src=0.4, z_src_shift=z, reweight=True)
        assert got.shape == (48, 48) and np.array_equal(got, one)


def test_sharded_bispectrum_ranks_share_one_gpu(tmp_path):
    """SURVEY.md §8e row 2 rehearsal: 3 processes on cuda:0 over gloo, triangle bins split over the ranks, grid
    broadcast from rank 0; against the single-GPU estimator on the same grid (same kernels: identical numbers)."""
    import os
    import socket
    import subprocess
    import sys
    import torch
    from astrild_amd import device as dev
    torch.cuda.set_device(0)
    world, n = 3, 128
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "bispec.npz")
    worker = os.path.join(os.path.dirname(__file__), "bispec_gpu_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(n), out]) for r in range(world)]
    assert [p.wait(timeout=600) for p in procs] == [0] * world
    got = np.load(out)
    pos = dev.synth_lattice_particles(n, n, 1000.0, seed=3, dtype=torch.float32)
    field = dev.paint(pos, None, n, 1000.0, "cic")
    edges = list(range(1, n // 2 + 1, 4))
    ref = dev.bispectrum(field, 1000.0, edges, [tuple(t) for t in got["tri"]])
    assert np.array_equal(got["ntri"], ref["ntri"])
    ok_ = ref["ntri"] > 0
    npt.assert_allclose(got["B"][ok_], ref["B"][ok_], rtol=1e-12)


def test_density_file_to_device_grid_to_power_spectrum(tmp_path):
    """§8f-4 on the GPU: a DTFE density binary (.a_den; float payload, header variants) goes through
    PowerSpectrum3D._read_data (file -> pinned staging -> device, conversion to the class dtype on the device) and
    through the P(k) path; compared with the oracle on the same array.  Reference: particles/hutils/density.py:345-442
    (reader), power_spectrum_3d.py:140-153 (_read_data), dtfe.py:70-80 (the .npy detour this replaces)."""
    from astrild_amd import formats
    from astrild_amd.power_spectra import PowerSpectrum3D
    rng = np.random.default_rng(21)
    n, L = 64, 500.0
    grid = (rng.standard_normal((n, n, n)) + 3.0).astype(np.float32)
    variants = {"snap_005.a_den": dict(file_type=1), "snap_005.a_velDiv": dict(file_type=13, redshift=0.5, Omega0=0.3),
                "snap_005.den": dict(file_type=50, HubbleParam=0.7, method=np.uint64(3))}
    for name, kw in variants.items():
        path = str(tmp_path / name)
        formats.write_density_grid(path, grid, L, **kw)
        sim = FakeSimulation(tmp_path, n, L, {5: path})
        for dtype in (torch.float64, torch.float32):
            ps = PowerSpectrum3D("particles", sim)
            ps.dtype = dtype
            dgrid = ps._read_data(path)
            assert dgrid.is_cuda and dgrid.dtype == dtype and tuple(dgrid.shape) == (n, n, n)
            npt.assert_array_equal(dgrid.cpu().numpy(), grid.astype(np.float64 if dtype == torch.float64 else np.float32))
            k, p = ps._power_spectrum_3d(dgrid)
            kr, pr = offt.power_spectrum_3d(grid.astype(np.float64), L)
            npt.assert_allclose(k, kr, rtol=1e-12)
            npt.assert_allclose(p, pr, rtol=1e-9 if dtype == torch.float64 else 1e-6)
    # a vector file (three components per grid point) is a 4-D array: refused like in the reference (:68)
    vec = str(tmp_path / "snap_005.a_vel")
    formats.write_density_grid(vec, np.stack([grid] * 3, axis=3), L, file_type=11)
    sim = FakeSimulation(tmp_path, n, L, {5: vec})
    ps = PowerSpectrum3D("particles", sim)
    assert tuple(ps._read_data(vec).shape) == (n, n, n, 3)
    with pytest.raises(BaseException):
        ps.compute(["vel"], [{"path": "x", "root": "snap", "extension": "a_vel"}], save=False)


def test_fp32_cross_spectrum_meets_the_accuracy_of_the_auto_path():
    """fp32 cross spectra at a tile-FFT size used to go through the fp32 transform with the O(1) mean left
    in; they are transformed in double now.  N = 256, against the float64 pipeline."""
    from astrild_amd import device as dev
    n, L = 256, 1000.0
    rng = np.random.default_rng(4)
    a = (rng.standard_normal((n, n, n)) * 0.1 + 1.0).astype(np.float32)
    b = (a + rng.standard_normal((n, n, n)).astype(np.float32) * 0.05).astype(np.float32)
    r32 = dev.fftpower_1d(dev.as_device(a), L, dev.as_device(b))
    r64 = dev.fftpower_1d(dev.as_device(a.astype(np.float64)), L, dev.as_device(b.astype(np.float64)))
    assert np.array_equal(r32["modes"], r64["modes"])
    npt.assert_allclose(r32["power"], r64["power"], rtol=1e-6)
