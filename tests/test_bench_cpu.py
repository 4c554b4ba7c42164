"""bench.py's chunked CPU baseline (the 1024^3 leg) paints with the oracle's arithmetic: checked against oracle.mesh.paint."""
import numpy as np
import pytest


@pytest.mark.parametrize("window", ["cic", "tsc"])
def test_chunked_cpu_paint_equals_the_oracle_paint(window):
    import bench
    from oracle import mesh as omesh
    n, L, planes, pad = 32, 100.0, 8, 6
    pos = omesh.lattice_particles(n, n, L, seed=5)
    grid = np.zeros((n, n, n))
    per = n * n * planes
    for c, i0 in enumerate(range(0, n, planes)):
        bench._cpu_paint_chunk(pos[c * per:(c + 1) * per], i0, planes, pad, n, L, window, grid)
    ref = omesh.paint(pos, None, n, L, window)
    np.testing.assert_allclose(grid, ref, rtol=1e-13, atol=1e-13)
    res = bench.cpu_baseline_chunked(32, window, L, planes=8, pad=6)
    assert res["value"] > 0 and res["cores"] == 1 and res["kind"] == "port"


def test_parallel_cpu_paint_child_paints_every_particle():
    """The threaded leg's paint runs in a child interpreter with forked workers: same particles, same grid total."""
    import bench
    res = bench.cpu_parallel_paint(80, "cic", 1000.0, workers=2)       # 5 chunks: the odd-count path
    assert "error" not in res, res
    assert res["nparticles"] == 80 ** 3 and res["paint_s"] > 0
    assert abs(res["grid_sum"] - 80 ** 3) < 1e-6 * 80 ** 3
    from oracle import mesh as omesh
    ref = omesh.paint(omesh.lattice_particles(80, 80, 1000.0, seed=20240601), None, 80, 1000.0, "cic")
    np.testing.assert_allclose(res["moments"], bench._grid_moments(ref), rtol=1e-12)


def test_auto_paint_method_thresholds():
    """device.auto_paint_method is plain arithmetic (no GPU): direct atomics for sparse catalogues, two-pass lists for small or
    clumpy-sparse ones, the probe-driven tiled paths for dense input; accumulate keeps round 1's threshold."""
    from astrild_amd.device import auto_paint_method as m
    n = 512
    per_tile = n ** 3 // 2048
    assert m(7 * per_tile, n, n, "tsc") == "direct" and m(8 * per_tile, n, n, "tsc") == "tiled2"
    assert m(15 * per_tile, n, n, "cic") == "direct" and m(16 * per_tile, n, n, "cic") == "tiled2"
    assert m(63 * per_tile, n, n, "cic") == "tiled2" and m(64 * per_tile, n, n, "cic") == "tiled"       # (>= 2^20 objects: the probe decides)
    assert m(1 << 19, 128, 128, "cic") == "tiled2"                                                       # dense but too small for the probe
    assert m(1 << 24, 256, 256, "cic", hint="scattered") == "tiled" and m(1 << 24, 256, 256, "cic") == "tiled"
    assert m(40 * per_tile, n, n, "cic", accumulate=True) == "direct" and m(64 * per_tile, n, n, "cic", accumulate=True) == "tiled"
    assert m(10 ** 7, 100, 100, "cic") == "direct"                                                       # nmesh not a multiple of 32
    assert m(1000, n, n, "cic") == "direct"
