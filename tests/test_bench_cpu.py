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
