"""Self-validation of the bispectrum oracle.  CPU only."""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import bispectrum as ob


@pytest.mark.parametrize("n", [8, 12])
def test_fft_estimator_matches_brute_force_triangle_sum(n):
    rng = np.random.default_rng(n)
    f = rng.standard_normal((n, n, n))
    f = f + 0.3 * f ** 2                       # some real non-Gaussianity
    edges = ob.shell_edges(n)                  # unit shells 1..n/2
    nsh = len(edges) - 1
    tri = [(i, j, l) for i in range(nsh) for j in range(i, nsh) for l in range(j, nsh)]
    bf, nf = ob.bispectrum_fft(f, 10.0, edges, tri)
    bb, nb = ob.bispectrum_brute_force(f, 10.0, edges, tri)
    npt.assert_array_equal(np.rint(nf).astype(np.int64), nb)       # triangle counts: integers, exact
    npt.assert_allclose(nf, nb, atol=1e-6)
    ok = nb > 0
    npt.assert_allclose(bf[ok], bb[ok], rtol=1e-9, atol=1e-9 * np.nanmax(abs(bb)))


def test_three_plane_waves_closed_triangle():
    # delta = sum_a A_a cos(k_a.x) with k1 + k2 + k3 = 0: <d(k1) d(k2) d(k3)> = A1 A2 A3 / 8
    n, L = 16, 50.0
    k1, k2 = np.array([2, 0, 0]), np.array([-1, 2, 0])
    k3 = -(k1 + k2)
    amps = (0.5, 0.3, 0.2)
    x = np.arange(n) / n
    X = np.stack(np.meshgrid(x, x, x, indexing="ij"), axis=-1)
    f = sum(a * np.cos(2 * np.pi * (X @ k)) for a, k in zip(amps, (k1, k2, k3)))
    edges = ob.shell_edges(n)
    sh = [int(np.floor(np.sqrt((k ** 2).sum()))) - 1 for k in (k1, k2, k3)]      # |k| = 2, 2.236, 2.236 -> shell 1
    b, ntri = ob.bispectrum_fft(f, L, edges, [tuple(sorted(sh))])
    bb, nb = ob.bispectrum_brute_force(f, L, edges, [tuple(sorted(sh))])
    assert int(round(ntri[0])) == nb[0] > 0
    # ordered triplets (k_a, k_b, k_c) that hit the three waves: 3! permutations x 2 signs
    expected = L ** 6 * 12 * np.prod(amps) / 8 / nb[0]
    npt.assert_allclose(b[0], expected, rtol=1e-10)
    npt.assert_allclose(bb[0], expected, rtol=1e-10)


def test_gaussian_field_bispectrum_is_consistent_with_zero():
    rng = np.random.default_rng(3)
    n, L = 32, 100.0
    f = rng.standard_normal((n, n, n))
    edges = ob.shell_edges(n, width=4, m_min=2)
    nsh = len(edges) - 1
    tri = [(i, i, i) for i in range(nsh)]
    b, ntri = ob.bispectrum_fft(f, L, edges, tri)
    p = L ** 3 / n ** 3                                     # white-noise P
    sigma = np.sqrt(6 * p ** 3 * L ** 3 / ntri)             # Gaussian variance of the equilateral estimator
    assert np.all(np.abs(b) < 5 * sigma)
