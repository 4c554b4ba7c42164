"""The unordered (two-level scatter) paint and the fp64 / bispectrum legs, repeated: grids and spectra must be bit-identical
from call to call (fixed-point tiles, fixed-order reductions; the order in which workgroups append to the segments varies).
usage: python scripts/soak_scattered_repeat.py"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from astrild_amd import device as dev
for n, reps in ((256, 40), (512, 20), (1024, 6)):
    pos = dev.synth_lattice_particles(n, n, 1000.0, seed=11, shuffle=True, dtype=torch.float32)
    for window in ("cic", "tsc"):
        ref, bad = None, 0
        for i in range(reps):
            g = dev.paint(pos, None, n, 1000.0, window, method="tiled", accumulate=False, hint="scattered", check_dropped=False)
            if ref is None:
                ref = g.clone()
            elif not torch.equal(ref, g):
                bad += 1
            del g
        print(f"n={n} shuffled {window}: {reps} paints, {bad} differ from the first", flush=True)
        del ref
    del pos
n = 512
grid = dev.paint(dev.synth_lattice_particles(n, n, 1000.0, seed=3, dtype=torch.float32), None, n, 1000.0, "cic")
edges = list(range(1, n // 2 + 1, 8))
nsh = len(edges) - 1
tri = [(i, i, i) for i in range(nsh)] + [(0, i, i) for i in range(1, nsh)]
ref, bad = None, 0
for i in range(10):
    b = np.asarray(dev.bispectrum(grid, 1000.0, edges, tri)["B"])
    if ref is None:
        ref = b.copy()
    elif not np.array_equal(ref, b):
        bad += 1
print(f"bispectrum 512^3: 10 calls, {bad} differ from the first", flush=True)
