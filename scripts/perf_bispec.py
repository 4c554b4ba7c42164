"""Kernel times of the 512^3 bispectrum leg (config E).  usage: python scripts/perf_bispec.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrild_amd import device as dev
n, L, width = 512, 1000.0, 8
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pos = dev.synth_lattice_particles(n, n, L, seed=20240601, dtype=torch.float32)
grid = dev.paint(pos, None, n, L, "cic")
del pos
edges = list(range(1, n // 2 + 1, width))
nsh = len(edges) - 1
tri = [(i, i, i) for i in range(nsh)] + [(0, i, i) for i in range(1, nsh)] + [(i, i, min(nsh - 1, 2 * i)) for i in range(1, nsh // 2)]
dev.bispectrum(grid, L, edges, tri)
torch.cuda.synchronize()
dev.profile_enable(True)
import time
t0 = time.perf_counter()
for _ in range(reps):
    dev.bispectrum(grid, L, edges, tri)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps * 1e3
print(f"bispectrum {n}^3: {dt:.3f} ms per call;", {k: round(v[1] / reps, 3) for k, v in dev.profile_report().items()}, flush=True)
