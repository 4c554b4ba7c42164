"""The 31 masked inverse transforms of the 512^3 bispectrum leg, once, shell by shell (for rocprofv3 --pmc: per-dispatch
traffic of the x, y and z passes against the pruning model printed by scripts/pmc_per_launch.py).
usage: python scripts/bispec_shells_once.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrild_amd import device as dev
n, L, width = 512, 1000.0, 8
pos = dev.synth_lattice_particles(n, n, L, seed=20240601, dtype=torch.float32)
grid = dev.paint(pos, None, n, L, "cic")
del pos
spec = dev.r2c(grid, engine="tile")
edges = list(range(1, n // 2 + 1, width))
# scratch spectrum as device.bispectrum allocates it (NATURAL_PITCH=1: rows of n/2+1 elements, as before round 4)
pitch = n // 2 + 1 if os.environ.get("NATURAL_PITCH") else dev.tile_work_pitch(n)
work = torch.empty((n, n, pitch), dtype=spec.dtype, device=spec.device)
out = torch.empty((n, n, n), dtype=torch.float32, device=spec.device)
for lo, hi in zip(edges[:-1], edges[1:]):
    dev.c2r_tile_batch(spec, [(lo, hi)], [work], [out])
torch.cuda.synchronize()
print("shells", len(edges) - 1, float(out.double().sum()))
