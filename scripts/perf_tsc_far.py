"""Is the TSC grouping slow because of the 'far' path (nearest cell index == n for particles in the last half cell)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrild_amd import device as dev
n, L = 1024, 1000.0
grid = torch.empty((n, n, n), dtype=torch.float32, device="cuda")
pos = dev.synth_lattice_particles(n, n, L, dtype=torch.float32)
for clamp in (False, True):
    if clamp:
        pos.clamp_(max=L * (1023.4 / 1024.0))
    for w in ("cic", "tsc"):
        dev.paint(pos, None, n, L, w, out=grid, method="tiled", accumulate=False, defer_fold=True, offset=1.0)
        torch.cuda.synchronize()
        dev.profile_enable(True)
        for _ in range(5):
            dev.paint(pos, None, n, L, w, out=grid, method="tiled", accumulate=False, defer_fold=True, offset=1.0, check_dropped=False)
        torch.cuda.synchronize()
        print("clamped" if clamp else "plain  ", w, {k: round(v[1] / 5, 3) for k, v in dev.profile_report().items()}, flush=True)
        dev.profile_enable(False)
