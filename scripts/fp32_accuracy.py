"""fp32 pipeline vs fp64 pipeline on the bench input: per-shell relative deviation, with and
without removing the mean on load."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from astrild_amd import device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = 1000.0
pos = dev.synth_lattice_particles(n, n, L, seed=20240601, dtype=torch.float32)
g32 = dev.paint(pos, None, n, L, "cic")
ref = dev.fftpower_1d(dev.paint(pos.double(), None, n, L, "cic"), L)
for label, mean in (("plain", 0.0), ("mean removed on load", 1.0)):
    r = dev.finish_power(*dev.power_sums_fused(g32, L, mean=mean))
    e = np.abs(r["power"] / ref["power"] - 1)
    print(f"n={n} fp32 {label:22s}: max rel dev {e.max():.2e} (shell {e.argmax()}), shells>=N/8 {e[n//8:].max():.2e}, "
          f"lowest 8 {e[:8].max():.2e}, rel to peak {np.abs(r['power'] - ref['power']).max() / ref['power'].max():.2e}")
r = dev.fftpower_1d(g32, L, fused=False)
e = np.abs(r["power"] / ref["power"] - 1)
print(f"n={n} fp32 rocFFT-free unfused tile r2c + bin: max rel dev {e.max():.2e}")
