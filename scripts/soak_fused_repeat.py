import sys, numpy as np, torch
sys.path.insert(0, ".")
from astrild_amd import device as dev
for n in (256, 512, 1024):
    pos = dev.synth_lattice_particles(n, n, 1000.0, seed=11, dtype=torch.float32)
    ref, bad, worst = None, 0, 0.0
    N = 600 if n == 256 else 200 if n == 512 else 100
    for i in range(N):
        p = np.asarray(dev.paint_power_1d(pos, None, n, 1000.0, "cic")["power"])
        if ref is None:
            ref = p.copy()
        if not np.array_equal(ref, p):
            bad += 1
            worst = max(worst, float(np.max(np.abs(p / ref - 1))))
    print(f"n={n}: {N} calls, {bad} differ from the first, worst relative difference {worst:.3g}", flush=True)
