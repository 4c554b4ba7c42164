"""Where SubFind.power_spectrum's time goes (2e6 objects, TSC, nbins 512, float64, host arrays in): host unit conversion,
H2D, paint, transform + binning, D2H.  usage: python scripts/perf_subfind.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from astrild_amd import device as dev

nobj, nbins, boxsize, h = 2_000_000, 512, 500.0, 0.6774
rng = np.random.default_rng(20240601)
centres = rng.uniform(0.0, boxsize, size=(4096, 3))
which = rng.integers(0, 4096, size=nobj)
radius = rng.uniform(0.5, 4.0, size=4096)[which]
pos0 = np.mod(centres[which] + rng.standard_normal((nobj, 3)) * radius[:, None], boxsize) * 1e3 / h
mass0 = 10.0 ** rng.uniform(0.0, 3.0, size=nobj) * 1e10 / h


def timed(label, fn, reps=5, sync=True):
    fn()
    if sync:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    if sync:
        torch.cuda.synchronize()
    print(f"{label:58s} {(time.perf_counter() - t0) / reps * 1e3:8.3f} ms", flush=True)
    return out


pos_h = timed("host: pos * h / 1e3, mass * h / 1e10 (numpy)", lambda: (pos0 * h / 1e3, mass0 * h / 1e10), sync=False)
pos_d = timed("as_device(pos) + as_device(mass) (pageable H2D)", lambda: (dev.as_device(np.ascontiguousarray(pos_h[0]), torch.float64),
                                                                         dev.as_device(np.ascontiguousarray(pos_h[1]), torch.float64)))
dx = boxsize / nbins
timed("paint (direct atomics, fresh grid)", lambda: dev.paint(pos_d[0], pos_d[1], nbins, boxsize, "tsc", scale=1.0 / dx ** 3))
grid = dev.paint(pos_d[0], pos_d[1], nbins, boxsize, "tsc", scale=1.0 / dx ** 3)
timed("paint tiled (forced)", lambda: dev.paint(pos_d[0], pos_d[1], nbins, boxsize, "tsc", scale=1.0 / dx ** 3, method="tiled"))
timed("paint tiled2 (forced)", lambda: dev.paint(pos_d[0], pos_d[1], nbins, boxsize, "tsc", scale=1.0 / dx ** 3, method="tiled2"))
timed("fftpower_1d(grid) incl. D2H of the sums", lambda: dev.fftpower_1d(grid, boxsize))
timed("paint_power_1d (device arrays in)", lambda: dev.paint_power_1d(pos_d[0], pos_d[1], nbins, boxsize, "tsc", scale=1.0 / dx ** 3))
from astrild_amd.particles.hutils.stats_subfind import SubFind
import types
snap = types.SimpleNamespace(cat={"SubhaloPos": pos0, "SubhaloMass": mass0}, header=types.SimpleNamespace(hubble=h, boxsize=boxsize * 1e3))
timed("SubFind.power_spectrum (whole call)", lambda: SubFind.power_spectrum(snap, nbins=nbins, boxsize=boxsize))
dev.profile_enable(True)
SubFind.power_spectrum(snap, nbins=nbins, boxsize=boxsize)
torch.cuda.synchronize()
print({k: round(v[1], 3) for k, v in dev.profile_report().items()})
