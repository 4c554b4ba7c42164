"""Side 2048: single-precision passes against the double passes on the same fp32 grid (npside^3 lattice particles): per-shell
deviation and kernel times.  usage: python scripts/perf_big32.py [npside]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from astrild_amd import device as dev

n, L = 2048, 1000.0
npside = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pos = dev.synth_lattice_particles(npside, n, L, seed=20240601, dtype=torch.float32)
grid = dev.paint(pos, None, n, L, "cic", method="tiled", offset="mean", check_dropped=False)
del pos
torch.cuda.empty_cache()
res = {}
for name, off in (("fp32 passes", None), ("double passes", "1")):
    if off:
        os.environ["ASTRILD_FFT32_BIG_OFF"] = off
    dev.power_sums_fused64(grid, L, mean=0.0)
    torch.cuda.synchronize()
    dev.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(3):
        sums = dev.power_sums_fused64(grid, L, mean=0.0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    prof = dev.profile_report()
    dev.profile_enable(False)
    res[name] = dev.finish_power(*sums)
    print(f"{name}: {ms:.2f} ms  " + "  ".join(f"{k}={v[1] / 3:.2f}" for k, v in prof.items()), flush=True)
    os.environ.pop("ASTRILD_FFT32_BIG_OFF", None)
    dev._power_scratch.clear()
    torch.cuda.empty_cache()
rel = np.abs(res["fp32 passes"]["power"] / res["double passes"]["power"] - 1.0)
print("modes equal:", np.array_equal(res["fp32 passes"]["modes"], res["double passes"]["modes"]))
print("rel deviation, shells 0-39:", " ".join(f"{v:.1e}" for v in rel[:40]))
print("max over shells >= 40: %.2e; shells above 1e-6: %s" % (rel[40:].max(), np.nonzero(rel > 1e-6)[0].tolist()))
print("P(k) of the first shells:", " ".join(f"{v:.3e}" for v in res["double passes"]["power"][:12]))
