"""Per-pass times of the float64 3-D power spectrum at 1024^3 (fft64.rows_r2c / cols / cols_power).  usage: python scripts/perf_f64_passes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrild_amd import device as dev
n = 1024
g = torch.Generator(device="cuda").manual_seed(n)
t = torch.randn((n, n, n), dtype=torch.float64, device="cuda", generator=g)
for _ in range(2):
    dev.fftpower_1d(t, 1000.0)
torch.cuda.synchronize()
dev.profile_enable(True)
reps = 5
for _ in range(reps):
    res = dev.fftpower_1d(t, 1000.0)
torch.cuda.synchronize()
print("f64 1024^3:", {k: round(v[1] / reps, 3) for k, v in dev.profile_report().items()}, "P[10] = %.9e" % res["power"][10], flush=True)
