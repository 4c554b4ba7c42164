"""Plain float64 r2c at 1024^3 (z rows, y and x column passes with stores) beside the power pipeline's passes.
usage: python scripts/perf_f64_plain.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrild_amd import device as dev
n = 1024
t = torch.randn((n, n, n), dtype=torch.float64, device="cuda")
spec = torch.empty((n, n, n // 2 + 1), dtype=torch.complex128, device="cuda")
for _ in range(2): dev.r2c(t, out=spec)
torch.cuda.synchronize()
dev.profile_enable(True)
for _ in range(5): dev.r2c(t, out=spec)
torch.cuda.synchronize()
print("f64 r2c 1024^3 (natural pitch):", {k: round(v[1] / 5, 3) for k, v in dev.profile_report().items()}, flush=True)
