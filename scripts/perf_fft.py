import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrild_amd import device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
def timeit(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]
g = torch.randn((n, n, n), device="cuda")
spec = torch.empty((n, n, n // 2 + 1), dtype=torch.complex64, device="cuda")
for eng in ("rocfft", "tile"):
    t = timeit(lambda: dev.r2c(g, out=spec, engine=eng))
    print(f"n={n} r2c {eng}: {t:.3f} ms  {24 * n**3 / t / 1e6:.0f} GB/s of 24 B/cell")
dev.profile_enable(True)
for _ in range(5): dev.r2c(g, out=spec, engine="tile")
torch.cuda.synchronize()
print({k: round(v[1] / 5, 3) for k, v in dev.profile_report().items()})
