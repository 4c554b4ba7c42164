import sys, time, torch
sys.path.insert(0, ".")
from astrild_amd import device as dev
for n in (256, 512, 1024):
    g = torch.Generator(device="cuda").manual_seed(n)
    t = torch.randn((n, n, n), dtype=torch.float64, device="cuda", generator=g)
    for fused in (True, False):
        for _ in range(2): dev.fftpower_1d(t, 1000.0, fused=fused)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 20 if n < 1024 else 5
        for _ in range(reps): dev.fftpower_1d(t, 1000.0, fused=fused)
        torch.cuda.synchronize(); print(n, "fused64" if fused else "rocFFT+bin", round((time.perf_counter() - t0) / reps * 1e3, 3), "ms", flush=True)
    del t
