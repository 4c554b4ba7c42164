"""Exploratory per-stage timing of the 3D pipeline (dev tool, not the bench contract)."""
import sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from astrild_amd import device as dev

def timeit(fn, reps=5, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    methods = sys.argv[2].split(",") if len(sys.argv) > 2 else ["direct", "tiled"]
    windows = sys.argv[3].split(",") if len(sys.argv) > 3 else ["cic", "tsc"]
    L = 1000.0
    dt = torch.float32
    for shuffle in (False, True):
        pos = dev.synth_lattice_particles(n, n, L, shuffle=shuffle, dtype=dt)
        torch.cuda.synchronize()
        npart = pos.shape[0]
        grid = torch.zeros((n, n, n), dtype=dt, device="cuda")
        for method in methods:
            for window in windows:
                def f():
                    grid.zero_()
                    dev.paint(pos, None, n, L, window, out=grid, method=method, check_dropped=False)
                t = timeit(f, reps=3)
                print(f"n={n} shuffle={shuffle} paint {method:6s} {window}: {t:8.3f} ms  {npart / t / 1e6:9.1f} Mpart/s", flush=True)
        del pos
    spec = dev.r2c(grid)
    t = timeit(lambda: dev.r2c(grid, out=spec))
    print(f"n={n} r2c fp32: {t:8.3f} ms   {24 * n**3 / t / 1e6:8.1f} GB/s (24 B/cell alg)")
    sums = dev.power_bin_1d(spec, None, n, L)
    t = timeit(lambda: dev.power_bin_1d(spec, None, n, L, psum=sums[1]))
    print(f"n={n} bin  fp32: {t:8.3f} ms   {4 * n**3 / t / 1e6:8.1f} GB/s (4 B/cell alg)")

main()
