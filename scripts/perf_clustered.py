"""Paint of the clustered synthetic set (lattice collapsing onto attractors) by each path: which one for which input?
usage: python scripts/perf_clustered.py [n] [window]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrild_amd import device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
window = sys.argv[2] if len(sys.argv) > 2 else "cic"
L = 1000.0
for shuffle in (False, True):
    pos = dev.synth_clustered_particles(n, n, L, seed=20240601, shuffle=shuffle, dtype=torch.float32)
    pr = dev.probe_input(pos, n, L)
    print(f"n={n} {window} {'shuffled' if shuffle else 'file order'}: probe groupable {pr['groupable']:.3f} est overflow {pr['overflow'] / pos.shape[0]:.3%} "
          f"max tile {pr['max_tile'] / pr['mean_tile']:.1f} x mean", flush=True)
    grid = torch.empty((n, n, n), dtype=torch.float32, device="cuda")
    ref = None
    for hint in ("clustered", "scattered", "ordered"):
        st = {}
        for _ in range(2):
            dev.paint(pos, None, n, L, window, out=grid, method="tiled", check_dropped=False, accumulate=False, hint=hint)
        torch.cuda.synchronize()
        dev.profile_enable(True)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            dev.paint(pos, None, n, L, window, out=grid, method="tiled", check_dropped=False, accumulate=False, hint=hint)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        rep = dev.profile_report()
        dev.profile_enable(False)
        dev.paint(pos, None, n, L, window, out=grid, method="tiled", check_dropped=False, accumulate=False, hint=hint, stats=st)
        if ref is None:
            ref = grid.clone()
        err = float((grid - ref).abs().max()) / float(ref.abs().max())
        print(f"   hint={hint:10s} {ms:8.3f} ms  " + "  ".join(f"{k.split('.')[-1]}={v[1] / reps:.3f}" for k, v in rep.items()) +
              f"   overflow list {st.get('overflow')}  max|diff| / max {err:.1e}", flush=True)
    del pos, grid, ref
    torch.cuda.empty_cache()
