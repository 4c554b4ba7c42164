"""Per-kernel times and list statistics of the tiled overwrite paint (dev tool).
usage: python scripts/perf_paint.py [n=1024] [windows=cic,tsc] [orders=natural,shuffled] [reps=5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrild_amd import device as dev

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
windows = sys.argv[2].split(",") if len(sys.argv) > 2 else ["cic", "tsc"]
orders = sys.argv[3].split(",") if len(sys.argv) > 3 else ["natural", "shuffled"]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
L = 1000.0
grid = torch.empty((n, n, n), dtype=torch.float32, device="cuda")
for order in orders:
    pos = dev.synth_lattice_particles(n, n, L, shuffle=(order == "shuffled"), dtype=torch.float32)
    hint = "scattered" if order == "shuffled" else None
    for w in windows:
        st = {}
        dev.paint(pos, None, n, L, w, out=grid, method="tiled", accumulate=False, defer_fold=True, offset=1.0, hint=hint, stats=st)
        torch.cuda.synchronize()
        dev.profile_enable(True)
        for _ in range(reps):
            dev.paint(pos, None, n, L, w, out=grid, method="tiled", accumulate=False, defer_fold=True, offset=1.0,
                      hint=hint, check_dropped=False)
        torch.cuda.synchronize()
        prof = dev.profile_report()
        dev.profile_enable(False)
        ms = {k: round(v[1] / reps, 3) for k, v in prof.items()}
        chk = (float(grid.double().sum()), float((grid.double() ** 2).sum()))      # fixed-point sums: equal between variants
        print(f"n={n} {order:8s} {w}: total {sum(ms.values()):7.3f} ms  {ms}  lists {st}  check {chk}", flush=True)
    del pos
