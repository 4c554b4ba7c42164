"""Direct global atomics against the LDS-tile paint for SPARSE catalogues (objects per 8x8x32-cell tile from 1 to 64):
where `device.paint(method="auto")` should switch.  usage: python scripts/perf_sparse.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from astrild_amd import device as dev

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = 500.0
rng = np.random.default_rng(3)
for dtype, window in ((torch.float64, "tsc"), (torch.float32, "cic"), (torch.float32, "tsc")):
    for per_tile in (1, 2, 4, 8, 16, 32, 64):
        nobj = per_tile * n ** 3 // 2048
        centres = rng.uniform(0.0, L, size=(4096, 3))
        which = rng.integers(0, 4096, size=nobj)
        pos = np.mod(centres[which] + rng.standard_normal((nobj, 3)) * rng.uniform(0.5, 4.0, size=4096)[which][:, None], L)
        dpos = dev.as_device(pos, dtype)
        dmass = dev.as_device(10.0 ** rng.uniform(0.0, 3.0, size=nobj), dtype)
        row = []
        for method in ("direct", "tiled", "tiled2"):
            for _ in range(2):
                dev.paint(dpos, dmass, n, L, window, method=method, check_dropped=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                dev.paint(dpos, dmass, n, L, window, method=method, check_dropped=False)
            torch.cuda.synchronize()
            row.append("%s %7.3f" % (method, (time.perf_counter() - t0) / 5 * 1e3))
        print(f"n={n} {str(dtype)[6:]:8s} {window} {per_tile:3d} per tile ({nobj:9d} objects): " + "  ".join(row) + " ms", flush=True)
