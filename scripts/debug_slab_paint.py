"""Debug: slab-buffer paint of one rank's particles vs the same particles painted on the full grid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrild_amd import device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L = 1000.0
nloc = n // world
gl = gh = 5
for rank in range(world):
    ppr = n ** 3 // world
    pos = dev.synth_lattice_particles(n, n, L, seed=5, dtype=torch.float32, first=rank * ppr, count=ppr)
    x_start = (rank * nloc - gl) % n
    nx = nloc + gl + gh
    full = dev.paint(pos, None, n, L, "cic", method="direct")
    planes = (x_start + torch.arange(nx, device="cuda")) % n
    ref = full[planes]
    for method, acc in (("tiled", False), ("tiled2", False), ("tiled", True), ("direct", None)):
        st = {}
        try:
            got = dev.paint(pos, None, n, L, "cic", method=method, x_start=x_start, nx_alloc=nx, accumulate=acc, stats=st)
            diff = (got - ref).abs()
            bad = (diff > 1e-4).nonzero()
            print(f"rank {rank} {method} acc={acc}: max diff {float(diff.max()):.3e}, bad cells {len(bad)}, stats {st}", flush=True)
            if len(bad):
                print("   first bad:", bad[:5].tolist(), "planes with bad:", torch.unique(bad[:, 0])[:20].tolist(), flush=True)
        except Exception as e:
            print(f"rank {rank} {method} acc={acc}: EXC {e}", flush=True)
    del full, ref
