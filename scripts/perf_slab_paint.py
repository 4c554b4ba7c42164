"""Kernel breakdown of one rank's slab paint at the P = 8, 1024^3 shape (138 planes incl. ghosts, 134 M particles)."""
import sys, torch
sys.path.insert(0, ".")
from astrild_amd import device as dev, slab
P = 8; n, L = 1024, 1000.0; nloc = n // P
ops = slab.HipSlabOps(torch.float32)
ppr = n ** 3 // P
pos = ops.synth(n, n, L, 1, False, 0, ppr)
gl = 5
buf = ops.empty((nloc + 2 * gl, n, n))
mean = 1.0
f = lambda: ops.paint(pos, None, n, L, 'cic', buf, (0 - gl) % n, nloc + 2 * gl, offset=mean, owned=(gl, nloc))
f(); f(); torch.cuda.synchronize()
dev.profile_enable(True)
for _ in range(5): f()
torch.cuda.synchronize()
print({k: round(v[1] / 5, 4) for k, v in dev.profile_report().items()})
dev.profile_enable(False)
