"""Time the per-rank HIP pieces of the slab pipeline at the P=8, 1024^3 shape on one GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrild_amd import device as dev, slab
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, L = 1024, 1000.0
nloc, nz = n // P, n // 2 + 1
ops = slab.HipSlabOps(torch.float32)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]
ppr = n ** 3 // P
pos = ops.synth(n, n, L, 1, False, 0, ppr)
gl = 5
buf = ops.empty((nloc + 2 * gl, n, n))
print(f"P={P}: slab paint {ppr} particles:", timeit(lambda: ops.paint(pos, None, n, L, 'cic', buf, (0 - gl) % n, nloc + 2 * gl)), "ms")
owned = buf[gl:gl + nloc]
spec2d = ops.empty((nloc, n, nz), ops.cdtype)
print("2D R2C batch:", timeit(lambda: ops.fft2d_planes(owned, spec2d)), "ms")
packed = ops.empty((P, nloc, nloc, nz), ops.cdtype)
print("pack:", timeit(lambda: ops.pack(spec2d, packed, P)), "ms")
block = ops.empty((n, nloc, nz), ops.cdtype)
block.copy_(packed.reshape(n, nloc, nz))
print("axis-0 strided C2C:", timeit(lambda: ops.fft1d_axis0(block, 1.0)), "ms")
psum = ops.zeros((n // 2 - 1,), torch.float64)
ops.shell_geometry(n, L, (0, n), (0, nloc))
print("block power_bin:", timeit(lambda: ops.power_bin(block, n, L, (0, n), (0, nloc), psum)), "ms")
