"""Time the per-rank HIP pieces of the slab pipeline at the P-rank, 1024^3 shape on one GPU (default P = 8):
what one rank computes per step, without the exchanges."""
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
total = 0.0
def line(name, ms):
    global total
    total += ms
    print(f"{name:55s} {ms:7.3f} ms", flush=True)
ppr = n ** 3 // P
pos = ops.synth(n, n, L, 1, False, 0, ppr)
gl = int(sys.argv[2]) if len(sys.argv) > 2 else 4          # ghost + 1 planes on each side (bench.py: ghost = 3)
buf = ops.empty((nloc + 2 * gl, n, n))
mean = float(ppr) * P / float(n) ** 3
line(f"slab paint, {ppr} particles, {nloc + 2 * gl} planes (rho - mean on owned planes)",
     timeit(lambda: ops.paint(pos, None, n, L, 'cic', buf, (0 - gl) % n, nloc + 2 * gl, offset=mean, owned=(gl, nloc))))
dev.profile_enable(True)
for _ in range(5):
    ops.paint(pos, None, n, L, 'cic', buf, (0 - gl) % n, nloc + 2 * gl, offset=mean, owned=(gl, nloc))
torch.cuda.synchronize()
print("    paint kernels:", {k: round(v[1] / 5, 3) for k, v in dev.profile_report().items()}, flush=True)
dev.profile_enable(False)
owned = buf[gl:gl + nloc]
line("ghost add (2 x %d planes)" % gl, timeit(lambda: (ops.add_into(buf[gl:2 * gl], buf[:gl]), ops.add_into(buf[nloc:nloc + gl], buf[nloc + gl:]))))
line("low-k modes of the owned planes (side stream in the pipeline)", timeit(lambda: ops.lowk_modes(owned, n, 0)))
nzp = ops.spectrum_pitch(n, P) if not os.environ.get("SLAB_UNPADDED") else nz
print(f"    spectrum row pitch {nzp} (n/2+1 = {nz})")
spec2d = ops.empty((nloc, n, nzp), ops.cdtype)
chunks = 4
pc = nloc // chunks
packed = ops.empty((chunks, P, pc, nloc, nzp), ops.cdtype)
block = ops.empty((n, nloc, nzp), ops.cdtype)
def ffts():
    for c in range(chunks):
        ops.fft2d_planes_packed(owned[c * pc:(c + 1) * pc], spec2d[c * pc:(c + 1) * pc], packed[c], P, 0, block[c * pc:(c + 1) * pc])
line(f"z rows + y pass storing in send order, {chunks} chunks", timeit(ffts))
dev.profile_enable(True)
for _ in range(5):
    ffts()
torch.cuda.synchronize()
print("    fft kernels:", {k: round(v[1] / 5, 3) for k, v in dev.profile_report().items()}, flush=True)
dev.profile_enable(False)
psum = ops.zeros((n // 2 - 1,), torch.float64)
line("axis-0 pass fused with the block's shell binning", timeit(lambda: ops.fft1d_axis0_power(block, 1.0 / n ** 3, n, L, 0, psum, 5)))
print(f"{'sum (the low-k line overlaps the FFT chunks)':55s} {total:7.3f} ms")
