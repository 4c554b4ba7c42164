"""Experiment: 3D R2C plan vs (batched 1D R2C along z) + (strided 2D C2C over x,y)."""
import os, sys, ctypes as ct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrild_amd import device as dev, _lib
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nz = n // 2 + 1
def arr(v): return (ct.c_size_t * len(v))(*v)
def general(kind, rank, lengths, ins, outs, batch, idist, odist, scale, inplace):
    h = ct.c_void_p()
    _lib.check(L.ast_fft_plan_create_general(ct.byref(h), kind, 0, rank, arr(lengths), arr(ins), arr(outs), batch, idist, odist, scale, inplace))
    return h
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]
grid = torch.randn((n, n, n), device="cuda")
spec = torch.empty((n, n, nz), dtype=torch.complex64, device="cuda")
ref = dev.r2c(grid).clone()
print("3D plan:", timeit(lambda: dev.r2c(grid, out=spec)), "ms")
# (1) batched 1-D R2C along z: n*n transforms
p1 = general(0, 1, [n], [1], [1], n * n, n, nz, 1.0 / n**3, 0)
# (2) 2-D C2C over (x, y) for every kz: lengths (n, n) strides (n*nz, nz), batch nz, dist 1, in place
p2 = general(2, 2, [n, n], [n * nz, nz], [n * nz, nz], nz, 1, 1, 1.0, 1)
s = dev.stream()
def two():
    _lib.check(L.ast_fft_exec(p1, dev.ptr(grid), dev.ptr(spec), s))
    _lib.check(L.ast_fft_exec(p2, dev.ptr(spec), None, s))
print("1D r2c + strided 2D c2c:", timeit(two), "ms", " work bytes", L.ast_fft_plan_work_bytes(p1), L.ast_fft_plan_work_bytes(p2))
two(); torch.cuda.synchronize()
print("max abs diff vs 3D plan:", (spec - ref).abs().max().item(), "ref max", ref.abs().max().item())
# (3) three 1-D passes: z (r2c), y strided per... via 2D split into two strided 1-D plans with 2-level batch emulated by rank-2 trick
p3y = general(2, 1, [n], [nz], [nz], nz, 1, 1, 1.0, 1)      # one x-plane only (for timing reference)
print("single-plane strided y pass:", timeit(lambda: _lib.check(L.ast_fft_exec(p3y, dev.ptr(spec), None, s))), "ms (x", n, "planes)")
