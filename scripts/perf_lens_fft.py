"""Times the rocFFT pieces a zero-padded (2N)^2 convolution can be built from (N = 4096, double):
the full 2-D R2C / C2R plans the lens plan uses today against row transforms of only the N non-zero rows
plus strided column transforms.  Run on the GPU box: python scripts/perf_lens_fft.py"""
import ctypes as ct
import sys
import time
import torch
sys.path.insert(0, ".")
from astrild_amd import _lib, device as dev

L = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N2, NH = 2 * N, N + 1
F64 = 1
R2C, C2R, C2C_F, C2C_I = 0, 1, 2, 3


def general(kind, lengths, istr, ostr, batch, idist, odist, inplace):
    h = ct.c_void_p()
    arr = lambda v: (ct.c_size_t * len(v))(*v)
    _lib.check(L.ast_fft_plan_create_general(ct.byref(h), kind, F64, len(lengths), arr(lengths), arr(istr), arr(ostr),
                                            batch, idist, odist, 1.0, int(inplace)), "general")
    return h


def timeit(name, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{name:50s} {(time.perf_counter() - t0) / reps * 1e3:8.3f} ms", flush=True)


real_full = torch.zeros((N2, N2), dtype=torch.float64, device="cuda")
real_full[:N, :N].normal_()
spec = torch.empty((N2, NH), dtype=torch.complex128, device="cuda")
spec2 = torch.empty_like(spec)
s = dev.stream()
p_r2c = dev.fft_plan(R2C, F64, (N2, N2))
p_c2r = dev.fft_plan(C2R, F64, (N2, N2))
timeit("2-D R2C (2N)^2", lambda: p_r2c.execute(real_full, spec))
ref = spec.clone()
prod = spec.clone()
out_full = torch.empty_like(real_full)
timeit("2-D C2R (2N)^2", lambda: p_c2r.execute(prod, out_full))

# rows: N rows of 2N reals (row pitch 2N) -> N rows of N+1 complex (pitch N+1)
rows_r2c = general(R2C, [N2], [1], [1], N, N2, NH, False)
cols_fwd = general(C2C_F, [N2], [NH], [NH], NH, 1, 1, False)        # out of place: lower half of `spec` stays zero
cols_fwd_ip = general(C2C_F, [N2], [NH], [NH], NH, 1, 1, True)
cols_inv_ip = general(C2C_I, [N2], [NH], [NH], NH, 1, 1, True)
rows_c2r = general(C2R, [N2], [1], [1], N, NH, N2, False)
spec.zero_()
timeit("rows R2C, N rows", lambda: _lib.check(L.ast_fft_exec(rows_r2c, dev.ptr(real_full), dev.ptr(spec), s), "x"))
timeit("columns C2C fwd, out of place", lambda: _lib.check(L.ast_fft_exec(cols_fwd, dev.ptr(spec), dev.ptr(spec2), s), "x"))
err = (spec2 - ref).abs().max().item() / ref.abs().max().item()
print("rows + columns vs 2-D plan, relative:", err)
timeit("columns C2C fwd, in place", lambda: _lib.check(L.ast_fft_exec(cols_fwd_ip, dev.ptr(spec2), None, s), "x"))
timeit("columns C2C inv, in place", lambda: _lib.check(L.ast_fft_exec(cols_inv_ip, dev.ptr(spec2), None, s), "x"))
timeit("rows C2R, N rows", lambda: _lib.check(L.ast_fft_exec(rows_c2r, dev.ptr(spec2), dev.ptr(out_full), s), "x"))
