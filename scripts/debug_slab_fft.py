"""Debug: the slab path's FFT pieces at n (emulated ranks, one GPU) vs the single-GPU r2c."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from astrild_amd import device as dev, slab
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L = 1000.0
pos = dev.synth_lattice_particles(n, n, L, seed=5, dtype=torch.float32)
full = dev.paint(pos, None, n, L, "cic")
del pos
ref = dev.finish_power(*dev.power_sums_fused(full, L, mean=1.0))
spec_ref = dev.r2c(full)
r2 = dev.finish_power(*dev.power_bin_1d(spec_ref, None, n, L))
print("fused vs unfused single GPU: max rel", np.abs(r2["power"] / ref["power"] - 1).max(), flush=True)
ops = slab.HipSlabOps(torch.float32)
nloc, nz = n // P, n // 2 + 1
packed = []
for r in range(P):
    owned = full[r * nloc:(r + 1) * nloc].contiguous()
    spec2d = ops.empty((nloc, n, nz), ops.cdtype)
    ops.fft2d_planes(owned, spec2d)
    chk = torch.fft.rfft2(owned[:2].double()).to(torch.complex64)
    print(f"rank {r} 2D fft: max abs diff (2 planes)", float((spec2d[:2] - chk).abs().max()), "scale", float(chk.abs().max()), flush=True)
    pk = ops.empty((P, nloc, nloc, nz), ops.cdtype)
    ops.pack(spec2d, pk, P)
    packed.append(pk)
    del spec2d
ps_tot = torch.zeros(n // 2 - 1, dtype=torch.float64, device="cuda")
for r in range(P):
    block = torch.stack([packed[s][r] for s in range(P)], dim=0).reshape(n, nloc, nz).contiguous()
    ops.fft1d_axis0(block, 1.0 / float(n) ** 3)
    want = spec_ref[:, r * nloc:(r + 1) * nloc, :]
    d = (block - want).abs()
    print(f"rank {r} block vs ref: max abs diff {float(d.max()):.3e} (ref max {float(want.abs().max()):.3e})", flush=True)
    ps = torch.zeros_like(ps_tot)
    ops.power_bin(block, n, L, (0, n), (r * nloc, nloc), ps)
    ps_tot += ps
ks, nm = dev.shell_geometry(n, L)
res = dev.finish_power(ks, ps_tot, nm)
print("emulated slab vs single: max rel", np.abs(res["power"] / r2["power"] - 1).max(), res["power"][:4], r2["power"][:4], ref["power"][:4])
