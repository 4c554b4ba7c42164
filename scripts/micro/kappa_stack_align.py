"""kappa stack rate against the virtual-address alignment of the buffer its 64 planes live in (one allocation per trial)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from astrild_amd import lensing

npix, nplanes = 4096, 64
wnum, wden = lensing.synth_plane_weights(nplanes)
out = torch.empty((npix, npix), dtype=torch.float64, device="cuda")
keep = []


def rate(planes):
    for _ in range(2):
        lensing.kappa_stack(planes, wnum, wden, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        lensing.kappa_stack(planes, wnum, wden, out=out)
    e1.record(); e1.synchronize()
    return (nplanes + 1) * npix * npix * 8 / (e0.elapsed_time(e1) / 8) / 1e6


def align_of(ptr):
    a = 1
    while ptr % (a * 2) == 0 and a < (1 << 40):
        a *= 2
    return a


for trial in range(14):
    extra = (trial % 7) * (3 << 20) + (trial // 7) * (1 << 20)             # (different sizes: different places)
    big = torch.empty(nplanes * npix * npix + extra, dtype=torch.float64, device="cuda")
    big.normal_()
    keep.append(big)
    planes = [big[p * npix * npix:(p + 1) * npix * npix].view(npix, npix) for p in range(nplanes)]
    r0 = rate(planes)
    # the same buffer, planes shifted to the next 1-GiB boundary of the VIRTUAL address if there is room
    print(f"trial {trial:2d}: ptr {big.data_ptr():#x} aligned to {align_of(big.data_ptr()) >> 20} MiB: {r0:.0f} GB/s", flush=True)
