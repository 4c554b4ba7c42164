"""Does the kappa stack care where its 64 planes start?  ONE buffer (allocated once: fixed physical placement), the planes
as views `pad` doubles apart - and the same again in a second buffer, to tell the pad's effect from the allocation's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from astrild_amd import lensing, device as dev

npix, nplanes = 4096, 64
wnum, wden = lensing.synth_plane_weights(nplanes)
pads = [0, 16, 32, 64, 128, 256, 512, 528, 1024, 2048, 4096, 8192, 8224, 16384, 65536, 66080, 131072, 1 << 20]
out = torch.empty((npix, npix), dtype=torch.float64, device="cuda")
keep = []
for trial in range(3):
    big = torch.empty(nplanes * (npix * npix + max(pads)), dtype=torch.float64, device="cuda")
    big.normal_()
    keep.append(big)                      # (a new allocation each trial: the old ones stay, so the placement differs)
    row = []
    for pad in pads:
        planes = [big[p * (npix * npix + pad): p * (npix * npix + pad) + npix * npix].view(npix, npix) for p in range(nplanes)]
        for _ in range(2):
            lensing.kappa_stack(planes, wnum, wden, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            lensing.kappa_stack(planes, wnum, wden, out=out)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 8
        row.append((pad, ms))
    print(f"trial {trial} (buffer at {big.data_ptr():#x}): " + "  ".join(f"{pad}:{(nplanes + 1) * npix * npix * 8 / ms / 1e6:.0f}" for pad, ms in row), flush=True)
# separate allocations (what the caching allocator hands out plane by plane)
planes = [torch.randn(npix, npix, dtype=torch.float64, device="cuda") for _ in range(nplanes)]
for _ in range(2):
    lensing.kappa_stack(planes, wnum, wden, out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(8):
    lensing.kappa_stack(planes, wnum, wden, out=out)
e1.record(); e1.synchronize()
print("separate allocations: %.0f GB/s;  first pointers %s" % ((nplanes + 1) * npix * npix * 8 / (e0.elapsed_time(e1) / 8) / 1e6,
      [hex(p.data_ptr()) for p in planes[:4]]))
