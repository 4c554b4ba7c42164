"""kappa stack rate of ONE variant (AST_KSTACK_DEPTH / AST_KSTACK_GRID from the environment) over N buffers allocated one after the other."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from astrild_amd import lensing
npix, nplanes = 4096, 64
wnum, wden = lensing.synth_plane_weights(nplanes)
out = torch.empty((npix, npix), dtype=torch.float64, device="cuda")
keep, rates = [], []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    buf = torch.empty((8 << 30) // 8, dtype=torch.float64, device="cuda")
    buf.fill_(1.0)
    keep.append(buf)
    planes = [buf[p * npix * npix:(p + 1) * npix * npix].view(npix, npix) for p in range(nplanes)]
    lensing.kappa_stack(planes, wnum, wden, out=out)
    b = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lensing.kappa_stack(planes, wnum, wden, out=out); e1.record(); e1.synchronize()
        b = min(b, e0.elapsed_time(e1))
    rates.append((nplanes + 1) * npix * npix * 8 / b / 1e6)
print("depth %s grid %s: " % (os.environ.get("AST_KSTACK_DEPTH", "4"), os.environ.get("AST_KSTACK_GRID", "2048")) + " ".join("%.0f" % r for r in rates), flush=True)
