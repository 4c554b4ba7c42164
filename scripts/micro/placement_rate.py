"""Is the read rate of a buffer decided by where it was allocated?  N buffers of 8 GiB allocated one after the other (all
kept), each read with ast_stream_copy (mode 1), copied to itself+4GiB (mode 0) and read as 64 interleaved streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from astrild_amd import device as dev, lensing
from astrild_amd._lib import lib, check

gib = 8
nbytes = gib << 30
npix, nplanes = 4096, 64
wnum, wden = lensing.synth_plane_weights(nplanes)
out = torch.empty((npix, npix), dtype=torch.float64, device="cuda")
keep = []


def best(fn, reps=4):
    fn()
    b = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        b = min(b, e0.elapsed_time(e1))
    return b


for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    buf = torch.empty(nbytes // 8 + nplanes * 0, dtype=torch.float64, device="cuda")
    buf.fill_(1.0)
    keep.append(buf)
    rd = nbytes / best(lambda: check(lib().ast_stream_copy(dev.ptr(buf), dev.ptr(buf), nbytes, 1, dev.stream()), "r")) / 1e6
    half = nbytes // 2
    cp = 2 * half / best(lambda: check(lib().ast_stream_copy(buf.data_ptr() + half, dev.ptr(buf), half, 0, dev.stream()), "c")) / 1e6
    planes = [buf[p * npix * npix:(p + 1) * npix * npix].view(npix, npix) for p in range(nplanes)]
    st = (nplanes + 1) * npix * npix * 8 / best(lambda: lensing.kappa_stack(planes, wnum, wden, out=out)) / 1e6
    print(f"buffer {trial:2d} at {buf.data_ptr():#x} ({trial * gib:3d}-{(trial + 1) * gib:3d} GiB allocated): read {rd:6.0f}  copy {cp:6.0f}  64-stream stack {st:6.0f} GB/s", flush=True)
