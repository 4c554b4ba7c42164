"""Where device.to_numpy's time goes for a 134 MB map: the page-locked allocation, the copy, the numpy view."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from astrild_amd import device as dev

t = torch.randn(4096, 4096, dtype=torch.float64, device="cuda")


def timed(label, fn, reps=8):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    print(f"{label:70s} {(time.perf_counter() - t0) / reps * 1e3:7.3f} ms", flush=True)


timed("torch.empty(pin_memory=True), dropped at once", lambda: torch.empty(t.shape, dtype=t.dtype, pin_memory=True))
keep = []
timed("torch.empty(pin_memory=True), all kept (fresh blocks)", lambda: keep.append(torch.empty(t.shape, dtype=t.dtype, pin_memory=True)))
buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
timed("buf.copy_(t)  (blocking)", lambda: buf.copy_(t))
timed("buf.copy_(t, non_blocking=True) + synchronize", lambda: (buf.copy_(t, non_blocking=True), torch.cuda.current_stream().synchronize()))
timed("device.to_numpy(t), result dropped", lambda: dev.to_numpy(t))
res = []
timed("device.to_numpy(t), results kept", lambda: res.append(dev.to_numpy(t)))
timed("t.cpu().numpy()", lambda: t.cpu().numpy())
