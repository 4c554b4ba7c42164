"""D2H of a 4096^2 float64 map (134 MB): torch's copy_ into page-locked memory (SDMA) against a kernel that writes the
page-locked buffer directly (ast_stream_copy with a host destination), and the same for H2D."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from astrild_amd import device as dev
from astrild_amd._lib import lib, check

t = torch.randn(4096, 4096, dtype=torch.float64, device="cuda")
pin = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
nbytes = t.numel() * 8


def timed(label, fn, reps=8):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{label:60s} {ms:7.3f} ms  {nbytes / ms / 1e6:6.1f} GB/s", flush=True)


timed("D2H torch copy_ into a page-locked buffer (non_blocking)", lambda: pin.copy_(t, non_blocking=True))
timed("D2H kernel writing the page-locked buffer (ast_stream_copy)", lambda: check(lib().ast_stream_copy(pin.data_ptr(), dev.ptr(t), nbytes, 0, dev.stream()), "c"))
assert torch.equal(pin, t.cpu())
d = torch.empty_like(t)
timed("H2D torch copy_ from the page-locked buffer (non_blocking)", lambda: d.copy_(pin, non_blocking=True))
timed("H2D kernel reading the page-locked buffer (ast_stream_copy)", lambda: check(lib().ast_stream_copy(dev.ptr(d), pin.data_ptr(), nbytes, 0, dev.stream()), "c"))
assert torch.equal(d, t)
for tune in (0, 3, 7, 4, 12):
    timed(f"D2H kernel, variant {tune}", lambda: check(lib().ast_stream_copy(pin.data_ptr(), dev.ptr(t), nbytes, 256 | (tune << 4), dev.stream()), "c"))
