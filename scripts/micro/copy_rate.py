"""Streaming rates of ast_stream_copy's tuning variants (mode bits 4-7) on the GPU box: which one is the ceiling bench.py quotes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from astrild_amd import device as dev
from astrild_amd._lib import lib, check

nbytes = int(os.environ.get("GIB", "4")) << 30
a = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").fill_(1.0)
b = torch.empty_like(a)
for tune in range(16):
    row = []
    for name, op, moved in (("copy", 0, 2 * nbytes), ("read", 1, nbytes), ("write", 2, nbytes)):
        best = 1e9
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib().ast_stream_copy(dev.ptr(b), dev.ptr(a), nbytes, op | (tune << 4) | 256, dev.stream()), "copy")
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        row.append("%s %7.1f GB/s" % (name, moved / (best * 1e6)))
    print("tune %2d (plain loads %d, plain stores %d, one piece per workgroup %d, eight per lane %d): %s"
          % (tune, tune & 1, (tune >> 1) & 1, (tune >> 2) & 1, (tune >> 3) & 1, "  ".join(row)), flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); b.copy_(a); e1.record(); e1.synchronize()
e0.record(); b.copy_(a); e1.record(); e1.synchronize()
print("torch copy_: %.1f GB/s" % (2 * nbytes / (e0.elapsed_time(e1) * 1e6)))
