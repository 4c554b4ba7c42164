"""Do buffers that fit the 256 MB Infinity Cache stream faster than HBM?  Elementwise add (read a, write b) and fill over
buffers of 16 MB ... 2 GB, many calls back to back.  usage: python scripts/micro/ic_ceiling.py"""
import torch
def timeit(fn, inner, reps=7):
    for _ in range(inner): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner): fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    return sorted(ts)[reps // 2]
for mb in (16, 32, 64, 128, 256, 512, 2048):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    inner = max(4, 2048 // mb)
    t_add = timeit(lambda: torch.add(a, 1.0, out=b), inner)
    t_inpl = timeit(lambda: a.add_(1.0), inner)
    t_fill = timeit(lambda: b.fill_(1.0), inner)
    gb = mb * 1.048576e-3
    print(f"{mb:5d} MB per buffer: add a->b {2 * gb / t_add:6.2f} TB/s | in place {2 * gb / t_inpl:6.2f} TB/s | fill {gb / t_fill:6.2f} TB/s", flush=True)
    del a, b
