"""Device-to-device copy and read-only rates of this box (what a streaming pass with as many bytes written as read can reach).
usage: python scripts/micro/copy_ceiling.py"""
import torch
def timeit(fn, reps=9):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]
for gb in (1, 4, 8):
    n = gb * (1 << 30) // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    t2 = timeit(lambda: torch.add(a, 1.0, out=b))
    t3 = timeit(lambda: a.sum())
    t4 = timeit(lambda: b.zero_())
    print(f"{gb} GB: copy_ {2 * gb * 1.0737 / t:.2f} TB/s ({t:.3f} ms) | add {2 * gb * 1.0737 / t2:.2f} TB/s | sum (read only) {gb * 1.0737 / t3:.2f} TB/s | zero (write only) {gb * 1.0737 / t4:.2f} TB/s", flush=True)
    del a, b
