"""Does the column walk run faster when the positions it gathers are still in the 256 MB Infinity Cache?  (The premise of
every single-read paint design that hands tile rows from the grouping to the walk through the cache.)  Staged paint at
1024^3, natural order: GROUP once, then (a) walk the 128 tile rows one launch each (cold: 100 MB of positions per row
from HBM), (b) walk ONE row 128 times (warm: its positions, lists and grid lines stay in the cache).
    AST_PAINT_ZSEG=16 python scripts/perf_walk_ic.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrild_amd import device as dev

n, L = 1024, 1000.0
for window in ("cic", "tsc"):
    pos = dev.synth_lattice_particles(n, n, L, dtype=torch.float32)
    out = torch.empty((n, n, n), dtype=torch.float32, device="cuda")
    sp = dev.StagedPaint(pos, None, n, L, window, out, offset=1.0)
    sp.group()
    torch.cuda.synchronize()

    def timed(rows, reps=3):
        best = 1e9
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for r in rows:
                sp.walk(r, 1)
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        return best

    R = sp.nrows_total
    cold = timed(list(range(R)))
    warm = timed([R // 2] * R)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); sp.walk(0, R); b.record(); torch.cuda.synchronize()
    print(f"{window}: {R} rows one launch each (cold) {cold:.3f} ms; one row {R} times (warm) {warm:.3f} ms; "
          f"all rows in one launch {a.elapsed_time(b):.3f} ms; ZSEG={os.environ.get('AST_PAINT_ZSEG')}", flush=True)
    del sp, out, pos
    torch.cuda.empty_cache()
