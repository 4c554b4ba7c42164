"""The x-sorted chunk pipeline of the tiled paint (AST_PAINT_XSORTED) against the plain two-kernel paint (dev tool):
wall time per paint for several chunk counts, and that the grids are identical.
usage: python scripts/perf_xsorted.py [n=1024] [window=cic] [configs] [reps=5]
configs: comma-separated MB:margin:streams:zseg (chunk size in MB of positions, margin planes, 1|2 streams, z-segments)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrild_amd import device as dev

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = sys.argv[2] if len(sys.argv) > 2 else "cic"
configs = [c.split(":") for c in (sys.argv[3] if len(sys.argv) > 3 else "48:4:1:8,48:4:2:8,96:4:1:4,24:4:1:8").split(",")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
L = 1000.0
pos = dev.synth_lattice_particles(n, n, L, dtype=torch.float32)
ref = dev.paint(pos, None, n, L, w, method="tiled", accumulate=False, offset="mean")
grid = torch.empty_like(ref)


def timed(hint):
    dev.paint(pos, None, n, L, w, out=grid, method="tiled", accumulate=False, defer_fold=True, offset=1.0, hint=hint,
              check_dropped=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dev.paint(pos, None, n, L, w, out=grid, method="tiled", accumulate=False, defer_fold=True, offset=1.0, hint=hint,
                  check_dropped=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"n={n} {w}: plain paint {timed(None):.3f} ms", flush=True)
for k in configs:
    for name, v in zip(("AST_PAINT_XCHUNK_MB", "AST_PAINT_XMARGIN", "AST_PAINT_XSTREAMS", "AST_PAINT_ZSEG"), k):
        os.environ[name] = v
    st = {}
    g2 = dev.paint(pos, None, n, L, w, method="tiled", accumulate=False, offset="mean", hint="xsorted", stats=st)
    same = bool(torch.equal(g2, ref))
    del g2
    dev.profile_enable(True)
    ms = timed("xsorted")
    prof = dev.profile_report()
    dev.profile_enable(False)
    sites = {k2: round(v[1] / (reps + 1), 3) for k2, v in prof.items()}
    print(f"  xsorted {':'.join(k):>12s}: {ms:.3f} ms  identical={same}  overflow={st.get('overflow')}  sites {sites}", flush=True)
