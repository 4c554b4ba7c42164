"""What ONE rank of P (default 8) does per map of the kappa map stream (kappa_shard.MapStream) at 64 planes x 4096^2 fp64,
timed on one GPU without the collectives: the local stack of its nplanes / P planes, the rank-ordered sum of the P chunks
it receives, and - every P-th map - the per-map stages (unit conversion, smoothing, PDF, kappa -> alpha).
    python scripts/perf_kappa_stream_pieces.py [P]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from astrild_amd import lensing

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
npix, nplanes = 4096, 64
planes = lensing.synth_kappa_planes(nplanes, npix, ids=list(range(0, nplanes, P)))
wnum, wden = lensing.synth_plane_weights(nplanes)
wnum, wden = wnum[::P], wden[::P]
n = npix * npix
chunk = (n + P - 1) // P
send = torch.zeros(P * chunk, dtype=torch.float64, device="cuda")
recv = torch.randn(P * chunk, dtype=torch.float64, device="cuda")
mine = torch.empty(chunk, dtype=torch.float64, device="cuda")
full = torch.randn(n, dtype=torch.float64, device="cuda") * 0.01
lp, sp = lensing.lens_plan(npix, np.deg2rad(20.0)), lensing.smooth_plan(npix)
tail = lensing.kappa_map_tail(lp, sp, 1.0 / 60.0 * npix / 20.0)


def timeit(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]


t_stack = timeit(lambda: lensing.kappa_stack(planes, wnum, wden, out=send[:n]))
t_chunks = timeit(lambda: lensing.kappa_stack([recv[s * chunk:(s + 1) * chunk] for s in range(P)], out=mine))
keep = []
t_tail = timeit(lambda: keep.append(tail(0, full.clone())))
for p in keep:
    p.result()
wire = n * 8 / P / 1e6
print(f"P = {P}: local stack of {len(planes)} planes {t_stack:.3f} ms; rank-ordered sum of {P} chunks {t_chunks:.3f} ms; per-map stages "
      f"(incl. a 134 MB clone) {t_tail:.3f} ms, once every {P} maps = {t_tail / P:.3f} ms per map")
print(f"    compute per map and rank {t_stack + t_chunks + t_tail / P:.3f} ms; on every link per map: {wire:.1f} MB (all-to-all) + {wire:.1f} MB (gather to the "
      f"rotating root) = {2 * wire / 60:.3f} ms at 60 GB/s, on RCCL's stream beside the next map's stack")
for B in (40, 60):
    per_map = max(t_stack + t_chunks + t_tail / P, 2 * wire / B)
    print(f"    forecast at {B} GB/s per link: {per_map:.3f} ms per map = {1e3 / per_map:.0f} maps/s")
