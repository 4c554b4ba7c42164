"""Soak: one rank of eight (exchange stubbed) runs the staged step N times; the slab buffer, the send buffer and the shell
sums must be bit-identical from step to step (fixed-point paint, fixed-order sums): any race between the stages would
show as a difference.    python scripts/soak_slab_staged.py [steps] [rank]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from astrild_amd import slab

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
RANK = int(sys.argv[2]) if len(sys.argv) > 2 else 3
P, n, L = 8, 1024, 1000.0
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29578")
dist.init_process_group("gloo", rank=0, world_size=1)
dist.get_world_size = lambda group=None: P
dist.get_rank = lambda group=None: RANK
dist.all_reduce = lambda t, *a, **k: None
slab.exchange_planes = lambda *a, **k: []
slab.exchange_planes_disc = lambda *a, **k: []
for name in ("start", "start_upper", "start_lower"):
    setattr(slab.GhostExchange, name, lambda self: None)
def _finish(self):
    owned = self.buf[self.gl: self.gl + self.nloc]
    self.ops.add_into(owned[self.nloc - self.gl:], self.from_right)
    self.ops.add_into(owned[:self.gh], self.from_left)
slab.GhostExchange.finish = _finish
slab.comm_ready = lambda group=None: None
for streams in ("1", "2"):
    os.environ["ASTRILD_SLAB_STREAMS"] = streams
    pipe = slab.SlabPowerPipeline(n, L, n, window="cic", dtype=torch.float32, ghost=3)
    g = torch.Generator(device="cuda").manual_seed(5)
    pipe.ghosts.from_left.normal_(generator=g)
    pipe.ghosts.from_right.normal_(generator=g)
    pipe.block.zero_()
    pipe.step(check=True)
    torch.cuda.synchronize()
    ref = (pipe.buf.clone(), pipe.packed.clone(), pipe.block.clone(), pipe.psum.clone())
    bad = 0
    for i in range(steps):
        pipe.step()
        if i % 10 == 9 or i == steps - 1:
            torch.cuda.synchronize()
            now = (pipe.buf, pipe.packed, pipe.block, pipe.psum)
            same = [bool(torch.equal(torch.view_as_real(a) if a.is_complex() else a, torch.view_as_real(b) if b.is_complex() else b))
                    for a, b in zip(now, ref)]
            if not all(same):
                bad += 1
                d = (pipe.psum - ref[3]).abs()
                rel = (d / ref[3].abs().clamp_min(1e-300)).max().item()
                idx = torch.nonzero(d > 0).flatten().tolist()
                print(f"streams={streams} step {i}: differs (buf, packed, block, psum) = {same}; psum: {len(idx)} shells differ, "
                      f"first {idx[:6]}, max relative difference {rel:.3g}", flush=True)
    print(f"streams={streams}: {steps} staged steps (group parts {pipe.group_chunks}), {bad} checks differed", flush=True)
    del pipe, ref
    torch.cuda.empty_cache()
dist.destroy_process_group()
