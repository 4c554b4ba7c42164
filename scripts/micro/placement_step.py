"""Does the headline step depend on where its buffers were allocated?  The same power_leg several times in one process, with
a dummy allocation of growing size kept in between (so that particles, grid, workspace and scratch land elsewhere each time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from astrild_amd import device as dev

keep = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    dev._power_scratch.clear()
    torch.cuda.empty_cache()
    lg = bench.power_leg(dev, 1024, 1024, 1000.0, "cic", "natural", "f32", "auto", steps=10, warmup=3)
    st = lg["roofline"]["stages"]
    print(f"trial {trial}: {sum(t.numel() * t.element_size() for t in keep) / 2**30:5.1f} GiB held elsewhere: step {lg['ms_per_step']:.3f} ms  paint {st['paint']['ms']:.3f} "
          f"({st['paint']['kernels']})  fft {st['fft']['ms']:.3f}", flush=True)
    dev._power_scratch.clear()
    torch.cuda.empty_cache()
    keep.append(torch.empty((5 + 3 * trial) << 28, dtype=torch.float32, device="cuda"))       # 5, 8, 11, ... GiB
