"""Per-launch-site timing of the tiled paint via the library's HIP-event profiler."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from astrild_amd import device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
window = sys.argv[2] if len(sys.argv) > 2 else "cic"
orders = sys.argv[3].split(",") if len(sys.argv) > 3 else ["natural", "shuffled"]
L = 1000.0
ACC = os.environ.get("ACC", "0") == "1"
METHOD = os.environ.get("METHOD", "tiled")
for order in orders:
    pos = dev.synth_lattice_particles(n, n, L, shuffle=(order == "shuffled"), dtype=torch.float32,
                                      sigma_cells=float(os.environ.get("SIGMA", "0.5")))
    grid = torch.zeros((n, n, n), dtype=torch.float32, device="cuda")
    for _ in range(2):
        dev.paint(pos, None, n, L, window, out=grid, method=METHOD, check_dropped=False, accumulate=ACC)
    torch.cuda.synchronize()
    dev.profile_enable(True)
    reps = int(os.environ.get("REPS", "20"))
    for _ in range(reps):
        dev.paint(pos, None, n, L, window, out=grid, method=METHOD, check_dropped=False, accumulate=ACC)
    torch.cuda.synchronize()
    rep = dev.profile_report()
    dev.profile_enable(False)
    tot = sum(v[1] for v in rep.values()) / reps
    print(f"n={n} {window} {order}: total {tot:.3f} ms  " + "  ".join(f"{k.split('.')[-1]}={v[1]/reps:.3f}" for k, v in rep.items()), flush=True)
    del pos, grid
