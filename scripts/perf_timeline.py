"""Phase timeline of the column deposit kernel from s_memtime stamps (needs a -DPAINT_STAMPS build:
scripts/build_variants.sh stamps -DPAINT_STAMPS; ASTRILD_HIP_LIB=astrild_amd/libastrild_hip_stamps.so)."""
import ctypes as ct, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from astrild_amd import device as dev, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
window = sys.argv[2] if len(sys.argv) > 2 else "cic"
L = 1000.0
pos = dev.synth_lattice_particles(n, n, L, shuffle=False, dtype=torch.float32)
grid = torch.zeros((n, n, n), dtype=torch.float32, device="cuda")
lib = _lib.lib()
st = np.zeros(64 * 4096, dtype=np.uint64); cn = np.zeros(64, dtype=np.uint32)
f = lib.ast_debug_stamps; f.argtypes = [ct.c_void_p, ct.c_void_p]; f.restype = ct.c_int
for rep in range(2):
    dev.paint(pos, None, n, L, window, out=grid, method="tiled", check_dropped=False, accumulate=False, defer_fold=True)
    torch.cuda.synchronize()
    f(st.ctypes.data, cn.ctypes.data)          # read + reset; keep the second (warm) run
if os.environ.get("FILL"):
    names0 = {1: "(start)", 2: "trips", 3: "barrier 1", 4: "slots", 5: "barrier 2", 6: "reserve", 7: "barrier 3", 8: "scatter", 9: "barriers 4+5"}
names = {0: "start", 1: "iter", 2: "next_batch", 3: "issue loads", 4: "deposit", 5: "pre-flush", 6: "barrier A", 7: "flush", 8: "barrier B", 9: "end"}
tot = collections.Counter(); cnt = collections.Counter(); span = []
for w in range(64):
    k = int(cn[w])
    if k < 2: continue
    v = st[w * 4096: w * 4096 + min(k, 4096)]
    t = (v >> np.uint64(8)).astype(np.int64); ids = (v & np.uint64(255)).astype(int)
    span.append(t[-1] - t[0])
    for i in range(1, len(t)):
        tot[ids[i]] += t[i] - t[i - 1]; cnt[ids[i]] += 1
print(f"{len(span)} sampled workgroups, mean lifetime {np.mean(span):.0f} ticks (s_memtime: 100 MHz constant clock if ticks look small)")
allt = sum(tot.values())
if os.environ.get("FILL"):
    names = names0
for i in sorted(tot):
    print(f"  ends at {names.get(i, i):12s}: {100 * tot[i] / allt:5.1f} % of lifetime, {cnt[i] / len(span):6.1f} per workgroup, mean {tot[i] / cnt[i]:8.0f} ticks")
