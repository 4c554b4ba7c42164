"""A few per-map kappa pipelines (for rocprofv3 --kernel-trace --stats): dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from astrild_amd import lensing
r = lensing.bench_kappa_pipeline(steps=int(sys.argv[1]) if len(sys.argv) > 1 else 10, warmup=2)
print(round(r["value"], 1), "maps/s", round(r["ms_per_map"], 3), "ms", r["kernels_ms"])
