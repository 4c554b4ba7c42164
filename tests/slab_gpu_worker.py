"""Worker of tests/test_gpu_slab.py::test_two_ranks_share_one_gpu_over_gloo: rank r of a world of P
processes, all on cuda:0, runs the real SlabPowerPipeline (HipSlabOps, ghost fold, chunked exchange,
all-reduces) with the gloo backend and writes its result to a .npz file."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist


def main():
    rank, world, port, n, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    dtype = torch.float64 if sys.argv[6] == "f64" else torch.float32
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        from astrild_amd import device as dev, slab
        pipe = slab.SlabPowerPipeline(n, 1000.0, n, window=os.environ.get("SLAB_TEST_WINDOW", "cic"), dtype=dtype, seed=5, chunks=2)
        assert pipe.pipeline == os.environ.get("ASTRILD_SLAB_PIPELINE", "staged")
        if dtype == torch.float32 and n in (256, 512, 1024) and (n // (32 if n == 1024 else 16)) % world == 0:
            assert pipe.disc is not None          # the transpose travels in the disc layout (only what FFTPower keeps)
        ks, ps, nm = pipe.step(check=True)
        res = dev.finish_power(ks, ps, nm)
        if rank == 0:
            np.savez(out, k=res["k"], power=res["power"], modes=res["modes"])
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
