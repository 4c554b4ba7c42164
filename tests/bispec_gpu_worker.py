"""Worker of tests/test_gpu_api.py::test_sharded_bispectrum_ranks_share_one_gpu: rank r of P processes on cuda:0 over
gloo evaluates its chunk of the triangle bins with the real device estimator (bispec_shard.bispectrum_sharded)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist


def main():
    rank, world, port, n, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        from astrild_amd import bispec_shard, device as dev
        L, width = 1000.0, 4
        edges = list(range(1, n // 2 + 1, width))
        nsh = len(edges) - 1
        tri = [(i, i, i) for i in range(nsh)] + [(0, i, i) for i in range(1, nsh)] + [(i, i, min(nsh - 1, 2 * i)) for i in range(1, nsh // 2)]
        if rank == 0:
            pos = dev.synth_lattice_particles(n, n, L, seed=3, dtype=torch.float32)
            field = dev.paint(pos, None, n, L, "cic")
        else:
            field = torch.zeros((n, n, n), dtype=torch.float32, device="cuda")
        res = bispec_shard.bispectrum_sharded(field, L, edges, tri, root_has_field=True)
        if rank == 0:
            np.savez(out, B=res["B"], ntri=res["ntri"], tri=np.array(tri))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
