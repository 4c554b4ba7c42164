"""Worker of tests/test_gpu_kappa.py::test_sharded_stack_ranks_share_one_gpu: rank r of a world of P processes,
all on cuda:0 over gloo, stacks its planes of the synthetic config-D stack with the real HipStackOps and
kappa_stack_sharded; rank 0 writes the result."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist


def main():
    rank, world, port, nplanes, npix, out = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]),
                                             int(sys.argv[5]), sys.argv[6])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        from astrild_amd import kappa_shard, lensing
        ids = kappa_shard.my_plane_ids(nplanes)
        planes = lensing.synth_kappa_planes(nplanes, npix, ids=ids)
        wnum, wden = lensing.synth_plane_weights(nplanes)
        res = kappa_shard.kappa_stack_sharded(planes, wnum[ids], wden[ids])
        res2 = kappa_shard.kappa_stack_sharded(planes, wnum[ids], wden[ids])
        if rank == 0:
            assert torch.equal(res, res2)                   # fixed summation order: bit-reproducible
            np.save(out, res.cpu().numpy())
        else:
            assert res is None
        if len(sys.argv) > 7 and sys.argv[7] == "stream":
            # a stream of 2 P + 1 maps: rotating root, the per-map stages on the root's second stream
            nmaps = 2 * world + 1
            lp, sp = lensing.lens_plan(npix, np.deg2rad(20.0)), lensing.smooth_plan(npix)
            keep = {}
            tail = lensing.kappa_map_tail(lp, sp, 3.4, keep=keep)
            stream = kappa_shard.MapStream(npix * npix)
            pend = {}
            for m in range(nmaps):
                scale = 1.0 + 0.25 * m                       # every map its own weights
                stream.push(planes, wnum[ids] * scale, wden[ids], tail)
            pend = dict(stream.finish())
            assert set(pend) == {m for m in range(nmaps) if m % world == rank}
            torch.cuda.synchronize()
            np.savez(out + f".stream{rank}.npz", **{f"pdf{m}": p.result()[0] for m, p in pend.items()},
                     **{f"map{m}": v[0].cpu().numpy() for m, v in keep.items()},
                     **{f"a1_{m}": v[1].cpu().numpy() for m, v in keep.items()})
        if len(sys.argv) > 7 and sys.argv[7] == "api":
            # what sum_snapshots / sum_raytracing_snapshots do with a SEQUENCE of source redshifts: the rank's planes stay
            # resident, one weighted map per entry, map m lands on rank m mod P
            from astrild_amd.rays.rayramses import PlaneStacker
            nmaps = world + 2
            flat = [p.reshape(-1) for p in planes]
            wl = [None if m == 1 else (list(wnum[ids] * (1.0 + 0.5 * m)), list(wden[ids])) for m in range(nmaps)]
            maps = PlaneStacker._stack_many(flat, wl, dist.group.WORLD)
            assert [m for m, v in enumerate(maps) if v is not None] == [m for m in range(nmaps) if m % world == rank]
            np.savez(out + f".api{rank}.npz", **{f"map{m}": v for m, v in enumerate(maps) if v is not None})
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
