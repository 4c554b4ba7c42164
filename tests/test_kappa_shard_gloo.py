"""CPU, world_size 2 and 3 over gloo: the sharded kappa-plane stack's collective logic (plane p on rank p mod P,
local partial sums, one all-to-all of map chunks, rank-ordered chunk sums, gather) with a numpy double for
the local arithmetic, against the sequential oracle sum; and RayRamses / SimulationCollection on top of it."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import kappa as ok

NPIX = 37            # 1369 pixels: not divisible by 2 or 3 -> exercises the chunk padding


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _planes(nplanes):
    rng = np.random.default_rng(77)
    return [rng.standard_normal((NPIX, NPIX)) * 10.0 ** rng.integers(-3, 3) for _ in range(nplanes)]


def _weights(nplanes):
    mid = (np.arange(nplanes) + 0.5) * (1000.0 / nplanes)
    from astrild_amd.lensing import translate_redshift_weights
    return translate_redshift_weights(mid - 3.0, mid + 3.0, 1100.0, 700.0)


def _worker(rank, world, port, nplanes, weighted, all_ranks, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from astrild_amd import kappa_shard
        from tests.kappa_doubles import NumpyStackOps
        planes = _planes(nplanes)
        ids = kappa_shard.my_plane_ids(nplanes)
        assert ids == list(range(rank, nplanes, world))
        wn = wd = None
        if weighted:
            wnum, wden = _weights(nplanes)
            wn, wd = wnum[ids], wden[ids]
        res = kappa_shard.kappa_stack_sharded([planes[i] for i in ids], wn, wd, all_ranks=all_ranks, ops=NumpyStackOps())
        if res is not None:
            np.save(os.path.join(out_dir, f"rank{rank}.npy"), res.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nplanes,weighted,all_ranks", [(2, 7, False, False), (2, 8, True, True),
                                                                (3, 7, True, False), (3, 2, False, True)])
def test_sharded_stack_matches_sequential_sum(tmp_path, world, nplanes, weighted, all_ranks):
    mp.spawn(_worker, args=(world, _free_port(), nplanes, weighted, all_ranks, str(tmp_path)), nprocs=world, join=True)
    planes = _planes(nplanes)
    if weighted:
        wnum, wden = _weights(nplanes)
        seq = None
        for p, pl in enumerate(planes):
            q = pl * wnum[p] / wden[p]
            seq = q.copy() if seq is None else seq + q
    else:
        seq = ok.kappa_stack(planes)
    # the same sum, re-associated as (rank 0's planes) + (rank 1's planes) + ...: exact expectation
    expect = None
    for r in range(world):
        part = None
        for p in range(r, nplanes, world):
            q = planes[p] * wnum[p] / wden[p] if weighted else planes[p]
            part = q.copy() if part is None else part + q
        if part is None:
            part = np.zeros_like(planes[0])
        expect = part.copy() if expect is None else expect + part
    ranks = range(world) if all_ranks else [0]
    for r in ranks:
        got = np.load(tmp_path / f"rank{r}.npy").reshape(NPIX, NPIX)
        assert np.array_equal(got, expect)                                   # fixed summation order: bit-exact
        # vs the single-process running sum: re-association only
        np.testing.assert_allclose(got, seq, rtol=0, atol=world * 8 * 2.0 ** -53 * np.abs(np.stack(planes)).sum(axis=0).max())
    if not all_ranks:
        assert not any((tmp_path / f"rank{r}.npy").exists() for r in range(1, world))


def _api_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pandas as pd
        from astrild_amd import kappa_shard
        from astrild_amd.rays import rayramses
        from tests.kappa_doubles import NumpyStackOps
        # the API path with the numpy double in place of the device ops
        kappa_shard.HipStackOps = NumpyStackOps
        rayramses.as_device = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
        table, frames = _ray_table(), _ray_frames()
        loaded = []

        class MemRay(rayramses.RayRamses):
            def _load_ray_map(self, ray_file):
                box = int(ray_file.split("box")[1].split("/")[0])
                ray = int(ray_file.split("output")[1].split(".")[0])
                loaded.append((box, ray))
                return frames[(box, ray)].copy()

        rr = MemRay({"lc": "/lc/"}, ray_info_df=table)
        out = rr.sum_snapshots(None, ["kappa_2", "isw_rs"], [], {"z": [], "box": [0], "ray": [0]})
        np.savez(os.path.join(out_dir, f"api{rank}.npz"), kappa=out["kappa_2"].values, isw=out["isw_rs"].values,
                 loaded=np.array(loaded))
    finally:
        dist.destroy_process_group()


def _ray_table():
    import pandas as pd
    idx = pd.MultiIndex.from_tuples([(1, 1), (1, 2), (1, 3), (2, 1), (2, 2)], names=["box_nr", "snap_nr"])
    return pd.DataFrame({"redshift": [0.05, 0.10, 0.15, 0.20, 0.25]}, index=idx)


def _ray_frames():
    import pandas as pd
    out = {}
    for (b, r) in _ray_table().index:
        rng = np.random.default_rng(10 * b + r)
        out[(b, r)] = pd.DataFrame({"kappa_2": rng.standard_normal(NPIX * NPIX), "isw_rs": rng.standard_normal(NPIX * NPIX)})
    return out


def test_rayramses_sum_snapshots_shards_planes_over_ranks(tmp_path):
    world = 2
    mp.spawn(_api_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    frames, order = _ray_frames(), list(_ray_table().index)
    res = [np.load(tmp_path / f"api{r}.npz") for r in range(world)]
    for r in range(world):
        # rank r loaded only its planes: p = r, r + P, ...
        assert [tuple(x) for x in res[r]["loaded"]] == order[r::world]
    for col, key in (("kappa_2", "kappa"), ("isw_rs", "isw")):
        seq = ok.kappa_stack([frames[s][col].values for s in order])
        for r in range(world):
            np.testing.assert_allclose(res[r][key], seq, rtol=0, atol=1e-14 * np.abs(seq).max() + 1e-15)
        assert np.array_equal(res[0][key], res[1][key])                       # every rank holds the same bits


def _subgroup_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from astrild_amd import kappa_shard
        from tests.kappa_doubles import NumpyStackOps
        sub = dist.new_group([1, 2])                      # every rank creates it; ranks 1 and 2 use it
        if rank == 0:
            return
        planes = _planes(5)
        ids = kappa_shard.my_plane_ids(5, sub)            # group rank 0 = global rank 1
        res = kappa_shard.kappa_stack_sharded([planes[i] for i in ids], group=sub, root=0, ops=NumpyStackOps())
        assert (res is not None) == (rank == 1)            # the group's root is GLOBAL rank 1
        if res is not None:
            np.save(os.path.join(out_dir, "sub.npy"), res.numpy())
    finally:
        dist.destroy_process_group()


def test_sharded_stack_on_a_subgroup_that_does_not_start_at_global_rank_0(tmp_path):
    mp.spawn(_subgroup_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    planes = _planes(5)
    expect = (planes[0] + planes[2] + planes[4]) + (planes[1] + planes[3])
    got = np.load(tmp_path / "sub.npy").reshape(NPIX, NPIX)
    np.testing.assert_array_equal(got, expect)


def _stream_worker(rank, world, port, nplanes, nmaps, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from astrild_amd import kappa_shard
        from tests.kappa_doubles import NumpyStackOps
        planes = _planes(nplanes)
        ids = kappa_shard.my_plane_ids(nplanes)
        stream = kappa_shard.MapStream(NPIX * NPIX, ops=NumpyStackOps())
        got = {}
        for m in range(nmaps):
            # every map of the stream has its own weights (the source plane moves): a mixed-up buffer would show
            wn = np.linspace(0.5, 1.5, nplanes) * (m + 1)
            wd = np.full(nplanes, 2.0)
            stream.push([planes[i] for i in ids], wn[ids], wd[ids], tail=lambda mm, t: (mm, t.clone()))
            assert stream.root_of(m) == m % world
            # a lag of one map: map m is finished by push(m + 1)
            assert set(stream.results) == {q for q in range(m) if q % world == rank}
        res = stream.finish()
        assert set(res) == {q for q in range(nmaps) if q % world == rank}
        for m, (mm, t) in res.items():
            assert mm == m
            got[m] = t.numpy()
        np.savez(os.path.join(out_dir, f"stream{rank}.npz"), **{str(k): v for k, v in got.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nplanes,nmaps", [(2, 7, 5), (3, 4, 7)])
def test_map_stream_rotates_the_root_and_keeps_maps_apart(tmp_path, world, nplanes, nmaps):
    """kappa_shard.MapStream: map m is reduced onto rank m mod P (rotating root, rotating result buffers) and handed to
    the tail there; every map equals the rank-ordered sum of the ranks' weighted partial sums, bit for bit."""
    mp.spawn(_stream_worker, args=(world, _free_port(), nplanes, nmaps, str(tmp_path)), nprocs=world, join=True)
    planes = _planes(nplanes)
    res = [np.load(tmp_path / f"stream{r}.npz") for r in range(world)]
    for m in range(nmaps):
        wn = np.linspace(0.5, 1.5, nplanes) * (m + 1)
        expect = None
        for r in range(world):
            part = None
            for p in range(r, nplanes, world):
                q = planes[p] * wn[p] / 2.0
                part = q.copy() if part is None else part + q
            if part is None:
                part = np.zeros_like(planes[0])
            expect = part.copy() if expect is None else expect + part
        for r in range(world):
            assert (str(m) in res[r].files) == (r == m % world)
        np.testing.assert_array_equal(res[m % world][str(m)].reshape(NPIX, NPIX), expect)
