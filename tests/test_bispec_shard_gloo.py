"""CPU, world_size 2 and 3 over gloo: the distributed bispectrum's collective logic (triangle bins split over ranks,
one all-gather) with the oracle as the local estimator, against the single-process oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import bispectrum as ob

N, L = 12, 50.0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _field():
    f = np.random.default_rng(4).standard_normal((N, N, N))
    return f + 0.3 * f ** 2


def _triangles(nsh):
    return [(i, j, l) for i in range(nsh) for j in range(i, nsh) for l in range(j, nsh)][::-1]     # deliberately unsorted


class OracleOps:
    device = torch.device("cpu")
    calls = []

    def bispectrum(self, field, boxsize, edges, triangles):
        OracleOps.calls.append(list(triangles))
        b, ntri = ob.bispectrum_fft(field.numpy(), boxsize, np.asarray(edges), triangles)
        return {"B": b, "ntri": np.rint(ntri).astype(np.int64)}

    def tensor(self, a, dtype):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from astrild_amd import bispec_shard
        edges = ob.shell_edges(N)
        tri = _triangles(len(edges) - 1)
        field = torch.from_numpy(_field()) if rank == 0 else torch.zeros((N, N, N), dtype=torch.float64)
        res = bispec_shard.bispectrum_sharded(field, L, edges, tri, root_has_field=True, ops=OracleOps())
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), B=res["B"], ntri=res["ntri"], k=res["k"],
                 mine=np.array(OracleOps.calls[0] if OracleOps.calls else []).reshape(-1, 3))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_bispectrum_matches_single_process(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    edges = ob.shell_edges(N)
    tri = _triangles(len(edges) - 1)
    b, ntri = ob.bispectrum_fft(_field(), L, edges, tri)
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    seen = []
    for r in range(world):
        assert np.array_equal(res[r]["ntri"], np.rint(ntri).astype(np.int64))
        ok = res[r]["ntri"] > 0
        np.testing.assert_allclose(res[r]["B"][ok], b[ok], rtol=1e-12)
        seen += [tuple(t) for t in res[r]["mine"]]
    # every triangle evaluated exactly once, in sorted contiguous chunks
    assert sorted(seen) == sorted(tri) and len(seen) == len(tri)
    assert seen == sorted(tri)
    from astrild_amd.bispec_shard import split_triangles
    order, bounds = split_triangles(tri, world)
    assert max(np.diff(bounds)) - min(np.diff(bounds)) <= 1
