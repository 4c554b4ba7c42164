"""CPU, world_size 2 and 8 over gloo: the slab pipeline's collective logic (ghost fold,
pack + all-to-all transpose, strided axis-0 pass, per-rank shell binning,
all-reduce) with a numpy double for the local arithmetic, against the
single-process oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fftpower as offt, mesh as omesh

N, L, NPS = 16, 100.0, 16


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _particles(n=N, nps=NPS):
    return omesh.lattice_particles(nps, n, L, seed=7, sigma_cells=0.5)


def _worker(rank, world, port, window, out_dir, chunks=2, n=N, nps=NPS, ghost=2, pipeline="bulk", rows_per_stage=None,
            group_chunks=None, disc=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if group_chunks == 8:                 # stages of unequal size: consecutive parts of a stage are grouped in one launch
        os.environ["ASTRILD_SLAB_STAGES"] = "7|0,1,2,3|4,5|6"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from astrild_amd import slab
        from tests.slab_doubles import NumpySlabOps
        pos = _particles(n, nps)
        ppr = len(pos) // world
        mine = torch.from_numpy(np.ascontiguousarray(pos[rank * ppr:(rank + 1) * ppr]))
        ops = NumpySlabOps()
        ops.disc_geometry = disc              # (rows per block, columns per tile): the transpose in the disc layout
        pipe = slab.SlabPowerPipeline(n, L, nps, window=window, dtype=torch.float64, ghost=ghost,
                                      ops=ops, pos=mine, chunks=chunks, pipeline=pipeline,
                                      rows_per_stage=rows_per_stage, xsorted=bool(group_chunks), group_chunks=group_chunks)
        assert pipe.pipeline == pipeline and pipe.group_chunks == (group_chunks or 1)
        assert (pipe.disc is not None) == (disc is not None)
        ks, ps, nm = pipe.step(check=True)
        ks, ps, nm = pipe.step(check=True)          # a second step reuses buffers, schedule and staged paint
        owned = (pipe.buf[pipe.gl: pipe.gl + pipe.nloc] if world > 1 else pipe.buf).clone()
        if pipeline == "staged" and world > 1:
            kinds = [e[0] for e in pipe.schedule]
            if group_chunks:
                # grouping in parts: planes are transformed (and sent) before the last part has been grouped
                spans = [e[5] for e in pipe.schedule if e[0] == "group_part"]          # (consecutive parts of a stage: one launch)
                assert sum(spans) == group_chunks and len(spans) <= group_chunks and "group" not in kinds
                if group_chunks == 8:
                    assert spans == [1, 4, 2, 1]
                assert kinds.index("fft") < len(kinds) - 1 - kinds[::-1].index("group_part")
                kinds = [k.replace("ghost_start_upper", "ghost_start") for k in kinds if k != "ghost_start_lower"]
            # the ghost exchange starts before the interior is walked, and planes leave before the ghosts are waited for
            assert kinds.index("ghost_start") < len(kinds) - 1 - kinds[::-1].index("walk")
            if pipe.nloc > pipe.gl + pipe.gh:
                assert "fft" in kinds[kinds.index("ghost_start"):kinds.index("ghost_finish")]
            assert sum(e[2] for e in pipe.schedule if e[0] == "fft") == pipe.nloc
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), owned=owned.numpy(), ks=ks.numpy(), ps=ps.numpy(),
                 nm=nm.numpy(), block=pipe.block.numpy())
    finally:
        dist.destroy_process_group()


# world 8 is the driver's scaling run: 8 planes per rank, ghost 3 + 1 on each side, the interior chunks transformed and
# sent while the ghost planes travel, the edge chunks after them
@pytest.mark.parametrize("window,chunks,world,n,nps,ghost,pipeline,rps", [
    ("cic", 1, 2, N, NPS, 2, "bulk", None), ("cic", 4, 2, N, NPS, 2, "bulk", None), ("tsc", 2, 2, N, NPS, 2, "bulk", None),
    ("cic", 4, 8, 64, 32, 3, "bulk", None), ("tsc", 8, 8, 64, 32, 3, "bulk", None),
    # the staged order (group, ghost rows first, then walk -> fold -> transform -> send per stage): the double poisons
    # every plane until its tile row has been folded
    ("cic", 1, 2, N, NPS, 2, "staged", None), ("tsc", 1, 2, N, NPS, 2, "staged", 1), ("cic", 1, 2, N, NPS, 3, "staged", 2),
    ("cic", 1, 8, 64, 32, 3, "staged", None), ("tsc", 1, 8, 64, 32, 3, "staged", 1), ("tsc", 1, 4, 64, 32, 2, "staged", 3),
    # ... with the particles grouped in parts (x-ordered input; rps column = number of parts, negative): the double checks
    # that no row is walked before every part holding one of its particles is grouped, and that no part brings a
    # particle for a row already walked
    ("cic", 1, 2, 64, 64, 3, "staged", -4), ("tsc", 1, 2, 64, 64, 3, "staged", -2), ("cic", 1, 4, 128, 64, 3, "staged", -4),
    ("tsc", 1, 2, 128, 64, 3, "staged", -8)])
def test_slab_pipeline_ranks_match_single_process_oracle(tmp_path, window, chunks, world, n, nps, ghost, pipeline, rps):
    port = _free_port()
    parts = -rps if rps is not None and rps < 0 else None
    mp.spawn(_worker, args=(world, port, window, str(tmp_path), chunks, n, nps, ghost, pipeline, None if parts else rps, parts),
             nprocs=world, join=True)
    pos = _particles(n, nps)
    full = omesh.paint(pos, None, n, L, window)
    ref = offt.fftpower_1d(full, L)
    spec = offt.r2c(full)
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    nloc = n // world
    for r in range(world):
        # ghost fold reproduces the owned planes of the global paint
        np.testing.assert_allclose(res[r]["owned"], full[r * nloc:(r + 1) * nloc], rtol=1e-13, atol=1e-13)
        # transpose: rank r holds delta_k[:, r*nloc:(r+1)*nloc, :] for all kx
        np.testing.assert_allclose(res[r]["block"], spec[:, r * nloc:(r + 1) * nloc, :], rtol=1e-11, atol=1e-14)
    # all-reduced shell sums are identical on both ranks and match the oracle
    assert np.array_equal(res[0]["nm"], res[-1]["nm"]) and np.array_equal(res[0]["nm"], ref["modes"])
    np.testing.assert_allclose(res[0]["ps"], res[-1]["ps"], rtol=0, atol=0)
    np.testing.assert_allclose(res[0]["ks"] / res[0]["nm"], ref["k"], rtol=1e-13)
    np.testing.assert_allclose(res[0]["ps"] / res[0]["nm"], ref["power"].real, rtol=1e-10)


# The transpose in the DISC layout (slab.disc_layout): only the rows of every k_z tile that reach into the Nyquist disc travel,
# the k_y rows dealt to the ranks in blocks balanced by disc area.  What a rank receives is compared element by element with
# the oracle's spectrum, the shell sums with the oracle's: nothing FFTPower keeps was cut away.
@pytest.mark.parametrize("window,chunks,world,n,nps,ghost,pipeline,parts,disc", [
    ("cic", 2, 2, N, NPS, 2, "bulk", None, (2, 2)), ("tsc", 1, 2, N, NPS, 2, "staged", None, (4, 2)),
    ("cic", 4, 8, 64, 32, 3, "bulk", None, (2, 4)), ("tsc", 1, 8, 64, 32, 3, "staged", None, (2, 4)),
    ("cic", 1, 4, 128, 64, 3, "staged", 4, (8, 16)), ("cic", 1, 1, N, NPS, 2, "bulk", None, (2, 2))])
def test_slab_transpose_in_the_disc_layout(tmp_path, window, chunks, world, n, nps, ghost, pipeline, parts, disc):
    from astrild_amd import slab
    port = _free_port()
    mp.spawn(_worker, args=(world, port, window, str(tmp_path), chunks, n, nps, ghost, pipeline, None, parts, disc),
             nprocs=world, join=True)
    pos = _particles(n, nps)
    full = omesh.paint(pos, None, n, L, window)
    ref = offt.fftpower_1d(full, L)
    lay = slab.disc_layout(n, world, *disc)
    # the layout itself: every (row, tile) inside the disc exactly once, nothing outside; owners balanced
    half, tile = n // 2, disc[1]
    seen = np.zeros((n, lay["tiles"]), dtype=np.int64)
    for r in range(world):
        rr, cc = lay["rows"][r], lay["cols"][r]
        assert len(rr) == lay["S"][r] and np.all(cc % tile == np.arange(len(cc)) % tile)
        np.add.at(seen, (rr[::tile], cc[::tile] // tile), 1)
        assert sorted(set(rr // disc[0])) == lay["gk"][r]
    ky = np.where(np.arange(n) > half, np.arange(n) - n, np.arange(n))
    inside = ky[:, None] ** 2 + (tile * np.arange(lay["tiles"]))[None, :] ** 2 <= half * half
    assert np.array_equal(seen, inside.astype(np.int64))
    assert max(lay["S"]) <= 1.25 * np.mean(lay["S"]) and lay["total"] < 0.9 * n * lay["tiles"] * tile
    spec = offt.r2c(full)
    padded = np.zeros((n, n, lay["tiles"] * tile), dtype=spec.dtype)
    padded[:, :, :spec.shape[2]] = spec
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r in range(world):
        # before the last pass a rank holds the k_y-transformed planes of ITS rows; the worker saved the block as the
        # exchange left it (the double's last pass does not write back)
        got = np.fft.fft(res[r]["block"].reshape(n, -1), axis=0) / float(n) ** 3
        np.testing.assert_allclose(got, padded[:, lay["rows"][r], lay["cols"][r]], rtol=1e-11, atol=1e-14)
    assert np.array_equal(res[0]["nm"], res[-1]["nm"]) and np.array_equal(res[0]["nm"], ref["modes"])
    np.testing.assert_allclose(res[0]["ps"], res[-1]["ps"], rtol=0, atol=0)
    np.testing.assert_allclose(res[0]["ks"] / res[0]["nm"], ref["k"], rtol=1e-13)
    np.testing.assert_allclose(res[0]["ps"] / res[0]["nm"], ref["power"].real, rtol=1e-10)


def test_slab_rejects_bad_geometry():
    from astrild_amd import slab
    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from tests.slab_doubles import NumpySlabOps
        with pytest.raises(ValueError):
            slab.SlabPowerPipeline(16, 1.0, 16, ops=NumpySlabOps(), pos=torch.zeros((0, 3), dtype=torch.float64),
                                   chunks=3)
        # a single rank owns the whole periodic grid and needs no ghost zone
        pos = torch.from_numpy(_particles())
        pipe = slab.SlabPowerPipeline(N, L, NPS, dtype=torch.float64, ops=NumpySlabOps(), pos=pos, chunks=2)
        ks, ps, nm = pipe.step(check=True)
        ref = offt.fftpower_1d(omesh.paint(_particles(), None, N, L, "cic"), L)
        assert np.array_equal(nm.numpy(), ref["modes"])
        np.testing.assert_allclose(ps.numpy() / nm.numpy(), ref["power"].real, rtol=1e-10)
    finally:
        dist.destroy_process_group()


def _route_worker(rank, world, port, window, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from astrild_amd import slab
        from tests.slab_doubles import NumpySlabOps
        rng = np.random.default_rng(99)
        pos = rng.uniform(-0.2 * L, 1.2 * L, size=(4000, 3))            # unordered, some outside the box
        mine = torch.from_numpy(np.ascontiguousarray(pos[rank::world]))       # an arbitrary split over the ranks
        pipe = slab.SlabPowerPipeline(N, L, NPS, window=window, dtype=torch.float64, ops=NumpySlabOps(), pos=mine,
                                      chunks=1, route=True)
        assert pipe.gl == 1 and pipe.gh == 1
        ks, ps, nm = pipe.step(check=True)
        np.savez(os.path.join(out_dir, f"route{rank}.npz"), ps=ps.numpy(), nm=nm.numpy(), npart=len(pipe.pos),
                 xs=pipe.pos.numpy()[:, 0])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("window,world", [("cic", 2), ("tsc", 4)])
def test_particle_routing_all_to_all_v(tmp_path, window, world):
    """Unordered particles, split arbitrarily over the ranks, are routed to the slab of their base plane (count,
    group, all-to-all-v) and painted with a ghost zone of just the window's own reach; P(k) as from one process."""
    mp.spawn(_route_worker, args=(world, _free_port(), window, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(99)
    pos = rng.uniform(-0.2 * L, 1.2 * L, size=(4000, 3))
    ref = offt.fftpower_1d(omesh.paint(pos, None, N, L, window), L)
    res = [np.load(tmp_path / f"route{r}.npz") for r in range(world)]
    assert sum(int(r["npart"]) for r in res) == len(pos)
    nloc = N // world
    for r in range(world):
        s_ = res[r]["xs"] * (N / L)
        base = np.mod(np.floor(s_ if window == "cic" else s_ + 0.5), N)
        assert np.all((base >= r * nloc) & (base < (r + 1) * nloc))
        np.testing.assert_array_equal(res[r]["nm"], ref["modes"])
        np.testing.assert_allclose(res[r]["ps"] / res[r]["nm"], ref["power"].real, rtol=1e-9)


def test_watchdog_ends_a_stuck_rank_with_its_stage(tmp_path):
    """slab.Watchdog: a rank that stops making progress is ended (exit code 3, no re-exec) with its rank, the note of the
    last beat and the pipeline's host stage on stderr; a rank that keeps beating is left alone."""
    import subprocess
    import sys
    code = ("import sys, time, types; sys.path.insert(0, %r)\n"
            "from astrild_amd import slab\n"
            "wd = slab.Watchdog(timeout_s=1.0, rank=5)\n"
            "wd.watch(types.SimpleNamespace(stage_name='exchange.wait', progress_report=lambda: 'entry 7 of 21'))\n"
            "for _ in range(4):\n    time.sleep(0.5); wd.beat('warm-up')\n"
            "print('alive', flush=True)\n"
            "time.sleep(30)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert proc.returncode == 3
    assert "alive" in proc.stdout
    assert "rank 5" in proc.stderr and "warm-up" in proc.stderr and "exchange.wait" in proc.stderr and "entry 7 of 21" in proc.stderr


def _subgroup_slab_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from astrild_amd import slab
        from tests.slab_doubles import NumpySlabOps
        sub = dist.new_group([1, 2])                      # every rank creates it; global ranks 1 and 2 use it
        if rank == 0:
            return
        pos = _particles(N, NPS)
        r = dist.get_rank(sub)
        ppr = len(pos) // 2
        mine = torch.from_numpy(np.ascontiguousarray(pos[r * ppr:(r + 1) * ppr]))
        for pipeline in ("staged", "bulk"):
            pipe = slab.SlabPowerPipeline(N, L, NPS, window="cic", dtype=torch.float64, ghost=2, ops=NumpySlabOps(), pos=mine,
                                          chunks=2, group=sub, pipeline=pipeline)
            ks, ps, nm = pipe.step(check=True)
            np.savez(os.path.join(out_dir, f"sub_{pipeline}_{r}.npz"), ps=ps.numpy(), nm=nm.numpy())
    finally:
        dist.destroy_process_group()


def test_slab_pipeline_on_a_subgroup_that_does_not_start_at_global_rank_0(tmp_path):
    """the ghost exchange and the chunk exchange address their peers by GLOBAL rank (P2POp's convention) - a
    sub-group whose ranks are not 0..P-1 of the world runs the pipeline and gets the single-process spectrum."""
    mp.spawn(_subgroup_slab_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    ref = offt.fftpower_1d(omesh.paint(_particles(N, NPS), None, N, L, "cic"), L)
    for pipeline in ("staged", "bulk"):
        for r in range(2):
            res = np.load(tmp_path / f"sub_{pipeline}_{r}.npz")
            assert np.array_equal(res["nm"], ref["modes"])
            np.testing.assert_allclose(res["ps"] / res["nm"], ref["power"].real, rtol=1e-10)
