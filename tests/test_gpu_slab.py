"""GPU: every HIP piece of the slab path (slab paint with ghost planes, batched 2D
R2C, ast_slab_pack, strided axis-0 C2C, block shell binning) on ONE GPU, with the
ranks emulated in-process and the all-to-all done by tensor copies.  Compared with
the single-GPU 3D pipeline and the oracle."""
import numpy as np
import pytest

from oracle import fftpower as offt, mesh as omesh

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("P,dtype,tol,n", [(2, torch.float64, 1e-10, 64), (4, torch.float64, 1e-10, 64),
                                           (4, torch.float32, 2e-5, 64), (4, torch.float32, 2e-5, 256)])
def test_emulated_slab_ranks_match_single_gpu(hip, P, dtype, tol, n):
    # n = 256 fp32 goes through the hand-written tile FFT passes, n = 64 through rocFFT plans
    from astrild_amd import device as dev, slab
    torch.cuda.set_device(0)
    L, window, ghost = 500.0, "cic", 3
    npdt = np.float64 if dtype == torch.float64 else np.float32
    pos = omesh.lattice_particles(n, n, L, seed=11, dtype=npdt)
    ops = slab.HipSlabOps(dtype)
    nloc, nz = n // P, n // 2 + 1
    gl = gh = ghost + 1
    ppr = len(pos) // P
    bufs = []
    for r in range(P):
        buf = ops.empty((nloc + gl + gh, n, n))
        ops.paint(dev.as_device(np.ascontiguousarray(pos[r * ppr:(r + 1) * ppr])), None, n, L, window, buf,
                  (r * nloc - gl) % n, nloc + gl + gh, check=True)
        bufs.append(buf)
    # ghost fold by hand (what slab.ghost_fold does with send/recv)
    owned = [b[gl:gl + nloc].clone() for b in bufs]
    for r in range(P):
        ops.add_into(owned[(r - 1) % P][nloc - gl:], bufs[r][:gl].contiguous())
        ops.add_into(owned[(r + 1) % P][:gh], bufs[r][gl + nloc:].contiguous())
    full = torch.cat(owned, dim=0)
    ref_grid = omesh.paint(pos, None, n, L, window)
    np.testing.assert_allclose(full.cpu().numpy(), ref_grid, rtol=0, atol=(1e-12 if dtype == torch.float64 else 3e-6))

    packed = []
    for r in range(P):
        spec2d = ops.empty((nloc, n, nz), ops.cdtype)
        ops.fft2d_planes(owned[r], spec2d)
        pk = ops.empty((P, nloc, nloc, nz), ops.cdtype)
        ops.pack(spec2d, pk, P)
        packed.append(pk)
    psum_total = torch.zeros(n // 2 - 1, dtype=torch.float64, device="cuda")
    ksum_total = torch.zeros_like(psum_total)
    nm_total = torch.zeros(n // 2 - 1, dtype=torch.int64, device="cuda")
    spec_ref = dev.r2c(full)
    for r in range(P):
        block = torch.stack([packed[s][r] for s in range(P)], dim=0).reshape(n, nloc, nz).contiguous()   # the all-to-all
        ops.fft1d_axis0(block, 1.0 / float(n) ** 3)
        torch.testing.assert_close(block, spec_ref[:, r * nloc:(r + 1) * nloc, :].contiguous(),
                                   rtol=0, atol=(1e-14 if dtype == torch.float64 else 3e-7))
        ps = torch.zeros_like(psum_total)
        ops.power_bin(block, n, L, (0, n), (r * nloc, nloc), ps)
        ks, nm = ops.shell_geometry(n, L, (0, n), (r * nloc, nloc))
        psum_total += ps
        ksum_total += ks
        nm_total += nm
    res = dev.finish_power(ksum_total, psum_total, nm_total)
    ref = offt.fftpower_1d(ref_grid, L)
    assert np.array_equal(res["modes"], ref["modes"])
    np.testing.assert_allclose(res["k"], ref["k"], rtol=1e-12)
    np.testing.assert_allclose(res["power"], ref["power"].real, rtol=tol, atol=tol * ref["power"].real.max())


def test_slab_pack_unpack_round_trip(hip):
    from astrild_amd import device as dev, _lib
    torch.cuda.set_device(0)
    n0, n1, n2, parts = 6, 12, 9, 4
    x = torch.randn((n0, n1, n2), dtype=torch.complex64, device="cuda")
    packed = torch.empty((parts, n0, n1 // parts, n2), dtype=torch.complex64, device="cuda")
    _lib.check(hip.ast_slab_pack(dev.ptr(x), dev.ptr(packed), 0, n0, n1, n2, parts, dev.stream()))
    ref = x.reshape(n0, parts, n1 // parts, n2).permute(1, 0, 2, 3).contiguous()
    assert torch.equal(packed, ref)
    back = torch.empty_like(x)
    _lib.check(hip.ast_slab_unpack(dev.ptr(packed), dev.ptr(back), 0, n0, n1, n2, parts, dev.stream()))
    assert torch.equal(back, x)
    assert hip.ast_slab_pack(dev.ptr(x), dev.ptr(packed), 0, n0, n1, n2, 5, dev.stream()) < 0     # 12 % 5 != 0


@pytest.mark.parametrize("n,nplanes,parts,pad", [(256, 3, 4, 0), (512, 2, 8, 0), (256, 2, 1, 0), (1024, 2, 8, 0),
                                                  (1024, 3, 8, 15), (1024, 2, 32, 15), (512, 3, 4, 7), (256, 2, 16, 15)])
def test_y_pass_with_fused_pack_equals_y_pass_then_pack(hip, n, nplanes, parts, pad):
    """ast_fft_tile_c2c_packed == ast_fft_tile_c2c followed by ast_slab_pack, bit for bit; the rank's own piece lands
    in its separate destination and its slot of the send buffer stays untouched.  pad > 0: rows pitched to whole
    128-byte lines (n = 1024: the 32-column tiles); the padding columns are neither read (NaN there) nor written."""
    from astrild_amd import device as dev, _lib
    torch.cuda.set_device(0)
    nz = n // 2 + 1
    pitch = nz + pad
    g = torch.Generator(device="cuda").manual_seed(n + parts)
    x = torch.view_as_complex(torch.randn((nplanes, n, nz, 2), dtype=torch.float32, device="cuda", generator=g))
    ref = x.clone()
    _lib.check(hip.ast_fft_tile_c2c(dev.ptr(ref), 0, n, nz, nz, nplanes, n * nz, 1.0, dev.stream()))
    ref_packed = torch.empty((parts, nplanes, n // parts, nz), dtype=torch.complex64, device="cuda")
    _lib.check(hip.ast_slab_pack(dev.ptr(ref), dev.ptr(ref_packed), 0, nplanes, n, nz, parts, dev.stream()))
    xp = torch.full((nplanes, n, pitch), float("nan"), dtype=torch.complex64, device="cuda")
    xp[:, :, :nz] = x
    keep = xp.clone()
    packed = torch.full((parts, nplanes, n // parts, pitch), 5.0, dtype=torch.complex64, device="cuda")
    _lib.check(hip.ast_fft_tile_c2c_packed(dev.ptr(xp), dev.ptr(packed), 0, n, nz, pitch, nplanes, parts, -1, None, 1.0, dev.stream()))
    assert torch.equal(torch.view_as_real(xp).nan_to_num(nan=-1.0), torch.view_as_real(keep).nan_to_num(nan=-1.0))
    assert torch.equal(packed[..., :nz], ref_packed)
    assert bool(torch.all(packed[..., nz:] == 5.0))
    me = parts - 1
    packed2 = torch.full_like(packed, 7.0)
    mine = torch.full((nplanes, n // parts, pitch), 9.0, dtype=torch.complex64, device="cuda")
    _lib.check(hip.ast_fft_tile_c2c_packed(dev.ptr(xp), dev.ptr(packed2), 0, n, nz, pitch, nplanes, parts, me, dev.ptr(mine), 1.0,
                                           dev.stream()))
    assert torch.equal(mine[..., :nz], ref_packed[me]) and bool(torch.all(mine[..., nz:] == 9.0))
    assert torch.all(packed2[me] == 7.0)
    for s in range(parts):
        if s != me:
            assert torch.equal(packed2[s][..., :nz], ref_packed[s])
    assert hip.ast_fft_tile_c2c_packed(dev.ptr(xp), dev.ptr(packed), 0, n, nz, pitch, nplanes, 3, -1, None, 1.0, dev.stream()) < 0
    assert hip.ast_fft_tile_c2c_packed(dev.ptr(xp), dev.ptr(packed), 0, n, nz, nz - 1, nplanes, parts, -1, None, 1.0, dev.stream()) < 0
    if parts == 1:
        only = torch.empty_like(mine)
        _lib.check(hip.ast_fft_tile_c2c_packed(dev.ptr(xp), None, 0, n, nz, pitch, nplanes, 1, 0, dev.ptr(only), 1.0, dev.stream()))
        assert torch.equal(only[..., :nz], ref_packed[0])


def test_slab_pipeline_object_on_one_gpu_over_nccl(hip):
    """The real SlabPowerPipeline (HipSlabOps, chunked exchange, all-reduces) with
    torch.distributed's nccl (= RCCL) backend at world_size 1, against the single-GPU path."""
    import os
    import socket
    import torch.distributed as dist
    from astrild_amd import device as dev, slab
    torch.cuda.set_device(0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n, L = 256, 1000.0
        pipe = slab.SlabPowerPipeline(n, L, n, window="cic", dtype=torch.float32, seed=5, chunks=4)
        ks, ps, nm = pipe.step(check=True)
        res = dev.finish_power(ks, ps, nm)
        pos = dev.synth_lattice_particles(n, n, L, seed=5, dtype=torch.float32)
        ref = dev.paint_power_1d(pos, None, n, L, "cic")
        assert np.array_equal(res["modes"], ref["modes"])
        np.testing.assert_allclose(res["k"], ref["k"], rtol=1e-12)
        np.testing.assert_allclose(res["power"], ref["power"], rtol=2e-6)
        # the staged order on the whole periodic grid (no ghosts): same bits in the buffer, same spectrum
        assert pipe.pipeline == "bulk"
        keep = pipe.buf.clone()
        os.environ["ASTRILD_SLAB_DEFER_FOLD"] = "0"           # every row folded by the paint: the buffer is the complete grid
        try:
            pipe2 = slab.SlabPowerPipeline(n, L, n, window="cic", dtype=torch.float32, seed=5, pipeline="staged", rows_per_stage=5)
            ks2, ps2, nm2 = pipe2.step(check=True)
        finally:
            del os.environ["ASTRILD_SLAB_DEFER_FOLD"]
        assert torch.equal(pipe2.buf, keep)
        res2 = dev.finish_power(ks2, ps2, nm2)
        np.testing.assert_allclose(res2["power"], res["power"], rtol=1e-12)
        # default: the halo records are folded by the z pass as it loads the planes (same additions in the same order): the
        # buffer itself stays unfolded, the shell sums are the same bits
        ps2 = ps2.clone()
        pipe3 = slab.SlabPowerPipeline(n, L, n, window="cic", dtype=torch.float32, seed=5, pipeline="staged", rows_per_stage=5)
        ks3, ps3, nm3 = pipe3.step(check=True)
        assert pipe3._defer_fold and not torch.equal(pipe3.buf, keep)
        assert torch.equal(ps3, ps2)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,dt,rtol,pipeline,window", [
    (2, 64, "f64", 1e-9, "staged", "cic"), (4, 256, "f32", 2e-6, "staged", "cic"), (3, 384, "f64", 1e-9, "staged", "tsc"),
    (2, 64, "f64", 1e-9, "bulk", "cic"), (4, 256, "f32", 2e-6, "bulk", "tsc"),
    # config C rehearsal at its stated size: the 1024^3 problem on 2 / 4 ranks (fp32 slabs hold rho - mean, lowest
    # shells from the low-k channel), in the staged order (the default) and once in the bulk order
    (2, 1024, "f32", 2e-6, "staged", "cic"), (4, 1024, "f32", 2e-6, "staged", "cic"), (4, 1024, "f32", 2e-6, "bulk", "cic")])
def test_ranks_share_one_gpu_over_gloo(hip, tmp_path, world, n, dt, rtol, pipeline, window):
    """The real multi-rank data flow (staged paint, ghost exchange, per-stage transform + exchange, all-reduces) with
    HipSlabOps: `world` processes, all on cuda:0, gloo instead of RCCL (one GPU here), against the single-GPU path."""
    import os
    import socket
    import subprocess
    import sys
    from astrild_amd import device as dev
    torch.cuda.set_device(0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.npz")
    worker = os.path.join(os.path.dirname(__file__), "slab_gpu_worker.py")
    env = dict(os.environ, ASTRILD_SLAB_PIPELINE=pipeline, SLAB_TEST_WINDOW=window)
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(n), out, dt], env=env)
             for r in range(world)]
    codes = [p.wait(timeout=900) for p in procs]
    assert codes == [0] * world
    got = np.load(out)
    dtype = torch.float64 if dt == "f64" else torch.float32
    pos = dev.synth_lattice_particles(n, n, 1000.0, seed=5, dtype=dtype)
    ref = dev.paint_power_1d(pos, None, n, 1000.0, window)       # fp32: rho - mean grid, fused FFT + low-k channel
    assert np.array_equal(got["modes"], ref["modes"])
    np.testing.assert_allclose(got["k"], ref["k"], rtol=1e-12)
    np.testing.assert_allclose(got["power"], ref["power"], rtol=rtol)


@pytest.mark.parametrize("window,dtype", [("cic", torch.float32), ("tsc", torch.float64)])
def test_route_kernels_group_particles_by_destination_slab(hip, window, dtype):
    """ast_route_count / ast_route_scatter against numpy: counts exact, every particle lands in its slab's range
    exactly once (positions and masses carried along), including positions outside the box."""
    from astrild_amd import device as dev, slab
    from tests.slab_doubles import NumpySlabOps
    torch.cuda.set_device(0)
    rng = np.random.default_rng(21)
    n, L, parts, npart = 64, 100.0, 8, 300001
    pos = rng.uniform(-1.5 * L, 2.5 * L, size=(npart, 3))
    pos[:5, 0] = [0.0, L, -L, L * (1 - 2.0 ** -30), 0.5 * L / n]
    pos = pos.astype(np.float32 if dtype == torch.float32 else np.float64)
    mass = rng.uniform(1, 2, size=npart).astype(pos.dtype)
    ops = slab.HipSlabOps(dtype)
    tp, tm = dev.as_device(pos), dev.as_device(mass)
    counts = ops.route_count(tp, n, L, window, parts)
    ref = NumpySlabOps()
    want = ref.route_count(torch.from_numpy(pos.astype(np.float64)), n, L, window, parts)
    dest = ref._dest(torch.from_numpy(pos.astype(np.float64)), n, L, window, parts)
    assert np.array_equal(counts.cpu().numpy(), want.numpy())
    spos, smass = ops.route_scatter(tp, tm, n, L, window, parts, counts)
    spos, smass = spos.cpu().numpy(), smass.cpu().numpy()
    bounds = np.concatenate([[0], np.cumsum(want.numpy())])
    for p in range(parts):
        seg = slice(bounds[p], bounds[p + 1])
        got = np.concatenate([spos[seg], smass[seg, None]], axis=1)
        exp = np.concatenate([pos[dest == p], mass[dest == p, None]], axis=1)
        assert np.array_equal(got[np.lexsort(got.T)], exp[np.lexsort(exp.T)])


@pytest.mark.parametrize("ghost", [3, 4])
def test_one_rank_of_eight_at_1024_local_pieces(hip, ghost):
    """World 8 is what the driver's scaling run uses and cannot be rehearsed with 8 GPU processes on this box (6 at most):
    the LOCAL pieces of rank 3 of 8 at the 1024^3 shape in one process - slab buffer of 136 planes (ghost 3: whole tiles)
    or 138 (ghost 4: partial last tile), z-segmented walk, y pass storing in send order for 8 parts, axis-0 pass of the
    (1024, 128, 513) block fused with its shell binning - each against the plain route."""
    from astrild_amd import device as dev, slab
    torch.cuda.set_device(0)
    n, L, P, r = 1024, 1000.0, 8, 3
    nloc, nz = n // P, n // 2 + 1
    gl = ghost + 1
    ops = slab.HipSlabOps(torch.float32)
    ppr = n ** 3 // P
    pos = ops.synth(n, n, L, 5, False, r * ppr, ppr)
    x_start, nx_alloc = (r * nloc - gl) % n, nloc + 2 * gl
    buf = ops.empty((nx_alloc, n, n))
    mean = 1.0
    ops.paint(pos, None, n, L, "cic", buf, x_start, nx_alloc, check=True, offset=mean, owned=(gl, nloc))
    # the same particles into the whole periodic grid (8 x 8 x 32 tiles at another alignment, one segment per column,
    # another fixed-point quantum - it follows the tile capacity - and rho - mean formed after the rounding instead of
    # before it): the planes agree to an ulp of the largest cell
    full = dev.paint(pos, None, n, L, "cic", method="tiled", accumulate=False)
    ref = full[x_start:x_start + nx_alloc].clone()
    ref[gl:gl + nloc] -= mean
    assert float((buf - ref).abs().max()) <= 2.4e-7 * float(full.max())
    assert float(buf[:gl].sum(dtype=torch.float64) + buf[gl + nloc:].sum(dtype=torch.float64)) > 0.0      # ghosts are used
    del full, ref, pos
    owned = buf[gl:gl + nloc]
    # y pass in send order, 8 parts
    spec = ops.empty((nloc, n, nz), ops.cdtype)
    want = ops.empty((P, nloc, nloc, nz), ops.cdtype)
    ops.fft2d_planes(owned, spec)
    ops.pack(spec, want, P)
    packed = torch.full_like(want, 3.0)
    mine = ops.empty((nloc, nloc, nz), ops.cdtype)
    ops.fft2d_planes_packed(owned, spec, packed, P, r, mine)
    assert torch.equal(mine, want[r])
    for s in range(P):
        assert torch.equal(packed[s], want[s]) if s != r else bool(torch.all(packed[s] == 3.0))
    del packed, want, spec, mine, buf
    # axis-0 pass + binning of a block
    g = torch.Generator(device="cuda").manual_seed(3)
    block = torch.view_as_complex(torch.randn((n, nloc, nz, 2), generator=g, device="cuda", dtype=torch.float32))
    psum = ops.zeros((n // 2 - 1,), torch.float64)
    ops.fft1d_axis0_power(block.clone(), 1.0 / float(n) ** 3, n, L, r * nloc, psum, 0)
    blk = ops.fft1d_axis0(block, 1.0 / float(n) ** 3)
    psum2 = ops.zeros((n // 2 - 1,), torch.float64)
    ops.power_bin(blk, n, L, (0, n), (r * nloc, nloc), psum2)
    np.testing.assert_allclose(psum.cpu().numpy(), psum2.cpu().numpy(), rtol=2e-6)


@pytest.mark.parametrize("rank", [0, 3, 7])
def test_one_rank_of_eight_staged_step_with_grouping_in_parts(hip, monkeypatch, rank):
    """The driver's 8-GPU geometry (1024^3, 128 planes per rank, ghost 3 -> 136-plane buffer, particles grouped in eight
    parts over four stages, the last part first) for ONE rank in one process: the exchanges are stubbed out, everything the rank computes is real.
    No particle may turn up for a tile row that was walked already (check=True: the dropped counter stays zero, also for
    the ranks that hold the periodic wrap), and the slab buffer must equal the one-call paint of the same particles."""
    import torch.distributed as dist
    from astrild_amd import device as dev, slab
    torch.cuda.set_device(0)
    n, L, P = 1024, 1000.0, 8
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: P)
    monkeypatch.setattr(dist, "get_rank", lambda group=None: rank)
    monkeypatch.setattr(dist, "all_reduce", lambda t, *a, **k: None)
    monkeypatch.setattr(slab, "exchange_planes", lambda *a, **k: [])
    monkeypatch.setattr(slab, "exchange_planes_disc", lambda *a, **k: [])
    monkeypatch.setattr(slab, "comm_ready", lambda group=None: None)
    for name in ("start", "start_upper", "start_lower"):
        monkeypatch.setattr(slab.GhostExchange, name, lambda self: None)
    monkeypatch.setattr(slab.GhostExchange, "finish", lambda self: None)          # (no neighbours: nothing is added)
    monkeypatch.setenv("ASTRILD_SLAB_DEFER_FOLD", "0")       # every tile row folded by the paint: the buffer is the complete slab
    pipe = slab.SlabPowerPipeline(n, L, n, window="cic", dtype=torch.float32, ghost=3, seed=20240601)
    assert pipe.pipeline == "staged" and pipe.group_chunks == 8 and pipe.nx_alloc == 136
    sched = pipe._make_schedule(pipe.ops.staged_paint(pipe.pos, None, n, L, "cic", pipe.buf, pipe.x_start, pipe.nx_alloc))
    kinds = [e[0] for e in sched]
    assert [e[5] for e in sched if e[0] == "group_part"] == [1, 3, 3, 1] and kinds.index("fft") < len(kinds) - 1 - kinds[::-1].index("group_part")
    pipe.packed.zero_()                         # (the slot of the rank's own part in the send buffer is never written)
    pipe.step(check=True)                       # raises if a deposit left the buffer or a particle arrived late
    assert int(pipe.staged.dropped.item()) == 0
    staged = pipe.buf.clone()
    ref = torch.empty_like(staged)
    pipe.ops.paint(pipe.pos, None, n, L, "cic", ref, pipe.x_start, pipe.nx_alloc, check=True, offset=pipe.mean_offset,
                   owned=(pipe.gl, pipe.nloc))
    assert torch.equal(staged, ref)
    # default: only the tile rows that hold ghost planes (0 and 16) are folded by the paint, the others by the z pass as it
    # loads their planes - same additions in the same order: the rank's own piece of the spectrum and its send buffer are
    # the same bits, the buffer's interior rows stay unfolded
    mine_ref, packed_ref = pipe.block.clone(), pipe.packed.clone()
    monkeypatch.delenv("ASTRILD_SLAB_DEFER_FOLD")
    pipe2 = slab.SlabPowerPipeline(n, L, n, window="cic", dtype=torch.float32, ghost=3, seed=20240601)
    pipe2.packed.zero_()
    pipe2.step(check=True)
    assert pipe2._defer_fold and int(pipe2.staged.dropped.item()) == 0
    assert torch.equal(pipe2.buf[:8], ref[:8]) and torch.equal(pipe2.buf[128:], ref[128:]) and not torch.equal(pipe2.buf[8:128], ref[8:128])
    lo, hi = rank * pipe.nloc, (rank + 1) * pipe.nloc
    assert torch.equal(pipe2.block[lo:hi], mine_ref[lo:hi]) and torch.equal(pipe2.packed, packed_ref)


def test_c_abi_comm_entries_over_rccl_single_rank(hip):
    """SURVEY.md S8(b)'s communication entries (ast_comm_init, ast_slab_transpose, ast_comm_allreduce_sum) for a caller that
    binds the library alone: RCCL loaded lazily, a one-rank communicator on cuda:0 - the rank's own piece is copied, the sum
    over one rank is the identity.  (More than one rank needs more than one GPU; the Python pipelines go through
    torch.distributed.)"""
    import ctypes as ct
    from astrild_amd import _lib, device as dev
    torch.cuda.set_device(0)
    uid = (ct.c_char * 128)()
    _lib.check(hip.ast_comm_unique_id(uid, 128), "ast_comm_unique_id")
    comm = ct.c_void_p()
    _lib.check(hip.ast_comm_init(ct.byref(comm), 1, 0, uid, 128), "ast_comm_init")
    try:
        send = torch.arange(1000, dtype=torch.float32, device="cuda")
        recv = torch.zeros(1500, dtype=torch.float32, device="cuda")
        sz = ct.c_size_t * 1
        _lib.check(hip.ast_slab_transpose(comm, dev.ptr(send), sz(100), sz(600), dev.ptr(recv), sz(300), sz(600), 0, dev.stream()),
                   "ast_slab_transpose")
        torch.cuda.synchronize()
        assert torch.equal(recv[300:900], send[100:700]) and float(recv[:300].abs().sum()) == 0.0 and float(recv[900:].abs().sum()) == 0.0
        sums = torch.tensor([1.5, -2.0, 3.25], dtype=torch.float64, device="cuda")
        _lib.check(hip.ast_comm_allreduce_sum(comm, dev.ptr(sums), 3, dev.stream()), "ast_comm_allreduce_sum")
        torch.cuda.synchronize()
        assert sums.tolist() == [1.5, -2.0, 3.25]
        assert hip.ast_slab_transpose(comm, dev.ptr(send), sz(0), sz(10), dev.ptr(recv), sz(0), sz(20), 0, dev.stream()) != 0   # own piece: sizes differ
    finally:
        _lib.check(hip.ast_comm_destroy(comm), "ast_comm_destroy")
