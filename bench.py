#!/usr/bin/env python3
"""bench.py — particles/s through the hot path: mass assignment (CIC) + 3D R2C FFT
+ FFTPower shell binning, inputs resident in HBM (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]        # N=1: one process
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = zero the grid, paint all particles, forward FFT, shell-bin P(k).
N=1 workload: 1024^3 synthetic lattice+Gaussian particles on a
1024^3 grid, fp32.  N>1: the same 1024^3 problem slab-decomposed along axis 0
(strong scaling; one RCCL all-to-all per step for the slab transpose).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
LEG_STEPS, LEG_WARMUP = 10, 2   # the secondary legs (shuffled / TSC / float64): as many timed steps as the headline (0.1-0.3 s each)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ngrid", type=int, default=1024)
    ap.add_argument("--npside", type=int, default=0, help="particle lattice side (default = ngrid)")
    ap.add_argument("--window", default="cic", choices=["cic", "tsc"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--order", default="natural", choices=["natural", "shuffled"])
    ap.add_argument("--method", default="auto", choices=["auto", "tiled", "tiled2", "direct"])
    ap.add_argument("--cpu-sample", type=int, default=512, help="lattice side of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--kappa", type=int, default=1, help="also time the kappa-map pipeline (1/0)")
    ap.add_argument("--bispec", type=int, default=1, help="also time the 512^3 bispectrum (1/0)")
    ap.add_argument("--legs", type=int, default=1, help="also time the shuffled-order and TSC legs (1/0)")
    ap.add_argument("--slab", type=int, default=0, help="run the slab-decomposed pipeline even on one GPU (rehearsal)")
    ap.add_argument("--cpu-paint-child", type=int, default=0, help="(internal) child mode of the CPU baseline's parallel paint")
    ap.add_argument("--cpu-workers", type=int, default=16, help="processes of the CPU baseline's parallel paint")
    return ap.parse_args()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sample, window, boxsize, dev=None):
    """The oracle (numpy port of the reference's CPU path: pmesh-convention paint,
    rfftn, FFTPower binning; float64 like the reference) on a bounded sample of the
    workload: sample^3 particles on a sample^3 grid.  `value` is the single-thread figure
    (how astrild runs: one MPI rank, FFTW without threads, numpy); `threaded` repeats the
    FFT with scipy.fft on every host core and the paint as plane chunks of the
    same numpy arithmetic in forked worker processes (cpu_parallel_paint)."""
    from oracle import mesh as omesh, fftpower as offt
    import scipy.fft
    ncpu = os.cpu_count() or 1
    pos = omesh.lattice_particles(sample, sample, boxsize, seed=20240601)
    t0 = time.perf_counter()
    grid = omesh.paint(pos, None, sample, boxsize, window)
    t1 = time.perf_counter()
    ref = offt.fftpower_1d(grid, boxsize)
    t2 = time.perf_counter()
    check = None
    if dev is not None:
        # the spectrum the CPU leg has just computed is the checker for the device on the SAME sample: float64 (the
        # reference's dtype) and the fp32 pipeline that `value` times (north_star: 1e-6)
        d64 = dev.paint_power_1d(dev.as_device(pos), None, sample, boxsize, window)
        p32 = dev.as_device(pos.astype(np.float32))
        d32 = dev.paint_power_1d(p32, None, sample, boxsize, window)
        d64b = dev.paint_power_1d(p32.double(), None, sample, boxsize, window)
        check = {"modes_equal": bool(np.array_equal(d64["modes"], ref["modes"]) and np.array_equal(d32["modes"], ref["modes"])),
                 "cpu_check_max_rel": float(np.max(np.abs(d64["power"] / ref["power"].real - 1.0))),
                 "fp32_pipeline_max_rel": float(np.max(np.abs(d32["power"] / d64b["power"] - 1.0))),
                 "note": "cpu_check_max_rel: device float64 P(k) / the oracle's P(k) - 1 over all shells, same particles (the "
                         "spectrum this CPU leg computed anyway); fp32_pipeline_max_rel: the fp32 pipeline that `value` times "
                         "against that float64 device path on the same fp32-rounded positions (north_star: 1e-6)"}
        del p32
        torch.cuda.empty_cache()
    workers = min(16, ncpu)            # (the GPU box hands one GPU's share of the host to this job: 16 cores)
    par = cpu_parallel_paint(sample, window, boxsize, workers) if sample % 16 == 0 else {"error": "sample is not a multiple of the 16-plane chunk"}
    par_ok = "paint_s" in par and np.allclose(par.get("moments", [0.0, 0.0]), _grid_moments(grid), rtol=1e-10, atol=0.0)
    t2b = time.perf_counter()
    spec = scipy.fft.rfftn(grid, workers=ncpu) / grid.size
    t3 = time.perf_counter()
    p3d = (spec * np.conj(spec)).real * boxsize ** 3
    p3d[0, 0, 0] = 0.0
    offt.project_1d(p3d, sample, boxsize)
    t4 = time.perf_counter()
    n = pos.shape[0]
    return {
        "value": n / (t2 - t0), "unit": "particles/s", "cores": 1, "kind": "port",
        "sample": f"{sample}^3 particles on a {sample}^3 grid, float64, numpy bincount paint {t1 - t0:.2f}s + "
                  f"numpy rfftn/shell binning {t2 - t1:.2f}s (1024^3 is not run: the numpy port's temporaries "
                  f"need > 100 GB of host RAM)",
        "threaded": {"value": n / ((par["paint_s"] if par_ok else (t1 - t0)) + (t4 - t2b)), "cores": ncpu,
                     "fft_s": round(t3 - t2b, 3), "binning_s": round(t4 - t3, 3), "paint_s_single_thread": round(t1 - t0, 3),
                     "paint_s_parallel": round(par["paint_s"], 3) if par_ok else None, "paint_workers": workers if par_ok else 1,
                     "paint_parallel_error": None if par_ok else par.get("error", "grid moments differ"),
                     "note": f"paint: the same chunked numpy paint in {workers} forked processes of a child interpreter (even chunks, then "
                             f"odd ones, into one shared grid); scipy.fft.rfftn(workers={ncpu}); numpy shell binning, single thread"},
        "cpu_model": _cpu_model(), "logical_cores": ncpu, "check": check,
        # the same port on the benchmark's own configuration, by particle count (its paint, binning and FFT are all
        # O(N) or O(N log N) in the 8x larger problem): a projection, not a measurement
        "projected_1024_cubed": {"seconds_single_thread": round(1024 ** 3 / (n / (t2 - t0)), 1),
                                 "seconds_threaded": round(1024 ** 3 / (n / ((par["paint_s"] if par_ok else (t1 - t0)) + (t4 - t2b))), 1),
                                 "note": "particles of the 1024^3 workload / the sample's particles per second"},
    }


def _cpu_paint_chunk(pos, i0, planes, pad, n, boxsize, window, grid):
    """oracle.mesh.paint's arithmetic for particles whose base planes lie in [i0 - pad, i0 + planes + pad): bincounts over
    that slab of the grid only (a bincount over the whole 1024^3 grid per window corner would allocate 8.6 GB each)."""
    from oracle import mesh as omesh
    support = {"cic": 2, "tsc": 3}[window]
    nloc = planes + 2 * pad
    idx, wts = [], []
    for d in range(3):
        a, b = omesh.window_1d(pos[:, d] * (n / boxsize), window)
        idx.append(a)
        wts.append(b)
    rel = np.mod(idx[0] - (i0 - pad), n)
    assert int(rel.max()) + support <= nloc, "a particle left the chunk's slab"
    local = np.zeros(nloc * n * n)
    for a in range(support):
        for b in range(support):
            ib = np.mod(idx[1] + b, n)
            wab = wts[0][a] * wts[1][b]
            for c in range(support):
                flat = ((rel + a) * n + ib) * n + np.mod(idx[2] + c, n)
                local += np.bincount(flat, weights=wab * wts[2][c], minlength=nloc * n * n)
    rows = np.mod(np.arange(i0 - pad, i0 + planes + pad), n)
    grid[rows] += local.reshape(nloc, n, n)


def cpu_baseline_chunked(sample, window, boxsize, planes=32, pad=6, pos=None, dev=None):
    """The same CPU port at sizes whose temporaries do not fit the host in one piece ("one 1024^3 run if host
    RAM >= 64 GB"): the particles are generated and painted `planes` lattice planes at a time (the oracle's window weights,
    ONE numpy bincount per chunk over the chunk's slab of the grid; generation is not timed), the transform is one
    scipy.fft.rfftn (single thread, then all cores), the shells are binned x-slab by x-slab with the oracle's
    project_block.  Same arithmetic as cpu_baseline's port; float64 like the reference.  Opt-in: --cpu-sample >= 768."""
    from oracle import mesh as omesh, fftpower as offt
    import scipy.fft
    n = int(sample)
    ncpu = os.cpu_count() or 1
    h = boxsize / n
    g = (np.arange(n) + 0.5) * h
    rng = np.random.Generator(np.random.PCG64(20240601))
    grid = np.zeros((n, n, n))
    t_paint = 0.0
    given = pos                       # (n^3, 3) host array in lattice order (the device's own synthetic set), or None
    for i0 in range(0, n, planes):
        if given is not None:
            chunk = np.asarray(given[i0 * n * n:(i0 + planes) * n * n], dtype=np.float64)
        else:
            q = np.stack(np.meshgrid(g[i0:i0 + planes], g, g, indexing="ij"), axis=-1).reshape(-1, 3)
            chunk = np.mod(q + 0.5 * h * rng.standard_normal(q.shape), boxsize)
            del q
        t0 = time.perf_counter()
        _cpu_paint_chunk(chunk, i0, planes, pad, n, boxsize, window, grid)
        t_paint += time.perf_counter() - t0
        del chunk
    t0 = time.perf_counter()
    spec = scipy.fft.rfftn(grid, workers=1)
    t_fft1 = time.perf_counter() - t0
    del spec
    t0 = time.perf_counter()
    spec = scipy.fft.rfftn(grid, workers=ncpu)
    t_fftn = time.perf_counter() - t0
    del grid
    t0 = time.perf_counter()
    nb = n // 2 - 1
    ks, ps, nm = np.zeros(nb), np.zeros(nb), np.zeros(nb, dtype=np.int64)
    ng = float(n) ** 3
    for x0 in range(0, n, 8):
        blk = spec[x0:x0 + 8]
        p3d = (blk.real ** 2 + blk.imag ** 2) * (boxsize ** 3 / (ng * ng))
        if x0 == 0:
            p3d[0, 0, 0] = 0.0
        a, b, c = offt.project_block(p3d, n, boxsize, x0, 0)
        ks += a
        ps += np.real(b)
        nm += c
    t_bin = time.perf_counter() - t0
    npart = n ** 3
    assert np.isfinite(ps / nm).all()
    single = t_paint + t_fft1 + t_bin
    check = None
    if given is not None and dev is not None:
        # the CPU port's spectrum of the device's OWN particle set is the checker at the benchmark's size: the float64 device
        # pipeline and the fp32 pipeline that `value` times, on the same (fp32-rounded) positions
        p32 = dev.as_device(np.ascontiguousarray(given, dtype=np.float32))
        d32 = dev.paint_power_1d(p32, None, n, boxsize, window)
        d64 = dev.paint_power_1d(p32.double(), None, n, boxsize, window)
        del p32
        torch.cuda.empty_cache()
        cpu_p = ps / nm
        check = {"modes_equal": bool(np.array_equal(d64["modes"], nm) and np.array_equal(d32["modes"], nm)),
                 "float64_pipeline_vs_cpu_max_rel": float(np.max(np.abs(d64["power"] / cpu_p - 1.0))),
                 "fp32_pipeline_vs_cpu_max_rel": float(np.max(np.abs(d32["power"] / cpu_p - 1.0))),
                 "note": "same particles (the device's synthetic set, fp32-rounded, copied to the host); north_star: 1e-6"}
    return {"value": npart / single, "unit": "particles/s", "cores": 1, "kind": "port", "check": check,
            "sample": f"{n}^3 particles on a {n}^3 grid, float64: chunked numpy bincount paint {t_paint:.1f}s + scipy rfftn (1 thread) "
                      f"{t_fft1:.1f}s + shell binning {t_bin:.1f}s (the benchmark's own configuration; particles generated and "
                      f"painted {planes} lattice planes at a time)",
            "threaded": {"value": npart / (t_paint + t_fftn + t_bin), "cores": ncpu, "fft_s": round(t_fftn, 2),
                         "note": f"scipy.fft.rfftn(workers={ncpu}); paint and binning as in the single-thread leg"},
            "seconds_single_thread": round(single, 1), "paint_s": round(t_paint, 1), "fft_s_single_thread": round(t_fft1, 1),
            "binning_s": round(t_bin, 1), "cpu_model": _cpu_model(), "logical_cores": ncpu}


def cpu_parallel_paint_child(sample, window, boxsize, workers, planes=16, pad=6):
    """Runs in a CHILD process of bench.py (`--cpu-paint-child`), which never touches the GPU: the chunked CPU paint of the
    lattice set with `workers` forked processes writing into one shared grid - the chunks of a phase (even ones, then odd
    ones) touch disjoint slabs of it (chunk + 2 pad planes < two chunks).  Prints one JSON line: paint seconds, grid sum."""
    import multiprocessing as mp
    from oracle import mesh as omesh
    n = int(sample)
    pos = omesh.lattice_particles(n, n, boxsize, seed=20240601)
    shared = mp.RawArray("d", n * n * n)
    grid = np.frombuffer(shared, dtype=np.float64).reshape(n, n, n)
    per = n * n * planes

    def work(c):
        g = np.frombuffer(shared, dtype=np.float64).reshape(n, n, n)
        _cpu_paint_chunk(pos[c * per:(c + 1) * per], c * planes, planes, pad, n, boxsize, window, g)
        return c
    chunks = list(range(n // planes))
    ctx = mp.get_context("fork")
    global _CPU_CHILD_WORK
    _CPU_CHILD_WORK = work
    t0 = time.perf_counter()
    with ctx.Pool(processes=workers) as pool:
        tail = [chunks.pop()] if len(chunks) % 2 and len(chunks) > 1 else []   # an odd count: the last chunk wraps onto chunk 0
        pool.map(_cpu_child_call, chunks[0::2])
        pool.map(_cpu_child_call, chunks[1::2])
        pool.map(_cpu_child_call, tail)
    dt = time.perf_counter() - t0
    print(json.dumps({"paint_s": dt, "workers": workers, "grid_sum": float(grid.sum()), "moments": _grid_moments(grid),
                      "nparticles": int(pos.shape[0])}))


def _grid_moments(grid):
    """First moments of a grid along x and z: the fingerprint by which the parent compares the child's paint with its own."""
    ramp = np.arange(1, grid.shape[0] + 1, dtype=np.float64)
    return [float(grid.sum(axis=(1, 2)) @ ramp), float(grid.sum(axis=(0, 1)) @ ramp)]


_CPU_CHILD_WORK = None


def _cpu_child_call(c):
    return _CPU_CHILD_WORK(c)


def cpu_parallel_paint(sample, window, boxsize, workers):
    """The threaded leg's paint: a child process (fresh interpreter, no GPU) paints the same sample with `workers` processes."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-paint-child", str(int(sample)), "--window", window,
           "--cpu-workers", str(int(workers))]
    try:
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"))
        line = [ln for ln in res.stdout.decode().splitlines() if ln.startswith("{")][-1]
        return json.loads(line)
    except Exception as exc:          # the baseline is a reported figure: a failure here must not take the bench line down
        return {"error": repr(exc)}


def stream_ceiling(dev, gib=4, reps=5):
    """The on-box streaming rates: `ast_stream_copy` over `gib` GiB (far beyond the 256 MB Infinity Cache),
    HIP events on the stream it is launched on, best of `reps`: copy (read + write bytes / time), read only, write only."""
    from astrild_amd._lib import lib, check
    nbytes = int(gib) << 30
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev.device()).fill_(1.0)
    b = torch.empty_like(a)
    out = {}
    for name, mode, moved in (("copy", 0, 2 * nbytes), ("read", 1, nbytes), ("write", 2, nbytes)):
        best = float("inf")
        for _ in range(reps + 1):                      # (the first one warms up)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib().ast_stream_copy(dev.ptr(b), dev.ptr(a), nbytes, mode, dev.stream()), "ast_stream_copy")
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        out[name + "_GBps"] = round(moved / (best * 1e6), 1)
    assert bool(torch.equal(a[:1 << 20], torch.ones(1 << 20, device=a.device)))
    del a, b
    torch.cuda.empty_cache()
    out["buffer_GiB"] = gib
    out["note"] = ("hand-written 16-byte-per-lane streaming kernel of this library on this GPU (the fastest of its 16 access variants "
                   "per operation, scripts/micro/copy_rate.py), best of %d; the guide's figure for a float4 copy is 6290 GB/s "
                   "(MI355X_MICROARCH.md)" % reps)
    return out


def subfind_leg(dev, nobj=2_000_000, nbins=512, boxsize=500.0, reps=5):
    """SubFind.power_spectrum's real shape (stats_subfind.py:109-153): 2e6 mass-weighted objects, TSC, nbins = 512,
    float64, through the Python API from host arrays - H2D copies, mass bound, paint, /dx^3, FFTPower.  The catalogue:
    positions drawn around 4096 centres (Gaussian clumps of 0.5-4 Mpc/h), masses log-uniform over three decades."""
    import types
    from astrild_amd.particles.hutils.stats_subfind import SubFind
    rng = np.random.default_rng(20240601)
    h = 0.6774
    centres = rng.uniform(0.0, boxsize, size=(4096, 3))
    which = rng.integers(0, 4096, size=nobj)
    radius = rng.uniform(0.5, 4.0, size=4096)[which]
    pos = np.mod(centres[which] + rng.standard_normal((nobj, 3)) * radius[:, None], boxsize)
    mass = 10.0 ** rng.uniform(0.0, 3.0, size=nobj)
    snap = types.SimpleNamespace(cat={"SubhaloPos": pos * 1e3 / h, "SubhaloMass": mass * 1e10 / h},
                                 header=types.SimpleNamespace(hubble=h, boxsize=boxsize * 1e3))
    k, pk = SubFind.power_spectrum(snap, nbins=nbins, boxsize=boxsize)           # warm-up: plans, scratch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        k, pk = SubFind.power_spectrum(snap, nbins=nbins, boxsize=boxsize)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert np.isfinite(pk).all()
    dpos = dev.as_device(pos)
    dmass = dev.as_device(mass)
    st = {}
    dev.paint(dpos, dmass, nbins, boxsize, "tsc", stats=st)
    dense = dev.auto_paint_method(nobj, nbins, nbins, "tsc") != "direct"
    return {"metric": f"SubFind.power_spectrum: {nobj} mass-weighted objects, TSC, nbins {nbins}, float64, host arrays in, (k, Pk) out",
            "ms_per_call": round(dt * 1e3, 3), "objects_per_s": nobj / dt,
            "h2d_MB": round((pos.nbytes + mass.nbytes) / 1e6, 1),
            "paint_path": st.get("path") or ("direct global atomics (sparse catalogue: %.1f objects per 8x8x32-cell tile)" % (nobj * 2048 / nbins ** 3)
                                             if not dense else "tiled"),
            "paint_attempts": st.get("attempts", 1),
            "note": "the whole API call from numpy arrays in catalogue units: two H2D copies, paint (unit factors folded into the cell "
                    "lookup and the mass scale; the reference's host-side conversion alone takes 7.5 ms here), /dx^3, fused float64 FFT + "
                    "FFTPower shells, D2H of (k, Pk)"}


def bispectrum_leg(dev, n=512, width=8):
    """Config E: matter bispectrum by FFT triangle counting on a 512^3 grid (equilateral +
    one squeezed and one isosceles family over shells of width 8 k_F), fp32, one GPU."""
    L = 1000.0
    pos = dev.synth_lattice_particles(n, n, L, seed=20240601, dtype=torch.float32)
    grid = dev.paint(pos, None, n, L, "cic")
    del pos
    edges = list(range(1, n // 2 + 1, width))
    nsh = len(edges) - 1
    tri = [(i, i, i) for i in range(nsh)] + [(0, i, i) for i in range(1, nsh)] + \
          [(i, i, min(nsh - 1, 2 * i)) for i in range(1, nsh // 2)]
    dev._tri_cache.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev.bispectrum(grid, L, edges, tri)                      # first call: rocFFT plans + the triangle counts (fp64 I-fields),
    torch.cuda.synchronize()                                 # which depend on (N, shells, triangles) only and are cached
    first_ms = (time.perf_counter() - t0) * 1e3
    dev.profile_enable(True)
    t0 = time.perf_counter()
    res = dev.bispectrum(grid, L, edges, tri)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = dev.profile_report()
    dev.profile_enable(False)
    ng = n ** 3
    traffic = None
    import glob
    tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_bispectrum.json")))
    if tf:                                    # PMC passes kept under profiles/ (scripts/refresh_profiles.sh): NOT measured by this run
        import hashlib
        doc = json.load(open(tf[-1]))
        now = {f: hashlib.sha256(open(os.path.join(ROOT, "astrild_amd", "csrc", f), "rb").read()).hexdigest()[:16]
               for f in ("fft_tile.hip", "power_bin.hip")}
        want = {"n": n, "shell_width": width, "shells": nsh, "triangle_bins": len(tri)}
        # quoted only if the file was made for THIS grid, these shells and triangles and the kernel sources as they are now
        same = doc.get("config") == want and doc.get("src_sha256_16") == now
        traffic = {"GB_per_call": doc.get("numerator_corrected_GB_per_call") if same else None,
                   "file": os.path.relpath(tf[-1], ROOT), "measured_in_this_run": False, "matches_this_run": same,
                   "profiled": {"config": doc.get("config"), "src_sha256_16": doc.get("src_sha256_16")},
                   "current": {"config": want, "src_sha256_16": now}}
    # Algorithmic bytes of the numerator: the forward transform of the grid (24 B per cell), per shell the three PRUNED
    # inverse passes - what a shell of outer radius m leaves nonzero, at the passes' 16-column tile granularity: the x pass
    # reads the rows |k_x| < m of the (k_y, k_z tile) pairs inside the disc and writes those pairs' columns, the y pass reads
    # them and writes the k_z tiles below m for every (x, k_y), the z pass reads k_z < m and writes the real cube - and one
    # read of the 31 cubes by the triangle sums.  (scripts/pmc_per_launch.py prints the same model beside the counters.)
    nz = n // 2 + 1
    # (round 5: the z passes of all shells and the triangle sums are ONE kernel - the real cubes are neither written nor read)
    fused_tail = nsh <= 32 and os.environ.get("ASTRILD_BISPEC_FUSED", "1") != "0"
    alg = 24 * ng + (0 if fused_tail else nsh * 4 * ng)
    ky2 = np.minimum(np.arange(n), n - np.arange(n)).astype(np.int64) ** 2
    for sh in range(nsh):
        m = edges[sh + 1]
        cols16 = sum(int(np.count_nonzero(ky2 + c0 * c0 < m * m)) * min(16, nz - c0) for c0 in range(0, nz, 16))
        kz_t = sum(min(16, nz - c0) for c0 in range(0, nz, 16) if c0 < m)
        alg += 8 * (cols16 * min(n, 2 * m) + 2 * cols16 * n + n * n * kz_t + n * n * min(m, nz)) + (0 if fused_tail else 4 * ng)
    unpruned = nsh * 24 * ng + nsh * 4 * ng
    return {"metric": f"bispectrum on {n}^3 grid: {nsh} shells of width {width} k_F, {len(tri)} triangle bins, fp32",
            "value": len(tri) / dt, "unit": "triangle bins/s", "ms_total": dt * 1e3,
            "alg_GB": round(alg / 1e9, 2), "GBps": round(alg / dt / 1e9, 1), "frac": round(alg / dt / 1e9 / HBM_PEAK_GBS, 4),
            "unpruned_GB": round(unpruned / 1e9, 2),
            "traffic": traffic,
            "frac_of_peak_on_bytes_moved": (round(traffic["GB_per_call"] / dt / HBM_PEAK_GBS, 4)
                                            if traffic and traffic["GB_per_call"] else None),
            "ntri_total": int(np.sum(res["ntri"])), "ntri_residual": res["ntri_residual"],
            "first_call_ms_with_triangle_counts": round(first_ms, 1),
            "fused_tail": fused_tail,
            "note": "value / ms_total time the estimator's numerator (31 masked, pruned x / y inverse passes, then ONE kernel that runs "
                    "the z passes of all shells row by row and forms the 75 triangle sums from LDS - no real cube reaches HBM; "
                    "ASTRILD_BISPEC_FUSED=0: 31 cubes written and read back by the triangle kernel); alg_GB / frac price the PRUNED passes (what a shell leaves nonzero; unpruned_GB: three "
                    "full passes per shell), frac_of_peak_on_bytes_moved uses the bytes the PMC counters saw; the triangle counts (31 "
                    "forward float64 transforms of the shell indicators + the sums) are geometry, computed on the first call "
                    "and cached - first_call_ms includes them",
            "kernels_ms": {k: round(v[1], 3) for k, v in prof.items()}}


def kappa_leg(dev, steps, warmup, group=None):
    """64 planes x 4096^2 fp64 -> stack -> Gaussian FFT smoothing -> kappa->alpha (config D);
    with a process group the planes are sharded over its ranks."""
    from astrild_amd import lensing
    return lensing.bench_kappa_pipeline(nplanes=64, npix=4096, steps=max(3, min(steps, 10)), warmup=min(warmup, 2),
                                        group=group)


def kappa_cpu_baseline(dev, nplanes=64, npix=4096, theta_deg=20.0, sigma_arcmin=1.0):
    """The oracle (numpy port of rayramses.py:186-232 + sky_utils.py:318-339 + filters.py:181-225 + lensing_funcs.c:45-115 +
    sky_array.py:428-433, float64, one thread) on ONE map of the kappa leg's own workload - the same 64 synthetic planes,
    fetched from the device -, timed on the host; and its map and deflection field as the checker of the device's."""
    from oracle import kappa as ok
    from astrild_amd import lensing
    planes_d = lensing.synth_kappa_planes(nplanes, npix)
    wnum, wden = lensing.synth_plane_weights(nplanes)
    bsz = np.deg2rad(theta_deg)
    sigma_px = sigma_arcmin / 60.0 * npix / theta_deg
    # the device's map first (its own code path, as in the timed leg)
    out = lensing.kappa_stack(planes_d, wnum, wden)
    lensing.convert_code_to_phy_units("kappa_2", out)
    lensing.smooth_plan(npix).gaussian(out, sigma_px, "gaussianFFT")
    a1_d, _ = lensing.lens_plan(npix, bsz).alphas(out)
    kappa_d, a1_d = out.cpu().numpy(), a1_d.cpu().numpy()
    planes = [p.cpu().numpy() for p in planes_d]
    del planes_d, out
    torch.cuda.empty_cache()
    mid = (np.arange(nplanes) + 0.5) * (1000.0 / nplanes)
    half = 0.5 * 1000.0 / nplanes
    t0 = time.perf_counter()
    total = ok.kappa_stack(planes, mid - half, mid + half, 1100.0, 1000.0)
    t1 = time.perf_counter()
    img = ok.gaussian_smooth(ok.convert_code_to_phy_units("kappa_2", total), theta_deg, sigma_arcmin, "gaussianFFT")
    t2 = time.perf_counter()
    a1, a2 = ok.kappa0_to_alphas(img, npix, bsz)
    t3 = time.perf_counter()
    ok.pdf(img, 100)
    t4 = time.perf_counter()
    scale_k, scale_a = float(np.abs(img).max()), float(np.abs(a1).max())
    return {"value": 1.0 / (t4 - t0), "unit": "maps/s", "cores": 1, "kind": "port",
            "sample": f"one map of the leg's own workload ({nplanes} planes x {npix}^2 float64): stack {t1 - t0:.2f}s + units and "
                      f"Gaussian FFT smoothing {t2 - t1:.2f}s + kappa->alpha on the padded {2 * npix}^2 array {t3 - t2:.2f}s + PDF {t4 - t3:.2f}s",
            "check": {"kappa_max_abs_over_max": float(np.abs(kappa_d - img).max() / scale_k),
                      "alpha1_max_abs_over_max": float(np.abs(a1_d - a1).max() / scale_a),
                      "note": "the device's smoothed map and alpha1 against the CPU leg's own, same planes"}}


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(ngpus):
    """`python bench.py --gpus N` without a launcher: start the N rank processes ourselves (one per GPU, through
    torch.distributed.run) as CHILDREN of this process, which has not touched the GPU and never will; relay rank 0's
    JSON line and exit with the launcher's status."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for ln in proc.stdout.decode(errors="replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    if proc.returncode != 0 or line is None:
        raise SystemExit(proc.returncode or 1)


def main():
    args = parse()
    if args.cpu_paint_child:
        cpu_parallel_paint_child(args.cpu_paint_child, args.window, 1000.0, args.cpu_workers)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)
        return
    # stdout carries exactly ONE JSON line: everything else a library may print there
    # (RCCL's version banner at communicator creation, for one) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    # rehearsal of the N > 1 code path on a one-GPU box (timings are meaningless there):
    # ASTRILD_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo instead of RCCL
    rehearsal = os.environ.get("ASTRILD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    use_slab = world > 1 or bool(args.slab)
    if use_slab:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            import datetime
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=int(os.environ.get("ASTRILD_SLAB_TIMEOUT_S", "240"))))

    from astrild_amd import device as dev
    n = args.ngrid
    npside = args.npside or n
    L = 1000.0
    wd = None
    if use_slab:
        # every wait of a multi-rank run is bounded: the watchdog ends THIS process (a fresh non-zero exit, never a re-exec)
        # with the rank, its host stage, the last schedule entry its GPU completed and the host threads' stacks
        from astrild_amd import slab
        wd = slab.Watchdog(rank=rank)
        wd.beat("pipeline construction (particles, geometry all-reduces)")

    def barrier():
        if use_slab:
            dist.barrier()
        torch.cuda.synchronize()

    if not use_slab:
        leg = power_leg(dev, n, npside, L, args.window, args.order, args.dtype, args.method, args.steps, args.warmup)
    else:
        leg = slab_leg(dev, dist, n, npside, L, args, world, barrier, wd)
    ms_per_step = leg["ms_per_step"]
    npart_total = npside ** 3
    value = npart_total / (ms_per_step * 1e-3)
    roofline = leg["roofline"]

    out = {
        "metric": "particles/sec CIC+3D-FFT P(k) on 1024^3 grid" if n == 1024 and args.window == "cic"
                  else f"particles/sec {args.window.upper()}+3D-FFT P(k) on {n}^3 grid",
        "value": value, "unit": "particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{npside}^3 lattice+Gaussian(0.5 cell) particles ({args.order} order) -> "
                               f"{args.window.upper()} paint on {n}^3 grid -> 3D R2C -> FFTPower 1d shells",
                   "ngrid": n, "nparticles": npart_total, "boxsize": L,
                   "order_note": "natural = lattice order, the spatially coherent best case for the scatter; the "
                                 "shuffled (worst case), TSC and float64 figures are in `legs`",
                   "outside_timed_step": "FFT twiddles, per-shell geometry sums (sum w|k|, mode counts: data independent, "
                                         "cached per (N, L) like an FFT plan), workspace allocation, the input probe that "
                                         "picks the paint path (once per particle array: config.paint_path.probe_ms)",
                   "parallelism": "single GPU" if not use_slab else f"axis-0 slabs x{world}, RCCL all-to-all transpose"},
        "roofline": roofline,
    }
    if rank == 0 and not use_slab:
        # every fraction above is of the data sheet's 8 TB/s; the same figures against what this GPU streams (measured here)
        ceil = stream_ceiling(dev)
        ceil["reading_the_fractions"] = ("frac_of_copy_rate = algorithmic GB/s / copy_GBps; a stage that moves fewer bytes than the "
                                         "algorithmic count - the FFT passes pruned to the Nyquist disc - can read above 1")
        roofline["streaming_ceiling"] = ceil
        roofline["frac_of_copy_rate"] = round(roofline["achieved"] / ceil["copy_GBps"], 4)
        roofline["end_to_end"]["frac_of_copy_rate"] = round(roofline["end_to_end"]["GBps"] / ceil["copy_GBps"], 4)
        for st in roofline["stages"].values():
            st["frac_of_copy_rate"] = round(st["GBps"] / ceil["copy_GBps"], 4)
    if "diag" in leg:
        out["multi_gpu"] = leg["diag"]
    if "paint_path" in leg:
        out["config"]["paint_path"] = leg["paint_path"]
    if rank == 0 and not use_slab:
        if args.legs:
            # the other orderings / windows of the same workload, a few steps each (same kernels, same accounting)
            legs = {}
            for name, (win, order) in {"shuffled_cic": ("cic", "shuffled"), "natural_tsc": ("tsc", "natural"),
                                       "shuffled_tsc": ("tsc", "shuffled"),      # what stats_subfind.py:125-131 feeds the paint
                                       "natural_cic": ("cic", "natural")}.items():
                if (win, order) == (args.window, args.order):
                    continue
                lg = power_leg(dev, n, npside, L, win, order, args.dtype, args.method, steps=LEG_STEPS, warmup=LEG_WARMUP)
                legs[name] = {"ms_per_step": round(lg["ms_per_step"], 3), "particles_per_s": npart_total / (lg["ms_per_step"] * 1e-3),
                              "end_to_end_frac": lg["roofline"]["end_to_end"]["frac"], "paint_path": lg["paint_path"],
                              "stages": {k: {"ms": v["ms"], "frac": v["frac"]} for k, v in lg["roofline"]["stages"].items()}}
            # CLUSTERED input - what the reference's paint calls really see (evolved snapshots; stats_subfind.py:125-131): the
            # lattice collapsing onto 256 attractors, tile occupancies ~100 x the mean, in file (lattice) order and in
            # pseudo-random order; 512^3 particles on a 512^3 grid
            nc = min(n, 512)
            for name, order in (("clustered_natural_cic", "natural"), ("clustered_shuffled_cic", "shuffled")):
                lg = power_leg(dev, nc, nc, L, "cic", order, args.dtype, args.method, steps=LEG_STEPS, warmup=LEG_WARMUP, clustered=True)
                legs[name] = {"ngrid": nc, "nparticles": nc ** 3, "ms_per_step": round(lg["ms_per_step"], 3),
                              "particles_per_s": nc ** 3 / (lg["ms_per_step"] * 1e-3), "paint_path": lg["paint_path"],
                              "end_to_end_frac": lg["roofline"]["end_to_end"]["frac"],
                              "stages": {k: {"ms": v["ms"], "frac": v["frac"], "kernels": v["kernels"]} for k, v in lg["roofline"]["stages"].items()}}
            legs["subfind_power_spectrum"] = subfind_leg(dev)
            if args.dtype == "f32":
                # the reference's own dtype: float64 particles and grid through the double-precision passes
                lg = power_leg(dev, n, npside, L, "cic", "natural", "f64", args.method, steps=LEG_STEPS, warmup=LEG_WARMUP)
                legs["natural_cic_f64"] = {"ms_per_step": round(lg["ms_per_step"], 3),
                                           "particles_per_s": npart_total / (lg["ms_per_step"] * 1e-3),
                                           "end_to_end_frac": lg["roofline"]["end_to_end"]["frac"], "dtype": "f64",
                                           "stages": {k: {"ms": v["ms"], "frac": v["frac"]} for k, v in lg["roofline"]["stages"].items()}}
            out["legs"] = legs
        if args.cpu_sample >= 768:        # the benchmark's own size, chunked (minutes of CPU time: opt-in), beside the bounded sample
            out["cpu_baseline"] = cpu_baseline(512, args.window, L, dev)
            host_pos = dev.synth_lattice_particles(args.cpu_sample, args.cpu_sample, L, seed=20240601, dtype=torch.float32).cpu().numpy()
            torch.cuda.empty_cache()
            out["cpu_baseline"]["at_benchmark_size"] = cpu_baseline_chunked(args.cpu_sample, args.window, L, pos=host_pos, dev=dev)
            del host_pos
        elif args.cpu_sample:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.window, L, dev)
        torch.cuda.empty_cache()
        if args.bispec:
            out["bispectrum"] = bispectrum_leg(dev)
            dev._tri_cache.clear()
            torch.cuda.empty_cache()
        if args.kappa:
            out["kappa"] = kappa_leg(dev, args.steps, args.warmup)
            torch.cuda.empty_cache()
            from astrild_amd import lensing
            out["kappa"]["api"] = lensing.bench_kappa_api()
            out["kappa"]["api_per_map_chain"] = lensing.bench_skyarray_chain()
            if args.cpu_sample:
                out["kappa"]["cpu_baseline"] = kappa_cpu_baseline(dev)
            # the stack reads 64 planes for one map written: against the measured READ rate; the whole pipeline against the copy rate
            out["kappa"]["stack"]["frac_of_read_rate"] = round(out["kappa"]["stack"]["GBps"] / ceil["read_GBps"], 4)
            out["kappa"]["roofline"]["frac_of_copy_rate"] = round(out["kappa"]["roofline"]["achieved"] / ceil["copy_GBps"], 4)
    if use_slab and world > 1 and args.kappa:
        # config D on N GPUs: lens planes sharded over the ranks (every rank takes part)
        del leg
        torch.cuda.empty_cache()
        if wd is not None:
            wd.watch(None)
            wd.beat("kappa leg (planes sharded over the ranks)")
        kl = kappa_leg(dev, args.steps, args.warmup, group=dist.group.WORLD)
        if wd is not None:
            wd.beat("kappa leg done")
        if rank == 0:
            out["kappa"] = kl
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_slab:
        dist.destroy_process_group()
    if wd is not None:
        wd.stop()


def _stage_table(prof, steps, npart_rank, ng_rank, esz, fused_bin, concurrent=()):
    """Roofline per logical stage: SURVEY.md §8(d) algorithmic bytes / HIP-event time.  `concurrent`: sites that run
    on a second stream beside the stage's other kernels - listed, but their time is not added to the stage's."""
    stage_sites = {
        "paint": [k for k in prof if k.startswith("paint")],
        "fft": [k for k in prof if k.startswith("rocfft") or k.startswith("fft_tile") or k.startswith("fft64") or k.startswith("fft32big") or k.startswith("slab.")],
        "power_bin": [k for k in prof if k == "power_bin"],        # absent when fused into the last FFT pass
    }
    stage_bytes = {
        "paint": npart_rank * 3 * esz + ng_rank * esz,       # read positions once, write the grid once
        "fft": 3 * 2 * ng_rank * esz,                         # 3 axis passes x (read + write)
        "power_bin": ng_rank * esz,                           # half spectrum read once (~esz B per real cell)
    }
    if fused_bin and not stage_sites["power_bin"]:
        stage_bytes["fft"] += stage_bytes.pop("power_bin")    # fused: one stage carries both terms
        stage_sites.pop("power_bin")
    stages = {}
    for name, sites in stage_sites.items():
        ms = sum(prof[s][1] for s in sites if s not in concurrent) / steps
        if ms > 0:
            gbs = stage_bytes[name] / ms / 1e6
            stages[name] = {"ms": round(ms, 4), "alg_GB": round(stage_bytes[name] / 1e9, 3),
                            "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                            "kernels": {s: round(prof[s][1] / steps, 4) for s in sites}}
            side = [s for s in sites if s in concurrent]
            if side:
                stages[name]["concurrent"] = {"sites": side, "note": "not added to `ms`: fft_tile.lowk = the low-k channel's "
                                              "y / x / shell kernels on a second stream beside the y pass (its z sums are "
                                              "formed inside rows_r2c at N = 1024); paint_tiled.fill / .deposit = the "
                                              "grouping and column-walk launches of the x-sorted pipeline, which overlap on "
                                              "two streams - paint_tiled.pipeline is their wall time on the launch stream"}
    return stages, stage_sites, stage_bytes


def _traffic_from_profiles():
    """HBM traffic of the paint stage from the PMC passes kept under profiles/ (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, gfx950 x2 read correction).  It is NOT measured by this run: the newest
    profiles/r*_pmc_traffic.json is quoted together with where it came from, and only if the paint kernels'
    source is byte-identical to the one profiled (else null)."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    doc = json.load(open(files[-1]))
    src = os.path.join(ROOT, "astrild_amd", "csrc", "mesh_paint_tiled.hip")
    sha = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16] if os.path.isfile(src) else None
    source = {"file": os.path.relpath(files[-1], ROOT), "paint_src_sha256_16": doc.get("paint_src_sha256_16"),
              "current_paint_src_sha256_16": sha, "measured_in_this_run": False}
    gb = doc.get("paint_stage_corrected_GB_per_step")
    if gb is None or doc.get("paint_src_sha256_16") != sha:
        return None, source
    return gb * 1e9, source


def power_leg(dev, n, npside, L, window, order, dtype, method, steps, warmup, clustered=False):
    """One configuration of the 3D path on ONE GPU: paint -> 3D R2C -> shell binning, `steps` timed steps.
    clustered: the lattice collapsing onto 256 attractors (tile occupancies ~100 x the mean) instead of the jittered lattice."""
    tdt = torch.float32 if dtype == "f32" else torch.float64
    esz = 4 if dtype == "f32" else 8
    npart_total = npside ** 3
    if clustered:
        pos = dev.synth_clustered_particles(npside, n, L, seed=20240601, shuffle=(order == "shuffled"), dtype=tdt)
    else:
        pos = dev.synth_lattice_particles(npside, n, L, seed=20240601, shuffle=(order == "shuffled"), dtype=tdt)
    grid = torch.empty((n, n, n), dtype=tdt, device="cuda")
    psum = torch.zeros(n // 2 - 1, dtype=torch.float64, device="cuda")
    dev.shell_geometry(n, L)          # data independent, cached like the FFT plan
    fused = dev.fused_power_supported(grid)
    fused64 = dev.fused_power64_supported(grid) and not os.environ.get("ASTRILD_BENCH_ROCFFT64")
    # fp32 grids of the sides the fp32 tile passes do not cover (128, 2048): painted as rho - mean; side 128 transformed in DOUBLE
    # straight from the fp32 grid (ast_fft64_power_3d_f32), side 2048 by three-stage single-precision passes
    # (ast_fft32_big_power_3d) with the lowest shells from the double-precision side channel
    f32_via_double = dtype == "f32" and not fused and dev.fused_power64_supported(grid, allow_f32=True)
    spec = None
    if not fused and not fused64 and not f32_via_double:
        spec = torch.empty((n, n, n // 2 + 1), dtype=torch.complex64 if dtype == "f32" else torch.complex128, device="cuda")
    mean = npart_total / float(n) ** 3
    # WHICH paint path: decided from the input by device.probe_input (order in memory + tile-occupancy tail of a sample:
    # two small kernels and one 24-byte fetch), once per particle array - like an FFT plan, outside the timed steps (the
    # fetch would otherwise serialise the host with the GPU every step); its cost is reported as `probe_ms`.  A caller of
    # dev.paint without a hint gets the same probe inside the call.
    dev.probe_input(pos, n, L)            # (first call: allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    probe = dev.probe_input(pos, n, L)
    probe_ms = (time.perf_counter() - t0) * 1e3
    if probe is not None and probe["groupable"] < 0.25:
        hint = "scattered"                 # no spatial order in memory (clustered or not): two-level bucket scatter
    elif probe is not None and probe["overflow"] > npart_total // 64:
        hint = "clustered"                 # file order, clustered: exact two-pass lists, no capacity limit per tile
    else:
        hint = "ordered"                   # single pass: group records straight from the array
    pstats = {}
    collect = [None]                   # set to `pstats` for ONE untimed step after the timed ones: the same paint, with its list statistics

    def step():
        if fused and method in ("auto", "tiled"):
            # the paint stores rho - mean (subtracted in double before the fp32 rounding) and its halo fold rides
            # on the z pass of the FFT (one kernel and ~2 GB less)
            _, halo = dev.paint(pos, None, n, L, window, out=grid, method="tiled", check_dropped=False,
                                accumulate=False, defer_fold=True, offset=mean, hint=hint, stats=collect[0])
            psum.zero_()
            return dev.power_sums_fused(grid, L, psum=psum, halo=halo)
        if f32_via_double and method in ("auto", "tiled"):
            dev.paint(pos, None, n, L, window, out=grid, method="tiled", check_dropped=False, accumulate=False, offset=mean, hint=hint,
                      stats=collect[0])
            psum.zero_()
            return dev.power_sums_fused64(grid, L, psum=psum, mean=0.0)      # (rho - mean: side 2048 takes its fp32 passes + low-k box; 128 the double passes)
        if fused64 and method in ("auto", "tiled"):     # float64: the halo fold rides on the double z pass too
            _, halo = dev.paint(pos, None, n, L, window, out=grid, method="tiled", check_dropped=False,
                                accumulate=False, defer_fold=True, hint=hint, stats=collect[0])
            psum.zero_()
            return dev.power_sums_fused64(grid, L, psum=psum, halo=halo)
        dev.paint(pos, None, n, L, window, out=grid, method=method, check_dropped=False,
                  accumulate=False, hint=hint, stats=collect[0])       # overwrite mode: no zero-fill pass
        psum.zero_()
        if fused:                         # tile FFT with the shell binning fused into the last pass
            return dev.power_sums_fused(grid, L, psum=psum, mean=mean)
        if fused64:                       # float64: hand-written double passes, binning fused into the x pass
            return dev.power_sums_fused64(grid, L, psum=psum)
        dev.r2c(grid, out=spec)
        return dev.power_bin_1d(spec, None, n, L, psum=psum)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    dev.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        sums = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = dev.profile_report()
    dev.profile_enable(False)
    ms_per_step = elapsed / steps * 1e3
    # sanity: the spectrum that was timed is a real one (finite, positive at Nyquist-ish k)
    res = dev.finish_power(*sums)
    assert np.isfinite(res["power"]).all() and res["power"][-1] > 0
    overlapped = ("paint_tiled.fill", "paint_tiled.deposit") if "paint_tiled.pipeline" in prof else ()
    stages, stage_sites, stage_bytes = _stage_table(prof, steps, npart_total, n ** 3, esz, fused_bin=True,
                                                    concurrent=(("fft_tile.lowk",) if fused else ()) + overlapped)
    dom = max(stages, key=lambda k: stages[k]["ms"])
    traffic, source = (None, None)
    if dom == "paint" and n == 1024 and npside == 1024 and window == "cic" and dtype == "f32" and order == "natural":
        traffic, source = _traffic_from_profiles()
    total = sum(stage_bytes.values())
    roofline = {
        "bound": "hbm", "kernel": f"{dom} stage ({'+'.join(stage_sites[dom])})",
        "achieved": stages[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": stages[dom]["frac"], "traffic": traffic, "traffic_source": source,
        "end_to_end": {"alg_GB": round(total / 1e9, 3), "GBps": round(total / (ms_per_step * 1e6), 1),
                       "frac": round(total / (ms_per_step * 1e6) / HBM_PEAK_GBS, 4)},
        "stages": stages,
    }
    # which path the timed paint took, how often it was attempted (always once: the path is chosen up front), what went
    # through the overflow list: ONE more step, untimed, identical but for the list statistics it fetches
    collect[0] = pstats
    step()
    collect[0] = None
    torch.cuda.synchronize()
    path = {"hint": hint, "path": pstats.get("path"), "attempts": pstats.get("attempts"), "overflow_list": pstats.get("overflow"),
            "probe_ms": round(probe_ms, 3),
            "probe": None if probe is None else {"groupable_runs": round(probe["groupable"], 3), "est_overflow": probe["overflow"],
                                                 "est_max_tile_over_mean": round(probe["max_tile"] / max(1.0, probe["mean_tile"]), 1)}}
    del pos, grid, spec
    torch.cuda.empty_cache()
    return {"ms_per_step": ms_per_step, "roofline": roofline, "paint_path": path}


def slab_leg(dev, dist, n, npside, L, args, world, barrier, wd):
    from astrild_amd import slab
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    esz = 4 if args.dtype == "f32" else 8
    # shuffled order: a rank's particles lie anywhere in the box, so they are routed to their slabs first (all-to-all-v)
    pipe = slab.SlabPowerPipeline(n, L, npside, window=args.window, dtype=tdt, seed=20240601,
                                  shuffle=(args.order == "shuffled"), route=(args.order == "shuffled"),
                                  ghost=3)     # base cells up to 3 planes (6 sigma of the jitter) outside the slab; checked below
    # first contact with RCCL at N > 1: every wait below is bounded by the watchdog, which exits this process non-zero
    # with the rank, its host stage and the last schedule entry its GPU completed (a fresh exit, never a re-exec)
    wd.watch(pipe)
    wd.beat("first step (checked, with progress markers)")
    pipe.step(check=True, progress=True)      # once, untimed: no deposit may fall outside the ghost zone
    torch.cuda.synchronize()
    wd.beat("warm-up steps")
    for _ in range(args.warmup):
        pipe.step()
    barrier()
    wd.beat("timed steps")
    # the timed steps run WITHOUT the per-launch-site event pairs: a staged step makes ~45 small launches, and the events
    # around them cost 0.2-0.4 ms per step (measured, scripts/perf_slab_staged.py); the per-site kernel times of the line
    # come from `prof_steps` extra steps right after the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sums = pipe.step()
    barrier()
    elapsed = time.perf_counter() - t0
    wd.beat("timed steps done; profiled steps")
    prof_steps = max(2, min(args.steps, 5))
    pipe.stage_ms(args.steps)            # (reset the host-stage clock)
    dev.profile_enable(True)
    for _ in range(prof_steps):
        pipe.step()
    barrier()
    prof = dev.profile_report()
    dev.profile_enable(False)
    prof = {k: (v[0], v[1] * args.steps / prof_steps) for k, v in prof.items()}      # scaled to the timed step count
    wd.beat("profiled steps done")
    # diagnostics of the N > 1 line: every rank's own wall time and per-site kernel times, and what a step puts on
    # the links (so that the first run on real xGMI says where the time went)
    mine = {"rank": dist.get_rank(), "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "kernels_ms": {k: round(v[1] / args.steps, 4) for k, v in prof.items()},
            "stage_ms": {k: round(v, 4) for k, v in pipe.stage_ms(prof_steps).items()}}
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    res = dev.finish_power(*sums)
    assert np.isfinite(res["power"]).all() and res["power"][-1] > 0
    overlapped = ("paint_tiled.fill", "paint_tiled.deposit") if "paint_tiled.pipeline" in prof else ()
    stages, stage_sites, stage_bytes = _stage_table(prof, args.steps, npside ** 3 / world, n ** 3 / world, esz, fused_bin=False,
                                                    concurrent=overlapped)
    dom = max(stages, key=lambda k: stages[k]["ms"])
    total = sum(stage_bytes.values())
    roofline = {
        "bound": "hbm", "kernel": f"{dom} stage ({'+'.join(stage_sites[dom])}), rank 0 of {world}",
        "achieved": stages[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": stages[dom]["frac"], "traffic": None,
        "end_to_end": {"alg_GB": round(total * world / 1e9, 3), "GBps": round(total * world / (ms_per_step * 1e6), 1),
                       "frac": round(total / (ms_per_step * 1e6) / HBM_PEAK_GBS, 4)},
        "stages": stages, "ranks": world, "backend": dist.get_backend(),
    }
    return {"ms_per_step": ms_per_step, "roofline": roofline,
            "diag": {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                     "wire_bytes_per_step_per_rank": pipe.wire_bytes(), "pipeline": pipe.pipeline,
                     "kernel_times_from": f"{prof_steps} steps run right after the timed region with the per-site events on "
                                          "(the timed steps run without them)",
                     "schedule": [list(e) for e in pipe.schedule] if pipe.schedule else None, "chunks": pipe.chunks,
                     "spectrum_row_pitch": pipe.nzp, "nx_alloc": pipe.nx_alloc,
                     "transpose_layout": ("rows" if pipe.disc is None else
                                          {"kind": "disc (only what FFTPower keeps; k_y row blocks dealt to the ranks by disc area)",
                                           "plane_elems_per_part": pipe.disc["S"], "full_pitched_plane_elems": pipe.nloc * pipe.nzp}),
                     "per_rank": per_rank}}


if __name__ == "__main__":
    main()
