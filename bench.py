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
    ap.add_argument("--slab", type=int, default=0, help="run the slab-decomposed pipeline even on one GPU (rehearsal)")
    return ap.parse_args()


def cpu_baseline(sample, window, boxsize):
    """The oracle (numpy port of the reference's CPU path: pmesh-convention paint,
    rfftn, FFTPower binning; float64 like the reference) on a bounded sample of the
    workload: sample^3 particles on a sample^3 grid, single thread."""
    from oracle import mesh as omesh, fftpower as offt
    pos = omesh.lattice_particles(sample, sample, boxsize, seed=20240601)
    t0 = time.perf_counter()
    grid = omesh.paint(pos, None, sample, boxsize, window)
    t1 = time.perf_counter()
    offt.fftpower_1d(grid, boxsize)
    t2 = time.perf_counter()
    n = pos.shape[0]
    return {
        "value": n / (t2 - t0), "unit": "particles/s", "cores": 1, "kind": "port",
        "sample": f"{sample}^3 particles on a {sample}^3 grid, float64, numpy bincount paint {t1 - t0:.2f}s + "
                  f"numpy rfftn/shell binning {t2 - t1:.2f}s; host has {os.cpu_count()} logical cores",
    }


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
    dev.bispectrum(grid, L, edges, tri)                      # warm-up: plans, triangle counts (cached)
    torch.cuda.synchronize()
    dev.profile_enable(True)
    t0 = time.perf_counter()
    res = dev.bispectrum(grid, L, edges, tri)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = dev.profile_report()
    dev.profile_enable(False)
    ng = n ** 3
    alg = nsh * (2 * 4 * ng + 24 * ng) + len(tri) * 3 * 4 * ng      # per shell: filter R+W + c2r 3 passes; per triangle: 3 fields
    return {"metric": f"bispectrum on {n}^3 grid: {nsh} shells of width {width} k_F, {len(tri)} triangle bins, fp32",
            "value": len(tri) / dt, "unit": "triangle bins/s", "ms_total": dt * 1e3,
            "alg_GB": round(alg / 1e9, 2), "GBps": round(alg / dt / 1e9, 1), "frac": round(alg / dt / 1e9 / HBM_PEAK_GBS, 4),
            "ntri_total": int(np.sum(res["ntri"])), "kernels_ms": {k: round(v[1], 3) for k, v in prof.items()}}


def kappa_leg(dev, steps, warmup):
    """64 planes x 4096^2 fp64 -> stack -> Gaussian FFT smoothing -> kappa->alpha (config D)."""
    from astrild_amd import lensing
    return lensing.bench_kappa_pipeline(nplanes=64, npix=4096, steps=max(2, min(steps, 5)), warmup=min(warmup, 2))


def main():
    args = parse()
    # stdout carries exactly ONE JSON line: everything else a library may print there
    # (RCCL's version banner at communicator creation, for one) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
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
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from astrild_amd import device as dev
    n = args.ngrid
    npside = args.npside or n
    L = 1000.0
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    esz = 4 if args.dtype == "f32" else 8
    npart_total = npside ** 3

    if not use_slab:
        pos = dev.synth_lattice_particles(npside, n, L, seed=20240601, shuffle=(args.order == "shuffled"), dtype=tdt)
        grid = torch.empty((n, n, n), dtype=tdt, device="cuda")
        spec = torch.empty((n, n, n // 2 + 1), dtype=torch.complex64 if args.dtype == "f32" else torch.complex128,
                           device="cuda")
        psum = torch.zeros(n // 2 - 1, dtype=torch.float64, device="cuda")
        dev.shell_geometry(n, L)          # data independent, cached like the FFT plan

        fused = dev.fused_power_supported(grid)

        def step():
            if fused and args.method in ("auto", "tiled"):
                # the paint's halo fold rides on the z pass of the FFT (one kernel and ~2 GB less)
                _, halo = dev.paint(pos, None, n, L, args.window, out=grid, method="tiled", check_dropped=False,
                                    accumulate=False, defer_fold=True)
                psum.zero_()
                return dev.power_sums_fused(grid, L, psum=psum, mean=npart_total / float(n) ** 3, halo=halo)
            dev.paint(pos, None, n, L, args.window, out=grid, method=args.method, check_dropped=False,
                      accumulate=False)       # overwrite mode: no zero-fill pass
            psum.zero_()
            if fused:                         # tile FFT with the shell binning fused into the last pass
                return dev.power_sums_fused(grid, L, psum=psum, mean=npart_total / float(n) ** 3)
            dev.r2c(grid, out=spec)
            return dev.power_bin_1d(spec, None, n, L, psum=psum)
    else:
        from astrild_amd import slab
        pipe = slab.SlabPowerPipeline(n, L, npside, window=args.window, dtype=tdt, seed=20240601,
                                      shuffle=(args.order == "shuffled"))
        pipe.step(check=True)             # once, untimed: no deposit may fall outside the ghost zone
        step = pipe.step

    def barrier():
        if use_slab:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    dev.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sums = step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = dev.profile_report()
    dev.profile_enable(False)
    if use_slab:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = npart_total / (elapsed / args.steps)

    # sanity: the spectrum that was timed is a real one (finite, positive at Nyquist-ish k)
    res = dev.finish_power(*sums)
    assert np.isfinite(res["power"]).all() and res["power"][-1] > 0

    # ---- roofline per logical stage: SURVEY.md §8(d) algorithmic bytes / HIP-event time ----
    npart_rank = npart_total / world
    ng_rank = n ** 3 / world
    stage_sites = {
        "paint": [k for k in prof if k.startswith("paint")],
        "fft": [k for k in prof if k.startswith("rocfft") or k.startswith("fft_tile") or k.startswith("slab.")],
        "power_bin": [k for k in prof if k == "power_bin"],        # absent when fused into the last FFT pass
    }
    stage_bytes = {
        "paint": npart_rank * 3 * esz + ng_rank * esz,       # read positions once, write the grid once
        "fft": 3 * 2 * ng_rank * esz,                         # 3 axis passes x (read + write)
        "power_bin": ng_rank * esz,                           # half spectrum read once (~esz B per real cell)
    }
    if not use_slab and not stage_sites["power_bin"]:
        stage_bytes["fft"] += stage_bytes.pop("power_bin")    # fused: one stage carries both terms
        stage_sites.pop("power_bin")
    stages = {}
    for name, sites in stage_sites.items():
        ms = sum(prof[s][1] for s in sites) / args.steps
        if ms > 0:
            gbs = stage_bytes[name] / ms / 1e6
            stages[name] = {"ms": round(ms, 4), "alg_GB": round(stage_bytes[name] / 1e9, 3),
                            "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                            "kernels": {s: round(prof[s][1] / args.steps, 4) for s in sites}}
    dom = max(stages, key=lambda k: stages[k]["ms"])
    # HBM traffic of the dominant stage from the PMC passes of the same command (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate runs, gfx950 x2 read correction; profiles/r01_pmc_traffic.json says how)
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if dom == "paint" and not use_slab and n == 1024 and npside == 1024 and args.window == "cic" \
            and args.dtype == "f32" and args.order == "natural" and os.path.isfile(pmc_file):
        traffic = json.load(open(pmc_file)).get("paint_stage_corrected_GB_per_step")
        traffic = None if traffic is None else traffic * 1e9
    roofline = {
        "bound": "hbm", "kernel": f"{dom} stage ({'+'.join(stage_sites[dom])})",
        "achieved": stages[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": stages[dom]["frac"], "traffic": traffic,
        "end_to_end": {"alg_GB": round(sum(stage_bytes.values()) / 1e9, 3),
                       "GBps": round(sum(stage_bytes.values()) * world / (ms_per_step * 1e6), 1),
                       "frac": round(sum(stage_bytes.values()) * world / (ms_per_step * 1e6) / (HBM_PEAK_GBS * world), 4)},
        "stages": stages,
    }

    out = {
        "metric": "particles/sec CIC+3D-FFT P(k) on 1024^3 grid" if n == 1024 and args.window == "cic"
                  else f"particles/sec {args.window.upper()}+3D-FFT P(k) on {n}^3 grid",
        "value": value, "unit": "particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{npside}^3 lattice+Gaussian(0.5 cell) particles ({args.order} order) -> "
                               f"{args.window.upper()} paint on {n}^3 grid -> 3D R2C -> FFTPower 1d shells",
                   "ngrid": n, "nparticles": npart_total, "boxsize": L,
                   "parallelism": "single GPU" if not use_slab else f"axis-0 slabs x{world}, RCCL all-to-all transpose"},
        "roofline": roofline,
    }
    if rank == 0 and not use_slab:
        if args.cpu_sample:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.window, L)
        del pos, grid, spec
        torch.cuda.empty_cache()
        if args.bispec:
            out["bispectrum"] = bispectrum_leg(dev)
            dev._tri_cache.clear()
            torch.cuda.empty_cache()
        if args.kappa:
            out["kappa"] = kappa_leg(dev, args.steps, args.warmup)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_slab:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
