"""Build profiles/<tag>_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
usage: pmc_traffic_json.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, hashlib, json, os, sys, collections

def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"][:110]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}

fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
kernels = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    kernels[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "corrected_GB": round((2 * f + w) * 1024 / 1e9, 3), "launches": nf.get(k, 0)}
paint = sum(v["corrected_GB"] for k, v in kernels.items()
            if any(t in k for t in ("tile_index_kernel", "tile_group_kernel", "column_deposit_kernel", "column_fold_kernel",
                                    "overflow_deposit_kernel", "tile_deposit_kernel", "scatter_level_a_kernel",
                                    "scatter_level_b_kernel", "late_deposit_kernel", "z_seam_kernel")))
out = {
    "_doc": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --cpu-sample 0 --kappa 0 --bispec 0 "
            "--legs 0 --steps 2 --warmup 1" + (" " + " ".join(sys.argv[4:]) if len(sys.argv) > 4 else "") + "` on MI355X. Counter values are KiB per launch (mean over launches). Corrected bytes = "
            "(2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B). Calibration: the "
            "synth kernel writes the 12.885 GB of positions exactly once (WRITE_SIZE exact); the index kernel reads them once and "
            "2*FETCH_SIZE*1024 matches for 4/12-byte-per-lane loads. For 8/16-byte-per-lane loads (fft_tile, fold) the x2 "
            "under-counts reads (uncalibrated width); their WRITE_SIZE is exact.",
    "kernels": kernels,
    "paint_src_sha256_16": hashlib.sha256(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "astrild_amd",
                                                            "csrc", "mesh_paint_tiled.hip"), "rb").read()).hexdigest()[:16],
    "paint_stage_corrected_GB_per_step": round(paint, 2),
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("paint stage corrected GB/step:", round(paint, 2))
