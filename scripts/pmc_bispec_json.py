"""HBM traffic of ONE bispectrum evaluation (config E) from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py --cpu-sample 0 --kappa 0 --legs 0 --steps 1 --warmup 0`, whose bispectrum leg calls the estimator twice (the
first call also computes the cached triangle counts).  Corrected bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md).
usage: pmc_bispec_json.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, hashlib, json, os, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def src_sha(name):
    return hashlib.sha256(open(os.path.join(ROOT, "astrild_amd", "csrc", name), "rb").read()).hexdigest()[:16]


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"][:140]].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
CALLS = 2
# kernels of the estimator's numerator (every call): forward transform of the grid at 512^3, masked pruned inverse passes,
# the triangle sums over the fp32 shell fields.  The triangle counts (first call only) use the double-precision kernels.
numerator = {
    "rows_r2c_kernel<16, 16": "forward z pass (512^3)", "strided_c2c_kernel<16, 32, 16, false, false, false": "forward y / x passes",
    "strided_c2c_kernel<16, 32, 16, false, true": "masked, pruned inverse x / y passes (31 shells x 2)",
    "rows_c2r_kernel<16, 16": "inverse z pass (31 shells)", "triple_sums_kernel<float": "all triangle sums, one pass over the 31 fields",
}
out, total = {}, 0.0
for key, what in numerator.items():
    fk = [k for k in fetch if key in k]
    if not fk:
        continue
    f = sum(sum(fetch[k]) for k in fk)
    w = sum(sum(write.get(k, [])) for k in fk)
    n = sum(len(fetch[k]) for k in fk)
    gb = (2 * f + w) * 1024 / 1e9 / CALLS
    out[key] = {"what": what, "launches_per_call": n / CALLS, "corrected_GB_per_call": round(gb, 3)}
    total += gb
# what the file belongs to: bench.py's bispectrum leg (n = 512, shells of width 8 k_F, its triangle list) and the kernel
# sources profiled - bench.py quotes the file only while all of them still match
json.dump({"_doc": __doc__, "kernels": out, "numerator_corrected_GB_per_call": round(total, 2),
           "config": {"n": 512, "shell_width": 8, "shells": 31, "triangle_bins": 75},
           "src_sha256_16": {f: src_sha(f) for f in ("fft_tile.hip", "power_bin.hip")}}, open(sys.argv[3], "w"), indent=1)
print("bispectrum numerator, corrected GB per call:", round(total, 2))
