"""Per-dispatch FETCH_SIZE / WRITE_SIZE of the bispectrum's masked inverse passes (scripts/bispec_shells_once.py run under
`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`) beside the bytes the pruning needs at least.
usage: pmc_per_launch.py <fetch_dir> <write_dir>"""
import csv, glob, math, sys


def rows(d, counter):
    out = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    return sorted(out)


fe, wr = rows(sys.argv[1], "FETCH_SIZE"), rows(sys.argv[2], "WRITE_SIZE")
n, nz, width = 512, 257, 8
inv = [(i, k, v) for i, k, v in fe if "strided_c2c_kernel<16, 32, 16, false, true" in k or "rows_c2r_kernel" in k]
winv = {i: v for i, k, v in wr}
print("shell m_hi | x pass R W (model R W) | y pass R W (model R W) | z pass R W (model R W)   [MB; R = 2 x FETCH_SIZE]")
tot = [0.0] * 12
for s in range(len(inv) // 3):
    m = 1 + width * (s + 1)
    cols16 = 0                                   # (k_y, 16-column k_z tile) pairs inside the disc, as the x pass prunes
    for ky in range(-n // 2 + 1, n // 2 + 1):
        for c0 in range(0, nz, 16):
            if ky * ky + c0 * c0 < m * m:
                cols16 += min(16, nz - c0)
    kz_t = sum(min(16, nz - c0) for c0 in range(0, nz, 16) if c0 < m)
    model = [cols16 * min(n, 2 * m) * 8, cols16 * n * 8, cols16 * n * 8, n * n * kz_t * 8, n * n * min(m, nz) * 8, n * n * n * 4]
    line, k = f"{s:3d} {m:4d} |", 0
    for p in range(3):
        i, kn, f = inv[3 * s + p]
        r_, w_ = 2 * f * 1024 / 1e6, winv[i] * 1024 / 1e6
        line += f" {r_:7.1f} {w_:7.1f} ({model[2 * p] / 1e6:7.1f} {model[2 * p + 1] / 1e6:7.1f}) |"
        for j, v in enumerate((r_, w_, model[2 * p] / 1e6, model[2 * p + 1] / 1e6)):
            tot[4 * p + j] += v
    print(line)
print("sum [GB]  |" + "".join(f" {tot[4*p]/1e3:7.2f} {tot[4*p+1]/1e3:7.2f} ({tot[4*p+2]/1e3:7.2f} {tot[4*p+3]/1e3:7.2f}) |" for p in range(3)))
