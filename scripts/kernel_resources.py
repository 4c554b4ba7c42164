"""Per-kernel VGPRs / spills / LDS / occupancy of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python scripts/kernel_resources.py astrild_amd/csrc/mesh_paint_tiled.hip [name filter] [extra hipcc flags...]"""
import re, subprocess, sys
src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-munsafe-fp-atomics",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: .*?Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark: .*?\s+(VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" [")[0]] = int(m.group(2))
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["name"])
    name = re.sub(r"\(.*", "", name)
    if filt in name:
        print(f'{name:70s} vgpr {r.get("VGPRs",0):4d} sspill {r.get("SGPRs Spill",0):3d} vspill {r.get("VGPRs Spill",0):3d} '
              f'scratch {r.get("ScratchSize",0):4d} lds {r.get("LDS Size",0):6d} occ {r.get("Occupancy",0)}')
