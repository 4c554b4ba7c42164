import os, sys, subprocess
for sp in (16, 24, 30, 32, 40, 64):
    env = dict(os.environ, ASTRILD_SLAB_SEND_PLANES=str(sp), ONLY_DEFAULT="1")
    out = subprocess.run([sys.executable, "scripts/perf_slab_staged.py"], env=env, capture_output=True, text=True).stdout
    for line in out.splitlines():
        if line.startswith("---") or "forecast B =   60" in line or "forecast B =  inf" in line:
            print(line[:170])
