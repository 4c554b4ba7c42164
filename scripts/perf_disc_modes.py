"""A/B of the single-GPU power pipeline's k_y / last-pass handoff: AST_FFT_DISC unset (in place, pruned), 1 (disc layout,
plane-major), 2 (disc layout, x-major tiles).  Runs bench.py's headline leg once per mode.  usage: python scripts/perf_disc_modes.py"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for mode in ("", "1", "2"):
    env = dict(os.environ)
    env.pop("AST_FFT_DISC", None)
    if mode:
        env["AST_FFT_DISC"] = mode
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "3", "--cpu-sample", "0", "--kappa", "0",
                          "--bispec", "0", "--legs", "0"] + sys.argv[1:], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    k = d["roofline"]["stages"]["fft"]["kernels"]
    print(f"AST_FFT_DISC={mode or '-'}: step {d['ms_per_step']:.3f} ms  " + "  ".join(f"{a.split('.')[-1]}={b:.3f}" for a, b in k.items()), flush=True)
