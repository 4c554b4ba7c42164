"""One rank of P (default: rank 3 of 8) of the 1024^3 slab pipeline on ONE GPU with the communication stubbed out: the
rank's real host sequence and every kernel it launches per step, staged (walk -> fold -> transform per stage) against
bulk (paint everything, then transform), with the times at which each plane range is ready to be sent.
    python scripts/perf_slab_staged.py [P] [rank] [ghost]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from astrild_amd import device as dev, slab

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
RANK = int(sys.argv[2]) if len(sys.argv) > 2 else 3
GHOST = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n, L = 1024, 1000.0
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
dist.init_process_group("gloo", rank=0, world_size=1)
# the geometry of rank RANK of P, no exchange: sends / receives are dropped (the receive block keeps whatever it holds),
# all-reduces act on the rank's own values
dist_world, dist_rank = dist.get_world_size, dist.get_rank
dist.get_world_size = lambda group=None: P
dist.get_rank = lambda group=None: RANK
dist.all_reduce = lambda t, *a, **k: None
slab.exchange_planes = lambda *a, **k: []
slab.exchange_planes_disc = lambda *a, **k: []
slab.GhostExchange.start = lambda self: None
slab.GhostExchange.start_upper = lambda self: None
slab.GhostExchange.start_lower = lambda self: None
def _finish(self):
    owned = self.buf[self.gl: self.gl + self.nloc]
    self.ops.add_into(owned[self.nloc - self.gl:], self.from_right)
    self.ops.add_into(owned[:self.gh], self.from_left)
slab.GhostExchange.finish = _finish
slab.comm_ready = lambda group=None: None


def run(pipeline, rps=None, streams=1, reps=7, parts=None):
    os.environ["ASTRILD_SLAB_STREAMS"] = str(streams)
    pipe = slab.SlabPowerPipeline(n, L, n, window="cic", dtype=torch.float32, ghost=GHOST, pipeline=pipeline, rows_per_stage=rps,
                                  group_chunks=parts)
    if pipe.ghosts is not None:
        pipe.ghosts.from_left.zero_(); pipe.ghosts.from_right.zero_()
    for _ in range(2):
        pipe.step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:                      # plain steps: what a rank does per step
        a.record(); pipe.step(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        pipe.step()
    host_ms = (time.perf_counter() - t0) / reps * 1e3       # host time to ENQUEUE a step (the GPU runs behind)
    torch.cuda.synchronize()
    ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    dev.profile_enable(True)             # the same with an event pair around every launch site
    for a, b in ev2:
        a.record(); pipe.step(); b.record()
    torch.cuda.synchronize()
    prof = dev.profile_report()
    dev.profile_enable(False)
    ms_prof = sorted(a.elapsed_time(b) for a, b in ev2)[reps // 2]
    print(f"--- {pipeline} rows_per_stage={rps or 'auto'} streams={streams} group_parts={pipe.group_chunks} send_planes={pipe.send_planes}: step (compute only, no exchange) median {ms[reps // 2]:.3f} ms, min {ms[0]:.3f}; "
          f"buffer {pipe.nx_alloc} planes; with per-site profiling events {ms_prof:.3f} ms; host enqueue {host_ms:.3f} ms per step", flush=True)
    print("    kernels per step:", {k: round(v[1] / reps, 3) for k, v in prof.items()}, " sum", round(sum(v[1] for v in prof.values()) / reps, 3))
    print("    host enqueue ms per step:", {k: round(v, 3) for k, v in pipe.stage_ms(reps + 2).items()})
    if pipeline == "staged":
        pipe.trace = []
        pipe.step()
        torch.cuda.synchronize()
        pipe_trace = list(pipe.trace)
        t0 = pipe.trace[0][1]
        sent = 0
        for entry, e in sorted(pipe.trace[1:], key=lambda t: t0.elapsed_time(t[1])):
            tag = ""
            if entry[0] == "fft":
                sent += entry[2]
                tag = f"   -> {sent}/{pipe.nloc} planes ready to send"
            print(f"    {t0.elapsed_time(e):7.3f} ms  {entry}{tag}")
        pipe.trace = None
        # forecast: a piece of p planes is ready at t and needs p * (bytes per plane and link) / B on every link; a link
        # carries its messages one after the other in the order they were enqueued; then the axis-0 pass + binning and the
        # all-reduces (0.1 ms assumed).  The two links to the ring neighbours ALSO carry the ghost planes (upper ghosts to
        # rank + 1 when ghost_start_upper / ghost_start is enqueued, lower ghosts to rank - 1): every rank runs the same
        # schedule, so what this rank waits for at ghost_finish arrives when its own ghost message would; a late arrival
        # delays everything enqueued after ghost_finish.
        wb = pipe.wire_bytes()
        wire = wb["transpose"] / (P - 1) / pipe.nloc / 1e6          # MB per plane and link
        if pipe.disc is not None:
            # disc layout: the link to rank s carries planes of S[s] elements - the fullest link sets the pace (the parts are
            # balanced to a few per cent); every rank runs the same schedule, so this is also what the last receive waits for
            wire = max(pipe.disc["S"]) * 8 / 1e6
            print(f"    disc layout: plane sizes per part {pipe.disc['S']} (a full pitched plane: {pipe.nloc * pipe.nzp}); sent per step "
                  f"{wb['transpose'] / 1e6:.1f} MB, received {wb['transpose_received'] / 1e6:.1f} MB; fullest link {wire * pipe.nloc:.1f} MB")
        ghost_mb = wb["ghost"] / 2 / 1e6                            # MB of ghost planes per neighbour
        tail = (prof.get("fft_tile.c2c_power", (0, 0))[1] + prof.get("fft_tile.shell_reduce", (0, 0))[1]) / reps + 0.1
        times = [(t0.elapsed_time(e), entry) for entry, e in pipe_trace[1:]]
        t_finish = next(t for t, en in times if en[0] == "ghost_finish")
        single = float(os.environ.get("SINGLE_GPU_MS", "13.13"))
        for B in (40, 50, 60, 70, 1e9):
            def link(ghost_kinds, delay):
                """Completion of all messages on one link direction and of its ghost message: FIFO by enqueue time."""
                msgs = []
                for t, en in times:
                    t += delay if t > t_finish else 0.0
                    if en[0] == "fft":
                        msgs.append((t, en[2] * wire))
                    elif en[0] in ghost_kinds:
                        msgs.append((t, -ghost_mb))
                done, ghost_done = 0.0, 0.0
                for t, mb in sorted(msgs, key=lambda m: m[0]):
                    done = max(done, t) + abs(mb) / B
                    if mb < 0:
                        ghost_done = done
                return done, ghost_done
            delay = 0.0
            for _ in range(3):                                      # (the delay moves the last pieces, which moves nothing before them)
                _, g_up = link(("ghost_start", "ghost_start_upper"), delay)
                _, g_lo = link(("ghost_start", "ghost_start_lower"), delay)
                delay = max(0.0, max(g_up, g_lo) - t_finish)
            done = max(link((), delay)[0], link(("ghost_start", "ghost_start_upper"), delay)[0],
                       link(("ghost_start", "ghost_start_lower"), delay)[0])
            step = done + tail
            print(f"    forecast B = {B if B < 1e8 else 'inf':>4} GB/s per link and direction: ghosts in {max(g_up, g_lo):.3f} ms (waited for at {t_finish:.3f}), "
                  f"exchange done {done:.3f} ms, step {step:.3f} ms, {single / step:.2f}x of the single GPU's {single} ms")
    del pipe
    torch.cuda.empty_cache()


if os.environ.get("ONLY_DEFAULT"):
    run("staged", parts=None)
    dist.destroy_process_group()
    sys.exit(0)
if os.environ.get("STAGE_SPECS"):         # e.g. STAGE_SPECS="8:7|0,1,2,3|4,5|6;4:3|0,1|2": parts:stages, one run each
    for item in os.environ["STAGE_SPECS"].split(";"):
        k, spec = item.split(":")
        os.environ["ASTRILD_SLAB_STAGES"] = spec
        print(f"=== parts {k}, stages {spec}", flush=True)
        run("staged", parts=int(k), reps=5)
    dist.destroy_process_group()
    sys.exit(0)
run("bulk")
run("staged", parts=1)               # everything grouped first (first stage half as long)
run("staged", parts=4)               # grouping in four equal parts, the last one first: 3 | 0 1 | 2
run("staged", parts=None)            # the default for x-ordered input: eight parts in four stages, 7 | 0 1 2 | 3 4 5 | 6
run("staged", parts=16)              # 14 15 | 0-5 | 6-10 | 11 12 | 13
dist.destroy_process_group()
