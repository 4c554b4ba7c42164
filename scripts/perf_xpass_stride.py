"""Does the x pass (axis-0 transform fused with the shell binning) depend on how far apart its 1024 row pieces lie?
The same bytes are transformed as one (n, n, nz) block (row stride n * nz * 8 B = 4.2 MB) and as n / nloc blocks of
(n, nloc, nz) (row stride nloc * nz * 8 B).  Run on the GPU box: python scripts/perf_xpass_stride.py"""
import sys
import time
import torch
sys.path.insert(0, ".")
from astrild_amd import _lib, device as dev

L = _lib.lib()
n = 1024
nz = n // 2 + 1
pitch = (nz + 15) // 16 * 16
psum = torch.zeros(n // 2 - 1, dtype=torch.float64, device="cuda")
for nloc in (1024, 256, 64, 16):
    nblocks = n // nloc
    blocks = [torch.view_as_complex(torch.randn((n, nloc, pitch, 2), dtype=torch.float32, device="cuda")) for _ in range(min(nblocks, 4))]
    scratch = torch.empty(int(L.ast_fft_tile_block_power_scratch_bytes(n, nloc)), dtype=torch.uint8, device="cuda")

    def run():
        for b in range(nblocks):
            blk = blocks[b % len(blocks)]
            _lib.check(L.ast_fft_tile_block_power(dev.ptr(blk), dev.ptr(scratch), scratch.numel(), 0, n, nloc, b * nloc, pitch,
                                                  1.0, 1000.0, 0, 1, dev.ptr(psum), dev.stream()), "block_power")
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    print(f"nloc {nloc:5d}: row stride {nloc * pitch * 8 / 1e6:8.3f} MB, {nblocks:3d} launches, {(time.perf_counter() - t0) / 5 * 1e3:7.3f} ms", flush=True)
    del blocks, scratch
