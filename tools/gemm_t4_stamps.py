#!/usr/bin/env python3
"""Development: where a K-tile of the 256 x 192 kernel spends its cycles.  Needs tools/ab/t4/lib_t4_stamps.so (gemm_t4.hip built with
-DAPTAI_T4_STAMPS) as APTAI_HIP_LIB.  Prints, for wave 0 (group 0) and wave 4 (group 1), shader cycles per K-tile in each segment
(median over the blocks): reads (issue -> landed) | DMA issue | vmcnt + lgkmcnt waits | barrier | MFMA issue | vmcnt wait | barrier."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from aptai_amd import _lib, ops


def run(M, N, K, km=False):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    w = ((torch.randn(K, N, device="cuda", generator=g) if km else torch.randn(N, K, device="cuda", generator=g)) * 0.03).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, M, N, K, out=out, tile=448, b_kmajor=km)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm(a, w, M, N, K, out=out, tile=448, b_kmajor=km)
    e1.record()
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 2 * 16, dtype=np.uint64)
    lib = _lib.lib()
    lib.aptai_debug_read_t4_stamps.argtypes = [ctypes.c_void_p]
    lib.aptai_debug_read_t4_stamps.restype = ctypes.c_int
    assert lib.aptai_debug_read_t4_stamps(buf.ctypes.data) == 0
    nb = min(1024, ((M + 255) // 256) * ((N + 191) // 192))
    raw = buf.reshape(1024, 2, 16)[:nb].astype(np.float64)
    clk = np.median(raw[:, 0, 14] / np.maximum(raw[:, 0, 15], 1.0)) * 100.0
    print(f"   in-kernel clock over the main loop (shader cycles / 100 MHz ticks, median over blocks): {clk:.0f} MHz")
    st = raw[:, :, :14] / (K // 64)
    names = ["reads", "dma", "waits", "barrier", "mfma", "vmcnt", "barrier"]
    print(f"== 256x192 {'NN' if km else 'NT'} {M} x {N} x {K}: {nb} blocks, {e0.elapsed_time(e1) * 100:.1f} us per launch (instrumented); cycles per K-tile, median over blocks")
    for grp in (0, 1):
        med = np.median(st[:, grp, :], axis=0)
        print(f"   group {grp}: " + "  ".join(f"{n} {med[i]:6.0f}" for i, n in enumerate(names)) + f"   | K-tile total {med[:7].sum():6.0f} cycles (each stamp costs ~56)")


if __name__ == "__main__":
    run(8192, 3072, 768)
    run(8192, 2304, 768)
    run(4096, 3072, 1024)
    run(8192, 3072, 768, km=True)
