#!/usr/bin/env python3
"""Development: where a 256 x 256 tile's time goes.  Needs a build with -DAPTAI_STAMPS (tools/ab/lib_stamps.so, APTAI_HIP_LIB):
per-block wall-clock stamps (100 MHz) at entry / first operands landed / main loop done / epilogue issued / stores acknowledged."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from aptai_amd import _lib, ops


def run(M, N, K, lda=None, gelu=False, tile=256, km=False, heavy=False):
    g = torch.Generator(device="cuda").manual_seed(0)
    rows = M + 16 if lda is None else (M * lda) // 512 + 64
    a = (torch.randn(rows, 512 if lda else K, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(K, N, device="cuda", generator=g) * 0.03).to(torch.bfloat16) if km else (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
    out = torch.empty(M + 8, N, device="cuda", dtype=torch.bfloat16)
    kw = dict(lda=lda) if lda else {}
    if heavy:
        kw.update(bias=torch.randn(N, device="cuda"), residual=torch.randn(M, N, device="cuda").to(torch.bfloat16), dropout_p=0.1, seed=1)
    for _ in range(3):
        ops.gemm(a, w, M, N, K, out=out, ldc=N, gelu=gelu, tile=tile, b_kmajor=km, **kw)
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    lib = _lib.lib()
    lib.aptai_debug_read_stamps.argtypes = [ctypes.c_void_p]
    lib.aptai_debug_read_stamps.restype = ctypes.c_int
    assert lib.aptai_debug_read_stamps(buf.ctypes.data) == 0
    tm, tn = {256: (256, 256), 128: (128, 128), 192: (128, 192), 64: (64, 128)}[tile]
    nb = min(4096, ((M + tm - 1) // tm) * ((N + tn - 1) // tn))
    st = buf.reshape(4096, 8)[:nb].astype(np.int64)
    t0 = st[:, 0].min()
    rel = (st[:, :5] - t0) / 100.0                      # us
    d = np.diff(rel, axis=1)
    print(f"== tile {tile} {'NN' if km else 'NT'} {M} x {N} x {K} lda={lda} gelu={gelu} heavy={heavy}: {nb} blocks, kernel span {rel[:, 4].max():.1f} us")
    if tile == 256:
        names = ["entry->operands", "main loop", "epilogue", "store ack"]
        for i, n in enumerate(names):
            print(f"   {n:16s} mean {d[:, i].mean():6.2f}  p10 {np.percentile(d[:, i], 10):6.2f}  p90 {np.percentile(d[:, i], 90):6.2f} us")
    else:
        ml = (st[:, 2] - st[:, 0]) / 100.0
        print(f"   prologue + main loop mean {ml.mean():6.2f} us | epilogue mean {d[:, 2].mean():6.2f} p90 {np.percentile(d[:, 2], 90):6.2f} | store ack {d[:, 3].mean():5.2f}")
    print(f"   block total      mean {(rel[:, 4] - rel[:, 0]).mean():6.2f} us")
    # per CU: gap between a block's end and the next block's entry on the same CU
    hw = st[:, 5]
    key = ((hw >> 32) & 0xf) * 4096 + ((hw & 0xffffffff) >> 8 & 0xf) + (((hw & 0xffffffff) >> 13) & 0x7) * 16 + (((hw & 0xffffffff) >> 16) & 0x1) * 128
    gaps = []
    for k in np.unique(key):
        idx = np.where(key == k)[0]
        idx = idx[np.argsort(rel[idx, 0])]
        for a_, b_ in zip(idx[:-1], idx[1:]):
            gaps.append(rel[b_, 0] - rel[a_, 4])
    if gaps:
        gaps = np.array(gaps)
        print(f"   slots seen {len(np.unique(key))}; end->next entry on the same slot: mean {gaps.mean():6.2f}  p10 {np.percentile(gaps, 10):6.2f}  p90 {np.percentile(gaps, 90):6.2f} us")
    print(f"   entry times: first {np.sort(rel[:, 0])[:3]}, block 255 {np.sort(rel[:, 0])[min(255, nb - 1)]:.2f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "loop192":
        # shader-clock cycles of the five segments of the 192-row kernel's K-tile (K-contiguous operands), per K-tile, thread 0 of each block
        for (n, k) in ((768, 768), (768, 3072), (2304, 768)):
            M = 8192
            g = torch.Generator(device="cuda").manual_seed(0)
            a = torch.randn(M, k, device="cuda", generator=g).to(torch.bfloat16)
            w = torch.randn(n, k, device="cuda", generator=g).to(torch.bfloat16)
            for _ in range(3):
                ops.gemm(a, w, M, n, k, tile=192)
            torch.cuda.synchronize()
            buf = np.zeros(4096 * 8, dtype=np.uint64)
            lib = _lib.lib()
            lib.aptai_debug_read_stamps.argtypes = [ctypes.c_void_p]
            lib.aptai_debug_read_stamps.restype = ctypes.c_int
            assert lib.aptai_debug_read_stamps(buf.ctypes.data) == 0
            nb = (M // 128) * (n // 192)
            st = buf.reshape(4096, 8)[:nb]
            nk = k // 64
            seg = np.stack([st[:, 1], st[:, 5], st[:, 6], st[:, 7] & np.uint64(0xffffffff), st[:, 7] >> np.uint64(32)], 1).astype(np.float64) / nk
            names = ["MFMA half 0 (issue)", "wait + barrier", "DMA issue + reads half 0", "MFMA half 1 (issue)", "reads half 1 (issue)"]
            print(f"== 192-row tile NT {M} x {n} x {k}: cycles per K-tile (mean over {nb} blocks), total {seg.sum(1).mean():.0f}")
            for i, nm in enumerate(names):
                print(f"   {nm:26s} {seg[:, i].mean():7.0f}")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "small":
        for tile in (128, 192, 64):
            run(8192, 768, 768, tile=tile)
            run(8192, 768, 768, tile=tile, heavy=True)
            run(8192, 2304, 768, tile=tile)
            run(8192, 768, 3072, tile=tile, km=True)
        sys.exit(0)
    run(8192, 2048, 768)
    run(8192, 2048, 3072)
    run(255984, 512, 1536, lda=1024)
    run(255984, 512, 1536, lda=1024, gelu=True)
