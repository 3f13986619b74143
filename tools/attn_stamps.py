#!/usr/bin/env python3
"""Development: where an attention block's time goes (needs a -DAPTAI_STAMPS build, APTAI_HIP_LIB=tools/ab/lib_stamps.so)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from aptai_amd import _lib, ops

B, Tp, heads = 16, 512, 12
H = heads * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * Tp, 3 * H, device="cuda", generator=g).to(torch.bfloat16)
dctx = torch.randn(B * Tp, H, device="cuda", generator=g).to(torch.bfloat16)
lens = torch.tensor([499] * 16, dtype=torch.int32, device="cuda")
for p in (0.0, 0.1):
    for _ in range(3):
        ctx, st = ops.attention_fwd(qkv, lens, B, Tp, H, heads, dropout_p=p, seed=1, q_prescaled=True)
        ops.attention_bwd(qkv, lens, ctx, dctx, st, B, Tp, H, heads, dropout_p=p, seed=1, dctx_zero_beyond_len=True, q_prescaled=True)
    torch.cuda.synchronize()
    buf = np.zeros(3 * 1024 * 8, dtype=np.uint64)
    lib = _lib.lib()
    lib.aptai_debug_read_attn_stamps.argtypes = [ctypes.c_void_p]
    lib.aptai_debug_read_attn_stamps.restype = ctypes.c_int
    assert lib.aptai_debug_read_attn_stamps(buf.ctypes.data) == 0
    st = buf.reshape(3, 1024, 8)[:, :768].astype(np.int64)
    for k, name in enumerate(("fwd", "dK/dV", "dQ")):
        s = st[k]
        rel = (s[:, :5] - s[:, 0].min()) / 100.0
        d = np.diff(rel, axis=1)
        print(f"p={p} {name:6s}: span {rel[:, 4].max():5.1f} us | entry spread {rel[:, 0].max():4.1f} | prologue {d[:, 0].mean():5.2f} | loop {d[:, 1].mean():5.2f} "
              f"| epilogue {d[:, 2].mean():5.2f} | store ack {d[:, 3].mean():5.2f} | block {(rel[:, 4] - rel[:, 0]).mean():5.2f} us")
        if k == 2:
            e = (s[:, 5:8] - s[:, 0:1]) / 100.0
            print(f"         dQ prologue: rows staged at {e[:, 0].mean():5.2f}, barrier passed {e[:, 1].mean():5.2f}, delta done {e[:, 2].mean():5.2f} us after entry")
