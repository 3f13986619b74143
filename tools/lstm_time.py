"""Device time of the cooperating LSTM kernels at the Force_APTAI shape (B = 16, T = 499)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops


def main():
    B, T, Tp = 16, 499, 512
    g = torch.Generator(device="cuda").manual_seed(0)
    xproj = torch.randn(B * Tp, 2048, device="cuda", generator=g) * 0.5
    whh = torch.randn(2, 1024, 256, device="cuda", generator=g) * 0.05
    lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
    dh = torch.randn(B * Tp, 512, device="cuda", generator=g)
    hout, gates, cst = ops.lstm_fwd(xproj, whh, lens, B, Tp, T)
    ops.lstm_bwd(dh, whh, lens, gates, cst, B, Tp, T)
    torch.cuda.synchronize()
    for name, fn in (("fwd", lambda: ops.lstm_fwd(xproj, whh, lens, B, Tp, T)), ("bwd", lambda: ops.lstm_bwd(dh, whh, lens, gates, cst, B, Tp, T))):
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"lstm {name}: {min(ts):.3f} ms ({min(ts) / T * 1e3:.2f} us per frame), status {ops.lstm_status(xproj.device)}", flush=True)


if __name__ == "__main__":
    main()
