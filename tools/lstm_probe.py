#!/usr/bin/env python3
"""Time of the cooperative BiLSTM kernels (csrc/lstm.hip) at the Force_APTAI shape (16 utterances x 499 frames, hidden 256): forward and
backward, per frame, with the operands warm (back-to-back launches) and cold (1 GB written in between).  With a -DAPTAI_EXP_LSTM=<bits>
development build selected through APTAI_HIP_LIB it prices what the per-frame loads / stores beside the exchange cost (bit 1: forward
without input-projection loads, 2: forward without per-frame result stores, 4: backward without saved-state loads, 8: backward without
per-frame stores; results are wrong then - timing only)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aptai_amd import ops


def main():
    B, T, Tp, H = 16, 499, 512, 256
    g = torch.Generator(device="cuda").manual_seed(0)
    xproj = torch.randn(B * Tp, 2 * 4 * H, device="cuda", generator=g) * 0.5
    whh = torch.randn(2, 4 * H, H, device="cuda", generator=g) * 0.05
    lens = torch.full((B,), T, device="cuda", dtype=torch.int32)
    dh = torch.randn(B * Tp, 2 * H, device="cuda", generator=g) * 0.1
    junk = torch.empty(256 * 1024 * 1024, device="cuda", dtype=torch.float32)
    hout, gates, cst = ops.lstm_fwd(xproj, whh, lens, B, Tp, T)
    torch.cuda.synchronize()

    def timed(fn, cold):
        ts = []
        for _ in range(5):
            if cold:
                junk.fill_(1.0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return statistics.median(ts)

    for name, fn in (("forward", lambda: ops.lstm_fwd(xproj, whh, lens, B, Tp, T)),
                     ("backward", lambda: ops.lstm_bwd(dh, whh, lens, gates, cst, B, Tp, T))):
        w, c = timed(fn, False), timed(fn, True)
        print(f"{name:9s} warm {w:7.0f} us = {w / T:5.2f} us per frame   cold {c:7.0f} us = {c / T:5.2f} us per frame")
    print("status", ops.lstm_status("cuda"), " lib", os.environ.get("APTAI_HIP_LIB", "(product)"))


if __name__ == "__main__":
    main()
