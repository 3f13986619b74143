#!/usr/bin/env python3
"""Development: how long the two hipGraphs of the pipelined Force_APTAI step take when they run side by side (events on each stream),
against each alone.  Run on the GPU box; APTAI_HIP_LIB selects the build."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from aptai_amd.graphed import GraphedForceStep
from aptai_amd.optim import Adam


def main():
    args = argparse.Namespace(model="base", batch=None, seconds=10.0, n_tv=12, no_regularisers=False, encoder_precision="bf16_f32res",
                              workload="force", no_pipeline=False)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    model, cfg = bench.build_force(args, dev)
    model.set_encoder_precision("bf16_f32res")
    B, S = 16, 160000
    batch = bench.synth_batch(cfg, B, S, 9, 0, dev, n_phn=40)
    batch["phoneme_labels"] = bench.synth_ctc_labels(B, 40, 0, dev)
    bench.calibrate_blank_bias(model, batch)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = Adam(params, lr=1e-5, betas=(0.9, 0.999), eps=1e-8)
    r = GraphedForceStep(model, opt, batch)
    for _ in range(5):
        r.step()
    torch.cuda.synchronize()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    # each graph alone
    es = r._enc_stream
    a0, a1 = ev(), ev()
    with torch.cuda.stream(es):
        a0.record(es)
        for _ in range(10):
            r.g_enc.replay()
        a1.record(es)
    torch.cuda.synchronize()
    b0, b1 = ev(), ev()
    b0.record()
    for _ in range(10):
        r.g_heads.replay()
    b1.record()
    torch.cuda.synchronize()
    print(f"alone: encoder graph {a0.elapsed_time(a1) / 10:.3f} ms, heads graph {b0.elapsed_time(b1) / 10:.3f} ms")
    # side by side, as step() issues them
    orig_enc, orig_heads = r.g_enc.replay, r.g_heads.replay
    rec = []

    class Wrap:
        def __init__(self, g, stream_fn, tag):
            self.g, self.stream_fn, self.tag = g, stream_fn, tag

        def replay(self):
            s = self.stream_fn()
            e0, e1 = ev(), ev()
            e0.record(s)
            self.g.replay()
            e1.record(s)
            rec.append((self.tag, e0, e1))

    r.g_enc = Wrap(r.g_enc, lambda: torch.cuda.current_stream(), "enc")
    r.g_heads = Wrap(r.g_heads, lambda: torch.cuda.current_stream(), "heads")
    t0, t1 = ev(), ev()
    t0.record()
    for _ in range(20):
        r.step()
    t1.record()
    torch.cuda.synchronize()
    enc = [a.elapsed_time(b) for t, a, b in rec if t == "enc"]
    heads = [a.elapsed_time(b) for t, a, b in rec if t == "heads"]
    print(f"side by side: step {t0.elapsed_time(t1) / 20:.3f} ms | encoder graph {sum(enc) / len(enc):.3f} ms | heads graph {sum(heads) / len(heads):.3f} ms")


if __name__ == "__main__":
    main()
