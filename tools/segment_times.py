#!/usr/bin/env python3
"""Device time of each hipGraph segment of the APTAI train step (aptai_amd.graphed): every segment replayed 20 times in a row
between two events.  Run on the GPU box: python tools/segment_times.py [--model large]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    large = "--model" in sys.argv and sys.argv[sys.argv.index("--model") + 1] == "large"
    args = types.SimpleNamespace(model="large" if large else "base", n_tv=12, no_regularisers=False)
    dev = torch.device("cuda", 0)
    model, cfg = bench.build_model(args, dev)
    model.train()
    from aptai_amd.optim import Adam
    from aptai_amd.graphed import GraphedAPTAIStep
    opt = Adam([p for p in model.parameters() if p.requires_grad], lr=1e-5).publish_to(model)
    B = 8 if large else 16
    batch = bench.synth_batch(cfg, B, 160000, 12, 0, dev)
    r = GraphedAPTAIStep(model, opt, batch)
    for _ in range(3):
        r.step()
    torch.cuda.synchronize()

    def t(fn, n=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    L = cfg.num_hidden_layers
    rows = [("prep (parameter casts / weight-norm)", t(r.g_prep.replay)), ("front (conv stack, projection, positional conv)", t(r.g_front.replay)),
            ("layer forward (mean of all)", sum(t(g.replay) for g in r.g_fwd) / L), ("tail (heads fwd + loss + heads bwd)", t(r.g_tail.replay)),
            ("layer backward (mean of all)", sum(t(g.replay) for g in r.g_bwd) / L), ("front backward", t(r.g_front_bwd.replay)),
            ("optimiser step (aptai_adam_multi, all layers kept)", t(opt.step))]
    keep = L * (1.0 - cfg.layerdrop)
    total = rows[0][1] + rows[1][1] + keep * rows[2][1] + rows[3][1] + keep * rows[4][1] + rows[5][1] + rows[6][1]
    for name, us in rows:
        print(f"{name:55s} {us:8.1f} us")
    print(f"{'sum at the expected ' + format(keep, '.1f') + ' kept layers':55s} {total:8.1f} us")


if __name__ == "__main__":
    main()
