#!/usr/bin/env python3
"""Host time of one graphed APTAI train step (aptai_amd.graphed.GraphedAPTAIStep.step) by Python function: cProfile over
replays that are NOT separated by device syncs, so the numbers are launch costs, not kernel time.  Run on the GPU box."""
import cProfile, os, pstats, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    args = types.SimpleNamespace(model="base", n_tv=12, no_regularisers=False)
    dev = torch.device("cuda", 0)
    model, cfg = bench.build_model(args, dev)
    model.train()
    from aptai_amd.optim import Adam
    from aptai_amd.graphed import GraphedAPTAIStep
    opt = Adam([p for p in model.parameters() if p.requires_grad], lr=1e-5).publish_to(model)
    batch = bench.synth_batch(cfg, 16, 160000, 12, 0, dev)
    r = GraphedAPTAIStep(model, opt, batch)
    for _ in range(5):
        r.step()
    torch.cuda.synchronize()
    n = 30
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(n):
        r.step()
    pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host {1e3 * (t1 - t0) / n:.2f} ms/step (under cProfile), device-complete {1e3 * (t2 - t0) / n:.2f} ms/step")
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)


if __name__ == "__main__":
    main()
