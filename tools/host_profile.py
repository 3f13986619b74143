"""cProfile of the HOST side of one workload's eager step (where the Python / launch time goes)."""
import argparse
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aptai_amd.optim import Adam


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "force"
    args = argparse.Namespace(model="base", seconds=10.0, batch=16, layers=None, encoder_precision="bf16", seed=0, no_regularisers=False, n_tv=9)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    B, S = 16, 160000
    if wl == "force":
        model, cfg = bench.build_force(args, dev)
        batch = bench.synth_batch(cfg, B, S, 9, 0, dev, n_phn=40)
        batch["phoneme_labels"] = bench.synth_ctc_labels(B, 40, 0, dev)
        bench.calibrate_blank_bias(model, batch)
        nxt = (batch["audio_inputs"], batch["audio_lengths"])
        call = lambda: model(0, **batch, _prefetch_next=nxt)
    else:
        model, cfg = bench.build_aptai(args, dev)
        batch = bench.synth_batch(cfg, B, S, 9, 0, dev)
        call = lambda: model(0, **batch)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = Adam(params, lr=1e-5)
    if wl != "force":
        opt = opt.publish_to(model)

    def step():
        opt.zero_grad(set_to_none=True)
        out = call()
        out["loss"].backward()
        opt.step()

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(30)


if __name__ == "__main__":
    main()
