"""Event timeline of the pipelined Force_APTAI step: when the side-stream encoder pass and the main-stream heads (forward,
backward, optimiser) run relative to each other (rocprofv3 --kernel-trace serialises the queues, so it cannot show this)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import bench
from aptai_amd import force_aptai as fa
from aptai_amd.optim import Adam


def main():
    args = argparse.Namespace(model="base", seconds=10.0, batch=16, layers=None, encoder_precision="bf16", seed=0, no_regularisers=False, n_tv=9)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    model, cfg = bench.build_force(args, dev)
    B, S = 16, 160000
    batch = bench.synth_batch(cfg, B, S, 9, 0, dev, n_phn=40)
    batch["phoneme_labels"] = bench.synth_ctc_labels(B, 40, 0, dev)
    bench.calibrate_blank_bias(model, batch)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = Adam(params, lr=1e-5)
    marks = []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        marks.append((name, e))

    enc0 = model._encode

    def enc(*a, **k):
        mark("enc_begin"); r = enc0(*a, **k); mark("enc_end"); return r
    model._encode = enc
    apply0 = fa._ForceHeadsFn.apply
    nxt = (batch["audio_inputs"], batch["audio_lengths"])
    pipelined = os.environ.get("PIPE", "1") == "1"
    for i in range(8):
        if i == 5:
            marks.clear(); torch.cuda.synchronize()
        mark("step_begin")
        opt.zero_grad(set_to_none=True)
        out = model(0, **batch, _prefetch_next=nxt if pipelined else None)
        mark("fwd_end")
        out["loss"].backward()
        mark("bwd_end")
        opt.step()
        mark("opt_end")
    torch.cuda.synchronize()
    t0 = marks[0][1]
    for name, e in marks:
        print(f"{t0.elapsed_time(e):8.3f} ms  {name}")


if __name__ == "__main__":
    main()
