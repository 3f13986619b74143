#!/usr/bin/env python3
"""bench.py — utterances/sec of one APTAI train step (10 s @ 16 kHz) on 1..8 MI355X.

Workload at N=1 (BASELINE.json configs[1]): models/aptai.py on a wav2vec2-base backbone with the 12-track
regression head + 46-class frame phoneme head, batch 16 x 10 s, bf16 compute, conv feature encoder frozen
(the reference default, models/aptai.py:24,39), dropout / LayerDrop / SpecAugment at the HF defaults.
A "step" = zero_grad -> forward -> backward -> Adam step over one synthetic batch already resident in HBM.
N>1: pure data parallel, 16 utterances per GPU (weak scaling), bucketed RCCL gradient all-reduce.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     — dominant kernel (bf16 MFMA GEMM, NT layout) timed live with HIP events on its launch stream
  cpu_baseline — the oracle (CPU restatement of the reference, oracle/) timed on this host's cores (N=1 only)
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_GF = {  # algorithmic GFLOP / utterance, forward (multiply-add = 2)
    ("base", 10.0): dict(conv=49.08, enc=99.05), ("large", 10.0): dict(conv=49.08, enc=334.76),
    ("base", 4.0): dict(conv=19.63, enc=37.30), ("large", 30.0): dict(conv=147.25, enc=1152.98),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="base", choices=["base", "large"])
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default 16 base / 8 large)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--n-tv", type=int, default=12)
    ap.add_argument("--no-regularisers", action="store_true", help="dropout/LayerDrop/SpecAugment off (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-steps", type=int, default=2)
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam(fused=True) instead of aptai_amd.optim.Adam")
    ap.add_argument("--host-batch", action="store_true", help="hand the step a pinned HOST batch every iteration (PCIe-inclusive rate; "
                    "never the headline value: DESIGN.md section 8)")
    ap.add_argument("--eager", action="store_true", help="drive the step through autograd (the drop-in loop) instead of hipGraphs")
    return ap.parse_args()


def build_model(args, device):
    from aptai_amd.aptai import APTAI
    from aptai_amd.config import W2V2Config
    from aptai_amd.wav2vec2 import Wav2Vec2Model
    kw = {}
    if args.no_regularisers:
        kw = dict(hidden_dropout=0., activation_dropout=0., attention_dropout=0., feat_proj_dropout=0., final_dropout=0.,
                  layerdrop=0., apply_spec_augment=False)
    cfg = W2V2Config.base(vocab_size=46, **kw) if args.model == "base" else W2V2Config.large(vocab_size=46, **kw)
    torch.manual_seed(0)                                   # identical random-init weights on every rank
    with tempfile.TemporaryDirectory() as tmp:
        Wav2Vec2Model(cfg).save_pretrained(tmp)
        vocab = {f"p{i}": i for i in range(46)}
        model = APTAI(device, vocab, tmp, cfg, None, n_tv=args.n_tv, n_phn=46)
    return model.to(device), cfg


def synth_batch(cfg, B, S, n_tv, rank, device):
    from aptai_amd import hostlogic
    g = torch.Generator().manual_seed(1234 + rank)
    lens = torch.full((B,), S, dtype=torch.long)
    for b in range(B // 2, B):
        lens[b] = int(torch.randint(int(0.8 * S), S + 1, (1,), generator=g))
    audio = torch.randn(B, S, generator=g) * (torch.arange(S)[None, :] < lens[:, None])
    T = hostlogic.feat_extract_output_lengths(S, cfg.conv_kernel, cfg.conv_stride)
    fl = hostlogic.feat_extract_output_lengths(lens, cfg.conv_kernel, cfg.conv_stride)
    valid = torch.arange(T)[None, :] < fl[:, None]
    batch = {"audio_inputs": audio, "audio_lengths": lens,
             "phn_frames_49hz": (torch.randint(1, 46, (B, T), generator=g) * valid).long()}
    names = list(hostlogic.TV_NAMES) + [f"XTV{i}" for i in range(max(0, n_tv - 9))]
    for n in names[:n_tv]:
        tv = torch.randn(B, T, generator=g, dtype=torch.float64)
        batch[n] = torch.where(valid, tv, torch.full_like(tv, -100.0))
    return {k: v.to(device) for k, v in batch.items()}


def host_cores() -> int:
    """Threads this process may really use: affinity mask, capped by the cgroup CPU quota and by the 16-core share
    of a one-GPU box."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(args, cfg_gpu):
    """Oracle train step (fp32, CPU): fwd + bwd + Adam on B=2 x the bench clip length, regularisers at defaults."""
    import copy
    from oracle import heads_ref, synth
    from aptai_amd import hostlogic
    cores = host_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline on {cores} host threads ...", file=sys.stderr, flush=True)
    cfg = copy.deepcopy(cfg_gpu)
    shapes = synth.aptai_param_shapes(cfg, n_tv=9, n_phn=46)
    sd = synth.make_state_dict(shapes, 0)
    train = [v.requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32 and "feature_extractor" not in k]
    opt = torch.optim.Adam(train, lr=1e-5)
    S = int(16000 * args.seconds)
    batch = synth.synth_aptai_batch(cfg, 2, S, seed=1234)
    tv = [batch[n] for n in hostlogic.TV_NAMES]
    times = []
    for it in range(1 + args.cpu_baseline_steps):
        t0 = time.perf_counter()
        opt.zero_grad()
        keep = [bool(torch.rand([]) >= cfg.layerdrop) for _ in range(cfg.num_hidden_layers)]
        T = int(hostlogic.feat_extract_output_lengths(S, cfg.conv_kernel, cfg.conv_stride))
        fl = hostlogic.feat_extract_output_lengths(batch["audio_lengths"], cfg.conv_kernel, cfg.conv_stride)
        am = torch.arange(T)[None] < fl[:, None]
        mask = None
        if cfg.apply_spec_augment and cfg.mask_time_prob > 0:
            mask = torch.from_numpy(hostlogic.compute_mask_indices((2, T), cfg.mask_time_prob, cfg.mask_time_length,
                                                                   attention_mask=am, min_masks=cfg.mask_time_min_masks))
        out = heads_ref.aptai_forward(sd, cfg, batch["audio_inputs"], batch["audio_lengths"], batch["phn_frames_49hz"], tv,
                                      training=True, mask_time_indices=mask, layer_keep=keep)
        out["loss"].backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu_baseline step {it}: {times[-1]:.2f} s", file=sys.stderr, flush=True)
    t = sum(times[1:]) / len(times[1:])
    return {"value": round(2 / t, 4), "unit": "utterances/sec", "cores": cores, "kind": "port",
            "sample": f"oracle APTAI train step (fp32 torch CPU restatement), wav2vec2-{args.model}, B=2 x {args.seconds:g} s, "
                      f"1 warm-up + {args.cpu_baseline_steps} timed steps, {t:.2f} s/step"}


def ema_rmse_check(model, cfg, args, device):
    """BASELINE metric's second half, "EMA RMSE vs ref": the same weights and one B=2 batch through the build (MI355X,
    bf16, eval mode) and through the oracle (CPU fp32); tvs_metric_rmse (utility.py:393-418, mean over tracks) of each
    against the synthetic targets, their difference, and the RMSE between the two predictions."""
    import numpy as np
    from oracle import heads_ref
    from aptai_amd import hostlogic, metrics
    S = int(16000 * args.seconds)
    batch = synth_batch(cfg, 2, S, args.n_tv, 7, device)
    names = [k for k in batch if k not in ("audio_inputs", "audio_lengths", "phn_frames_49hz")]
    was_training = model.training
    model.eval()
    with torch.no_grad():
        out = model(0, **batch)
    pred_gpu = out["tvs_pred"].float().cpu().numpy()
    model.train(was_training)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cb = {k: v.cpu() for k, v in batch.items()}
    with torch.no_grad():
        ref = heads_ref.aptai_forward(sd, cfg, cb["audio_inputs"], cb["audio_lengths"], cb["phn_frames_49hz"],
                                      [cb[n] for n in names], training=False)
    pred_ref = ref["tvs_pred"].numpy()
    tgt = torch.stack([cb[n] for n in names], dim=-1).numpy()
    valid = tgt[..., 0] != -100.0
    g, a, b = tgt[valid], pred_gpu[valid], pred_ref[valid]
    r_build = metrics.ema_rmse(g, a, names)
    r_ref = metrics.ema_rmse(g, b, names)
    return {"build": round(r_build, 6), "oracle": round(r_ref, 6), "abs_diff": float(f"{abs(r_build - r_ref):.3e}"),
            "rmse_between_predictions": float(f"{float(np.sqrt(np.mean((a - b) ** 2))):.3e}"),
            "sample": f"eval forward, B=2 x {args.seconds:g} s, {len(names)} tracks, {int(valid.sum())} valid frames, same weights; "
                      "tvs_metric_rmse averaged over tracks (utility.py:393-418)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    # one rank per GPU; APTAI_BENCH_BACKEND=gloo rehearses the multi-rank plumbing on a box with fewer GPUs than ranks
    backend = os.environ.get("APTAI_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from aptai_amd import ops
    from aptai_amd.dp import GradBucketReducer
    B = args.batch or (16 if args.model == "base" else 8)
    S = int(16000 * args.seconds)
    model, cfg = build_model(args, device)
    model.wav2vec2.base_seed += rank
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    if args.torch_adam:
        opt = torch.optim.Adam(params, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, fused=True)
    else:                                                    # same update rule as one multi-tensor HIP kernel (csrc/optim.hip)
        from aptai_amd.optim import Adam
        opt = Adam(params, lr=1e-5, betas=(0.9, 0.999), eps=1e-8).publish_to(model)
    reducer = GradBucketReducer(params, bucket_mb=48.0, comm_dtype=torch.bfloat16) if world > 1 else None
    if world > 1:                            # masked-mean losses over the GLOBAL batch: DP gradients == single-process gradients
        from aptai_amd.dp import GlobalLossNorm
        model.dp_loss_norm = GlobalLossNorm()
    batch = synth_batch(cfg, B, S, args.n_tv, rank, device)

    runner = None
    if not args.eager:
        from aptai_amd.graphed import GraphedAPTAIStep
        try:
            if reducer is not None:
                reducer.remove()             # gradients are reduced explicitly after the captured backward
            if os.environ.get("APTAI_BENCH_FAIL_CAPTURE"):      # rehearsal of the fallback below
                raise RuntimeError("APTAI_BENCH_FAIL_CAPTURE is set")
            runner = GraphedAPTAIStep(model, opt, batch, reducer=reducer)
            step = runner.step
            if args.host_batch:              # the collate_fn's view of the boundary: host tensors in, H2D inside the step
                host = {k: v.cpu().pin_memory() for k, v in batch.items()}
                step = lambda: runner.step(host)
        except Exception as e:               # noqa: BLE001 - multi-rank only: same kernels through the eager loop, and say so
            if world == 1:
                raise
            print(f"[bench] rank {rank}: hipGraph capture failed ({e!r}); running the eager loop", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            runner = None
            args.eager = True
            from aptai_amd import _lib                   # what GraphedAPTAIStep.close() would have reset
            _lib.call("aptai_set_seed_salt", None)
            model.wav2vec2._cache_mode = None
            model.wav2vec2._cache.clear()
            reducer = GradBucketReducer(params, bucket_mb=48.0, comm_dtype=torch.bfloat16)
    if args.eager:
        def step():
            opt.zero_grad(set_to_none=True)
            out = model(0, **batch)
            out["loss"].backward()
            if reducer is not None:
                reducer.finish()
            opt.step()
            return out

    for _ in range(args.warmup):
        step()
    print(f"[bench] rank {rank}: warm-up done", file=sys.stderr, flush=True)
    probe = ops.GemmProbe(False, False, False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.set_gemm_probe(probe)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ops.set_gemm_probe(None)
    probe_note = "HIP events around every launch inside the timed region"
    if not args.eager:
        # kernels inside a hipGraph replay cannot be bracketed one by one: re-issue the SAME step eagerly right after the
        # timed region (same model, batch, shapes, regularisers) with the event probe on
        runner.close()
        probe = ops.GemmProbe(False, False, False)
        ops.set_gemm_probe(probe)
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            model(0, **batch)["loss"].backward()
        torch.cuda.synchronize()
        ops.set_gemm_probe(None)
        probe_note = "HIP events around every launch of 3 eager re-runs of the same step right after the timed region"
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(out["loss"].detach())
    if rank == 0:
        print(f"[bench] timed region: {dt / args.steps * 1e3:.2f} ms/step", file=sys.stderr, flush=True)
        ps = probe.summary()
        gemm_tflops = ps["flops"] / (ps["ms"] * 1e-3) / 1e12 if ps["ms"] > 0 else 0.0
        traffic = None
        try:                 # PMC traffic of the dominant kernel, collected offline (tools/pmc_summary.py), bytes per launch
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
            nt = [v for k, v in pm.items() if "<false, false, false>" in k and "gemm" in k]      # the NT family the probe brackets
            traffic = int(sum(v["hbm_bytes_per_launch_corrected"] * v["launches_in_trace"] for v in nt) /
                          max(sum(v["launches_in_trace"] for v in nt), 1))
        except Exception:
            pass
        gf = FWD_GF.get((args.model, args.seconds))
        step_tf = None
        if gf:
            step_tf = (gf["conv"] + 3 * gf["enc"]) * B / 1e3          # conv stack frozen: fwd only; encoder fwd+bwd
        res = {
            "metric": "utterances/sec (10 s @ 16 kHz) train-step", "value": round(world * B * args.steps / dt, 3),
            "unit": "utterances/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic N(0,1) 16 kHz waveforms, random-init weights",
            "config": {"workload": f"APTAI train step (models/aptai.py): wav2vec2-{args.model} + {args.n_tv}-dim EMA regression "
                                   f"head + 46-class frame phoneme head, conv feature encoder frozen",
                       "per_gpu_batch": B, "global_batch": world * B, "clip_seconds": args.seconds,
                       "frames_per_clip": int(S // 320 - (1 if S % 320 < 80 else 0)) if False else None,
                       "parallelism": f"dp{world}", "regularisers": "off" if args.no_regularisers else "HF defaults",
                       "optimizer": "Adam, fp32 state (torch fused)" if args.torch_adam else "Adam, fp32 state (aptai_adam_multi, refreshes the bf16 weight copies)",
                       "execution": "eager autograd loop" if args.eager else "hipGraph segments (aptai_amd.graphed)",
                       "inputs": "pinned host batch copied in every step (PCIe-inclusive)" if args.host_batch else "resident in HBM"},
            "loss": round(loss, 5),
            "roofline": {"bound": "mfma", "kernel": "bf16 MFMA GEMM, NT layout (gemm_kernel / gemm192_kernel / gemm256_kernel <false,false,false>): every launch of the step",
                         "achieved": round(gemm_tflops, 2), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(gemm_tflops / 2500.0, 4), "traffic": traffic,
                         "traffic_note": "HBM-side bytes per launch, launch-weighted over the same NT kernels, rocprofv3 PMC (2 x FETCH_SIZE + "
                                         "WRITE_SIZE, separate passes), profiles/r01_pmc_traffic.json",
                         "measured": probe_note, "launches": ps["launches"], "avg_launch_us": round(ps["ms"] * 1e3 / max(ps["launches"], 1), 2),
                         "step_algorithmic_tflop": step_tf,
                         "whole_step_frac_of_peak": round(step_tf / (dt / args.steps) / 2500.0, 4) if step_tf else None},
        }
        res["config"].pop("frames_per_clip")
        if world == 1 and not args.no_cpu_baseline:
            res["ema_rmse_vs_ref"] = ema_rmse_check(model, cfg, args, device)
            res["cpu_baseline"] = cpu_baseline(args, cfg)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
