#!/usr/bin/env python3
"""bench.py — utterances/sec of one train step of APTAI's hot path (10 s @ 16 kHz clips) on 1..8 MI355X.

Workloads (``--workload``; a "step" = zero_grad -> forward -> backward -> Adam step over one synthetic batch resident in HBM):
  aptai (default, BASELINE.json configs[1]): models/aptai.py on wav2vec2-base, 12-track regression head + 46-class frame
        phoneme head, 16 x 10 s per GPU, bf16, conv feature encoder frozen (models/aptai.py:24,39), regularisers at HF defaults.
  force (configs[2]): models/force_aptai.py on wav2vec2-base + 40-phoneme CTC recogniser: frozen encoder in inference mode
        (models/w2v2_pr.py:124-127), best-path decode, cross-attention aligner + forward-sum loss + BiLSTM regression; only
        the heads train.
  pr    (the fine-tuning loop behind configs[0], at 16 x 10 s): models/w2v2_pr.py, everything trainable incl. the conv stack,
        CTC loss (train/train_phoneme_recognizer.py:384-486).
N > 1: pure data parallel, one rank per GPU (weak scaling), gradient buckets averaged over RCCL.  ``python bench.py --gpus N``
starts the N ranks itself (torch.distributed.run) when it was not already started by a launcher (WORLD_SIZE unset).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     — dominant kernel family (bf16 MFMA GEMM, NT layout) timed live with HIP events on its launch stream
  cpu_baseline — the oracle (CPU restatement of the reference, oracle/) timed on this host's cores (N=1 only)
"""
import argparse
import json
import os
import socket
import subprocess
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
PROFILE_TAG = "r04"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="aptai", choices=["aptai", "force", "pr"])
    ap.add_argument("--no-pipeline", action="store_true",
                    help="force workload: run the frozen encoder inline instead of one batch ahead on a side stream")
    ap.add_argument("--model", default="base", choices=["base", "large"])
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default 16 base / 8 large)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--n-tv", type=int, default=12)
    ap.add_argument("--no-regularisers", action="store_true", help="dropout/LayerDrop/SpecAugment off (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-steps", type=int, default=2)
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam(fused=True) instead of aptai_amd.optim.Adam")
    ap.add_argument("--host-batch", action="store_true", help="hand the step a pinned HOST batch every iteration (PCIe-inclusive rate; "
                    "never the headline value: DESIGN.md section 11)")
    ap.add_argument("--encoder-precision", default="bf16_f32res", choices=["bf16_f32res", "bf16", "mxfp8", "f32x3", "f32x6"],
                    help="force workload: precision of the frozen encoder (mxfp8 = BASELINE configs[4]; bf16_f32res = bf16 GEMMs with an fp32 residual stream)")
    ap.add_argument("--plan-gpus", type=int, default=8, help="world size the JSON line's collective_plan is written for when the run itself is "
                                                             "one rank (dry plan of the gradient exchange; multi-rank runs describe their own)")
    ap.add_argument("--no-exact-line", action="store_true", help="force workload: skip the second measurement in the index-exact encoder mode")
    ap.add_argument("--eager", action="store_true", help="drive the step through autograd (the drop-in loop) instead of hipGraphs")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------- rank launcher
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpu_count() -> int:
    """GPUs this process may use, counted WITHOUT opening the HIP runtime (the parent of the ranks must never hold the devices):
    KFD topology nodes with SIMDs (/sys/class/kfd/kfd/topology/nodes/*/properties; CPU nodes have simd_count 0), narrowed by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set.  Returns -1 if the topology is unreadable (the
    ranks' own `torch.cuda.set_device` then reports a missing device)."""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return -1
    for f in nodes:
        try:
            with open(f) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
        except OSError:
            return -1
        if int(props.get("simd_count", 0)) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks_if_needed(args) -> None:
    """`python bench.py --gpus N` without a launcher: start N ranks of this file under torch.distributed.run and exit with
    their status.  Runs BEFORE anything touches the GPU in this process (a process that initialised HIP must never be
    replaced or forked into ranks); rank 0 of the children prints the JSON line on the inherited stdout."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world_env}")
        return
    if args.gpus <= 1:
        return
    backend = os.environ.get("APTAI_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        n_dev = visible_gpu_count()                          # sysfs only: nothing in the parent opens the HIP runtime
        if 0 <= n_dev < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node exposes {n_dev} GPU(s); refusing to measure fewer "
                             f"ranks than asked (APTAI_BENCH_BACKEND=gloo rehearses the multi-rank path on one card)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    rc = subprocess.call(cmd, env=env)
    raise SystemExit(rc)


# ------------------------------------------------------------------------------------------------- models and batches
def _cfg(args, vocab_size, **extra):
    from aptai_amd.config import W2V2Config
    kw = dict(extra)
    if args.no_regularisers:
        kw.update(hidden_dropout=0., activation_dropout=0., attention_dropout=0., feat_proj_dropout=0., final_dropout=0.,
                  layerdrop=0., apply_spec_augment=False)
    return W2V2Config.base(vocab_size=vocab_size, **kw) if args.model == "base" else W2V2Config.large(vocab_size=vocab_size, **kw)


def build_aptai(args, device):
    from aptai_amd.aptai import APTAI
    from aptai_amd.wav2vec2 import Wav2Vec2Model
    cfg = _cfg(args, 46)
    torch.manual_seed(0)                                   # identical random-init weights on every rank
    with tempfile.TemporaryDirectory() as tmp:
        Wav2Vec2Model(cfg).save_pretrained(tmp)
        vocab = {f"p{i}": i for i in range(46)}
        model = APTAI(device, vocab, tmp, cfg, None, n_tv=args.n_tv, n_phn=46)
    return model.to(device), cfg


def _vocab40():
    vocab = {"(blank)": 0, "(...)": 1}
    vocab.update({f"p{i}": i for i in range(2, 40)})
    return vocab


def build_pr(args, device):
    from aptai_amd.w2v2_pr import Wav2Vec2_PR
    from aptai_amd.wav2vec2 import Wav2Vec2Model
    cfg = _cfg(args, 40, ctc_loss_reduction="mean", ctc_zero_infinity=True)      # train/train_phoneme_recognizer.py:339-342
    torch.manual_seed(0)
    with tempfile.TemporaryDirectory() as tmp:
        Wav2Vec2Model(cfg).save_pretrained(tmp)
        model = Wav2Vec2_PR(cfg, None, tmp, _vocab40())
    return model.to(device), cfg


def build_force(args, device):
    """Force_APTAI over a random-init Wav2Vec2_PR checkpoint written the way train_phoneme_recognizer.py writes it
    (best-model-ckpt/{pytorch_model.bin, model_cfg.pkl}, models/force_aptai.py:60-75)."""
    import pickle
    from aptai_amd.force_aptai import Force_APTAI
    from aptai_amd.w2v2_pr import Wav2Vec2_PR
    from aptai_amd.wav2vec2 import Wav2Vec2Model
    cfg = _cfg(args, 40, ctc_loss_reduction="mean", ctc_zero_infinity=True)
    vocab = _vocab40()
    torch.manual_seed(0)
    with tempfile.TemporaryDirectory() as tmp:
        mdir = os.path.join(tmp, "w2v2")
        Wav2Vec2Model(cfg).save_pretrained(mdir)
        pr = Wav2Vec2_PR(cfg, None, mdir, vocab)
        ck = os.path.join(tmp, "pr", "best-model-ckpt")
        os.makedirs(ck)
        torch.save(pr.state_dict(), os.path.join(ck, "pytorch_model.bin"))
        with open(os.path.join(ck, "model_cfg.pkl"), "wb") as f:
            pickle.dump({"pretrain_cfg": cfg.to_dict(), "cache_dir": None, "huggingface_model_id": mdir}, f)
        model = Force_APTAI(os.path.join(tmp, "pr"), device, vocab)
    return model.to(device), cfg


def synth_batch(cfg, B, S, n_tv, rank, device, n_phn=46):
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
             "phn_frames_49hz": (torch.randint(1, n_phn, (B, T), generator=g) * valid).long()}
    names = list(hostlogic.TV_NAMES) + [f"XTV{i}" for i in range(max(0, n_tv - 9))]
    for n in names[:n_tv]:
        tv = torch.randn(B, T, generator=g, dtype=torch.float64)
        batch[n] = torch.where(valid, tv, torch.full_like(tv, -100.0))
    return {k: v.to(device) for k, v in batch.items()}


def synth_ctc_labels(B, vocab, rank, device, lo=20, hi=55):
    """CTC label rows: lengths uniform in [20, 55] (< 60, models/force_aptai.py:111), ids in [1, V-1], -100 padding."""
    g = torch.Generator().manual_seed(4321 + rank)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    lab = torch.full((B, int(lens.max())), -100, dtype=torch.int32)
    for b in range(B):
        lab[b, :lens[b]] = torch.randint(1, vocab, (int(lens[b]),), generator=g, dtype=torch.int32)
    return lab.to(device)


def calibrate_blank_bias(model, batch, lo=20, hi=55):
    """A trained recogniser emits blank on most frames; random-init weights emit ~T distinct labels per clip, more than the
    aligner's 60 phoneme slots (models/force_aptai.py:30,111).  Raise the blank logit's bias until every utterance of the
    synthetic batch decodes to a list of lo..hi phonemes (the length range of SURVEY.md 8d's CTC labels).  Setup only."""
    import numpy as np
    pr = model.w2v2_pr
    with torch.no_grad():
        _, logits = pr._logits_eval(batch["audio_inputs"], batch["audio_lengths"][:, None])
        lg = logits.float().cpu().numpy()
    best_other = lg[..., 1:].max(-1)
    margin = np.sort((best_other - lg[..., 0]).reshape(-1))             # blank wins a frame iff bias > margin
    bias, got = None, None
    for q in np.linspace(0.80, 0.995, 40):
        b = float(margin[int(q * (len(margin) - 1))])
        l2 = lg.copy()
        l2[..., 0] += b
        ids = l2.argmax(-1)
        ns = []
        for row in ids:
            keep = np.ones(len(row), dtype=bool)
            keep[1:] = row[1:] != row[:-1]
            r = row[keep]
            ns.append(int((r != 0).sum()))
        if max(ns) <= hi and (bias is None or min(ns) >= lo):
            bias, got = b, ns
            if min(ns) >= lo:
                break
    if bias is None:
        raise SystemExit("bench.py: could not calibrate the blank bias of the synthetic recogniser")
    with torch.no_grad():
        pr.pr_head.bias[0] += bias
    return bias, got


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


# ------------------------------------------------------------------------------------------------- CPU baseline (oracle)
def cpu_baseline(args, cfg_gpu):
    """Oracle train step of the SAME workload (fp32, CPU) on B=2 x the bench clip length, regularisers at defaults."""
    import copy
    from oracle import heads_ref, synth
    from aptai_amd import hostlogic
    cores = host_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline ({args.workload}) on {cores} host threads ...", file=sys.stderr, flush=True)
    cfg = copy.deepcopy(cfg_gpu)
    S = int(16000 * args.seconds)
    T = int(hostlogic.feat_extract_output_lengths(S, cfg.conv_kernel, cfg.conv_stride))
    if args.workload == "aptai":
        sd = synth.make_state_dict(synth.aptai_param_shapes(cfg, n_tv=9, n_phn=46), 0)
        train = [v.requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32 and "feature_extractor" not in k]
        batch = synth.synth_aptai_batch(cfg, 2, S, seed=1234)
        tv = [batch[n] for n in hostlogic.TV_NAMES]
    elif args.workload == "pr":
        sd = synth.make_state_dict(synth.pr_param_shapes(cfg), 0)
        train = [v.requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32]
        batch = synth.synth_pr_batch(cfg, 2, S, seed=1234)
    else:
        sd = synth.make_state_dict(synth.force_aptai_param_shapes(cfg, 40), 0)
        train = [v.requires_grad_(True) for k, v in sd.items()
                 if v.dtype == torch.float32 and not k.startswith("w2v2_pr.") and k != "pe_phn.pe"]
        batch = synth.synth_aptai_batch(cfg, 2, S, seed=1234, n_phn=40)
        tv = [batch[n] for n in hostlogic.TV_NAMES]
        g = torch.Generator().manual_seed(5)
        lists = [torch.randint(2, 40, (int(torch.randint(20, 56, (1,), generator=g)),), generator=g).numpy() for _ in range(2)]
    opt = torch.optim.Adam(train, lr=1e-5)
    times = []
    for it in range(1 + args.cpu_baseline_steps):
        t0 = time.perf_counter()
        opt.zero_grad()
        keep = [bool(torch.rand([]) >= cfg.layerdrop) for _ in range(cfg.num_hidden_layers)]
        mask = None
        if args.workload != "force" and cfg.apply_spec_augment and cfg.mask_time_prob > 0:
            lens = batch["audio_lengths"] if args.workload == "aptai" else batch["input_lengths"]
            fl = hostlogic.feat_extract_output_lengths(lens, cfg.conv_kernel, cfg.conv_stride)
            am = torch.arange(T)[None] < fl[:, None]
            mask = torch.from_numpy(hostlogic.compute_mask_indices((2, T), cfg.mask_time_prob, cfg.mask_time_length,
                                                                   attention_mask=am, min_masks=cfg.mask_time_min_masks))
        if args.workload == "aptai":
            out = heads_ref.aptai_forward(sd, cfg, batch["audio_inputs"], batch["audio_lengths"], batch["phn_frames_49hz"], tv,
                                          training=True, mask_time_indices=mask, layer_keep=keep)
        elif args.workload == "pr":
            out = heads_ref.pr_forward(sd, cfg, batch["input_values"], batch["input_lengths"], batch["phoneme_labels"],
                                       training=True, mask_time_indices=mask, layer_keep=keep)
        else:                       # encoder in inference mode (models/w2v2_pr.py:124-127), heads in training mode
            out = heads_ref.force_aptai_forward(sd, cfg, batch["audio_inputs"], batch["audio_lengths"], tv, phn_pred_list=lists,
                                                training=True)
        out["loss"].backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu_baseline step {it}: {times[-1]:.2f} s", file=sys.stderr, flush=True)
    t = sum(times[1:]) / len(times[1:])
    what = {"aptai": "APTAI", "pr": "Wav2Vec2_PR", "force": "Force_APTAI"}[args.workload]
    return {"value": round(2 / t, 4), "unit": "utterances/sec", "cores": cores, "kind": "port",
            "sample": f"oracle {what} train step (fp32 torch CPU restatement), wav2vec2-{args.model}, B=2 x {args.seconds:g} s, "
                      f"1 warm-up + {args.cpu_baseline_steps} timed steps, {t:.2f} s/step"}


def ema_rmse_check(model, cfg, args, device):
    """BASELINE metric's second half, "EMA RMSE vs ref": the same weights and one B=2 batch through the build (MI355X,
    bf16, eval mode) and through the oracle (CPU fp32); tvs_metric_rmse (utility.py:393-418, mean over tracks) of each
    against the synthetic targets, their difference, and the RMSE between the two predictions."""
    import numpy as np
    from oracle import heads_ref
    from aptai_amd import metrics
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
                      "tvs_metric_rmse averaged over tracks (utility.py:393-418); with random weights and N(0,1) targets both RMSEs "
                      "are ~1, so abs_diff is insensitive - rmse_between_predictions is the honest distance"}


# ------------------------------------------------------------------------------------------------- main
def main():
    args = parse()
    launch_ranks_if_needed(args)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
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
    from aptai_amd.dp import GlobalLossNorm, GradBucketReducer, default_algo
    B = args.batch or (16 if args.model == "base" else 8)
    S = int(16000 * args.seconds)
    wl = args.workload
    data_note = "synthetic N(0,1) 16 kHz waveforms, random-init weights"
    if wl == "aptai":
        model, cfg = build_aptai(args, device)
        w2v2 = model.wav2vec2
        batch = synth_batch(cfg, B, S, args.n_tv, rank, device)
        call = lambda: model(0, **batch)
    elif wl == "pr":
        model, cfg = build_pr(args, device)
        w2v2 = model.wav2vec2
        sb = synth_batch(cfg, B, S, 9, rank, device, n_phn=40)
        batch = {"input_values": sb["audio_inputs"], "input_lengths": sb["audio_lengths"],
                 "phoneme_labels": synth_ctc_labels(B, 40, rank, device)}
        call = lambda: model(**batch)
    else:
        model, cfg = build_force(args, device)
        model.set_encoder_precision(args.encoder_precision)
        w2v2 = model.w2v2_pr.wav2vec2
        batch = synth_batch(cfg, B, S, 9, rank, device, n_phn=40)
        batch["phoneme_labels"] = synth_ctc_labels(B, 40, rank, device)
        bias, counts = calibrate_blank_bias(model, batch)
        data_note += (f"; blank bias of the random-init recogniser raised by {bias:.3f} so that the best-path decode inside the "
                      f"step yields {min(counts)}..{max(counts)} phonemes per clip (a trained recogniser's regime)")
        call = lambda: model(0, **batch)
        if not args.no_pipeline:
            # every step: heads of this batch + the frozen-encoder pass of the next one on a side stream (Force_APTAI.prefetch);
            # the first call encodes inline, so each timed step still holds exactly one encoder pass and one heads pass
            nxt = (batch["audio_inputs"], batch["audio_lengths"])
            call = lambda: model(0, **batch, _prefetch_next=nxt)
    w2v2.base_seed += rank
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    if args.torch_adam:
        opt = torch.optim.Adam(params, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, fused=True)
    else:                                                    # same update rule as one multi-tensor HIP kernel (csrc/optim.hip)
        from aptai_amd.optim import Adam
        opt = Adam(params, lr=1e-5, betas=(0.9, 0.999), eps=1e-8)
        if wl != "force":                                    # force: the encoder is frozen, nothing to publish
            opt = opt.publish_to(model)
    reducer = GradBucketReducer(params, bucket_mb=48.0, comm_dtype=torch.bfloat16) if world > 1 else None
    if world > 1 and wl != "pr":             # masked-mean losses over the GLOBAL batch: DP gradients == single-process gradients
        model.dp_loss_norm = GlobalLossNorm()

    runner = None
    use_graph = (not args.eager) and (wl in ("aptai", "pr") or (wl == "force" and world == 1 and not args.no_pipeline))
    if use_graph:
        from aptai_amd.graphed import GraphedAPTAIStep, GraphedForceStep
        capture_error = None
        try:
            if reducer is not None:
                reducer.remove()             # gradients are reduced explicitly after the captured backward
            if wl == "force":                # encoder graph (side stream, one batch ahead) + heads graph
                runner = GraphedForceStep(model, opt, batch)
            else:
                runner = GraphedAPTAIStep(model, opt, batch, reducer=reducer)
        except Exception as e:               # noqa: BLE001 - multi-rank only: same kernels through the eager loop, and say so
            if world == 1:
                raise
            capture_error = e
            torch.cuda.synchronize()
        if world > 1:
            # the ranks must take the SAME path: the graph runner reduces per layer group, the eager loop per 48 MB bucket -
            # different collective sequences on different ranks would hang or corrupt the gradients
            ok = torch.tensor([0 if capture_error is not None else 1], device=device if backend == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if capture_error is not None:
                    print(f"[bench] rank {rank}: hipGraph capture failed ({capture_error!r})", file=sys.stderr, flush=True)
                if rank == 0:
                    print("[bench] at least one rank could not capture: ALL ranks run the eager loop", file=sys.stderr, flush=True)
                if runner is not None:
                    runner.close()
                runner = None
                use_graph = False
                w2v2._cache_mode = None
                w2v2._cache.clear()
                reducer = GradBucketReducer(params, bucket_mb=48.0, comm_dtype=torch.bfloat16)
    if use_graph:
        step = runner.step
        if args.host_batch:                  # the collate_fn's view of the boundary: host tensors in, H2D inside the step
            host = {k: v.cpu().pin_memory() for k, v in batch.items()}
            step = lambda: runner.step(host)
    else:
        def step():
            opt.zero_grad(set_to_none=True)
            out = call()
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
    if use_graph:
        # kernels inside a hipGraph replay cannot be bracketed one by one: re-issue the SAME step eagerly right after the
        # timed region (same model, batch, shapes, regularisers) with the event probe on
        runner.close()
        # The probe's event pairs time a launch correctly only while the queue is never empty (an idle queue stamps the start
        # event at once and the kernel arrives a host call later): ~30 ms of unrelated launches first, so that the host stays
        # ahead of the GPU for the three host-bound eager steps that follow.
        plug = [torch.zeros(8192, 8192, device=device, dtype=torch.bfloat16) for _ in range(3)]
        torch.cuda.synchronize()
        for _ in range(24):
            ops.gemm(plug[0], plug[1], 8192, 8192, 8192, out=plug[2])
        probe = ops.GemmProbe(False, False, False)
        ops.set_gemm_probe(probe)
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            call()["loss"].backward()
        torch.cuda.synchronize()
        ops.set_gemm_probe(None)
        del plug
        probe_note = "HIP events around every launch of 3 eager re-runs of the same step right after the timed region"
    if world > 1:
        t = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(out["loss"].detach())
    if rank == 0:
        print(f"[bench] timed region: {dt / args.steps * 1e3:.2f} ms/step", file=sys.stderr, flush=True)
        ps = probe.summary()
        gemm_tflops = ps["flops"] / (ps["ms"] * 1e-3) / 1e12 if ps["ms"] > 0 else 0.0
        traffic, traffic_src = None, None
        for tag in (PROFILE_TAG, "r01"):   # PMC traffic of the dominant kernel, collected offline (tools/pmc_summary.py)
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")))["kernels"]
                nt = [v for k, v in pm.items() if "<false, false, false>" in k and "gemm" in k]      # the NT family the probe brackets
                traffic = int(sum(v["hbm_bytes_per_launch_corrected"] * v["launches_in_trace"] for v in nt) /
                              max(sum(v["launches_in_trace"] for v in nt), 1))
                traffic_src = f"profiles/{tag}_pmc_traffic.json"
                break
            except Exception:
                continue
        if wl != "aptai":
            traffic, traffic_src = None, None            # the PMC passes were collected on the aptai workload
        gf = FWD_GF.get((args.model, args.seconds))
        step_tf = step_tf_exec = None
        if gf:
            if wl == "aptai":          # conv stack frozen: fwd only; encoder fwd+bwd
                step_tf = (gf["conv"] + 3 * gf["enc"]) * B / 1e3
                # LayerDrop skips a layer's forward AND backward with probability p: the executed encoder work is (1 - p) of
                # the layer share; `step_algorithmic_tflop` counts every layer (the reference's nominal work per step)
                keep = 1.0 - (0.0 if args.no_regularisers else cfg.layerdrop)
                step_tf_exec = (gf["conv"] + 3 * gf["enc"] * keep) * B / 1e3
            elif wl == "pr":           # everything trainable: 3 x forward
                step_tf = 3 * (gf["conv"] + gf["enc"]) * B / 1e3
                keep = 1.0 - (0.0 if args.no_regularisers else cfg.layerdrop)
                step_tf_exec = 3 * (gf["conv"] + gf["enc"] * keep) * B / 1e3
            else:                      # encoder forward only (inference); heads < 0.1 %
                step_tf = step_tf_exec = (gf["conv"] + gf["enc"]) * B / 1e3
        what = {"aptai": f"APTAI train step (models/aptai.py): wav2vec2-{args.model} + {args.n_tv}-dim EMA regression head + "
                         f"46-class frame phoneme head, conv feature encoder frozen",
                "pr": f"Wav2Vec2_PR train step (models/w2v2_pr.py): wav2vec2-{args.model} + CTC head (40 phonemes), everything "
                      f"trainable incl. the conv feature encoder",
                "force": f"Force_APTAI train step (models/force_aptai.py): frozen wav2vec2-{args.model} CTC recogniser in inference "
                         f"mode + best-path decode + cross-attention forced aligner (forward-sum loss) + BiLSTM 9-track "
                         f"regression; only the heads train"}[wl]
        from aptai_amd.dp import collective_plan
        plan_world = world if world > 1 else max(args.plan_gpus, 1)
        if use_graph and wl in ("aptai", "pr"):
            plan = collective_plan(runner.plan_groups(), plan_world)
            plan["launched_from"] = "after each hipGraph backward segment, on a side stream (dp.GradGroupReducer)"
        else:
            plan = collective_plan((reducer or GradBucketReducer(params, bucket_mb=48.0, comm_dtype=torch.bfloat16)).plan_groups(), plan_world)
            plan["launched_from"] = "post-accumulate-grad hooks, 48 MB buckets in reverse parameter order (dp.GradBucketReducer)"
        plan["measured"] = world > 1
        res = {
            "metric": "utterances/sec (10 s @ 16 kHz) train-step", "value": round(world * B * args.steps / dt, 3),
            "unit": "utterances/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": ("bf16" if (wl != "force" or args.encoder_precision in ("bf16", "bf16_f32res")) else
                      "mxfp8 (E4M3 + E8M0 block scales) in the encoder's Linear layers, bf16 elsewhere, fp32 heads" if args.encoder_precision == "mxfp8" else
                      f"fp32-class encoder ({args.encoder_precision}: bf16 split-operand products, fp32 accumulation and element-wise math), fp32 heads"),
            "data": data_note,
            "config": {"workload": what,
                       "per_gpu_batch": B, "global_batch": world * B, "clip_seconds": args.seconds,
                       "parallelism": f"dp{world}", "ranks": world,
                       "process_group_world_size": (dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1),
                       "collective": (f"{backend} ({'RCCL over xGMI' if backend == 'nccl' else 'rehearsal on CPU tensors'}), "
                                      f"bucket average = {default_algo()}") if world > 1 else None,
                       "collective_plan": plan,
                       "regularisers": "off" if args.no_regularisers else "HF defaults",
                       "optimizer": "Adam, fp32 state (torch fused)" if args.torch_adam else "Adam, fp32 state (aptai_adam_multi)",
                       "execution": (("encoder hipGraph one batch ahead on a side stream + heads hipGraph (aptai_amd.graphed.GraphedForceStep)"
                                      if wl == "force" else "hipGraph segments (aptai_amd.graphed)") if use_graph else
                                     ("eager autograd loop, frozen-encoder pass one batch ahead on a side stream (Force_APTAI.prefetch)"
                                      if (wl == "force" and not args.no_pipeline) else "eager autograd loop")),
                       "inputs": "pinned host batch copied in every step (PCIe-inclusive)" if args.host_batch else "resident in HBM"},
            "loss": round(loss, 5),
            "roofline": {"bound": "mfma", "kernel": "bf16 MFMA GEMM, NT layout (gemm_kernel / gemm192_kernel / gemm256_kernel <false,false,false>): every launch of the step",
                         "achieved": round(gemm_tflops, 2), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(gemm_tflops / 2500.0, 4), "traffic": traffic,
                         "traffic_note": (f"HBM-side bytes per launch, launch-weighted over the same NT kernels, rocprofv3 PMC (2 x FETCH_SIZE + "
                                          f"WRITE_SIZE, separate passes), {traffic_src}") if traffic else None,
                         "measured": probe_note, "launches": ps["launches"], "avg_launch_us": round(ps["ms"] * 1e3 / max(ps["launches"], 1), 2),
                         "step_algorithmic_tflop": step_tf,
                         "step_executed_tflop": round(step_tf_exec, 4) if step_tf_exec else None,
                         "whole_step_frac_of_peak": round(step_tf / (dt / args.steps) / 2500.0, 4) if step_tf else None,
                         "whole_step_frac_of_peak_executed": round(step_tf_exec / (dt / args.steps) / 2500.0, 4) if step_tf_exec else None,
                         "flop_note": "step_algorithmic_tflop counts every transformer layer (SURVEY.md 8d); LayerDrop (p = 0.1) skips "
                                      "10 % of the layer work in expectation: *_executed prices the expected executed work"},
        }
        if wl == "force" and world == 1 and not args.no_exact_line and args.encoder_precision in ("bf16", "bf16_f32res", "mxfp8"):
            # Every Force_APTAI record carries BOTH figures (round-3 review): the line above is the bf16-operand encoder, whose alignment
            # indices agree with the reference's only outside the arithmetic noise (2-13 % of the decisions sit inside it); the
            # north_star-conformant figure ("alignment indices bit-exact") is the exact-index mode f32x3 (every index equal on the
            # reference fixture, tests/test_gpu_exact.py).  Measured by a child process of this script (its own model, graphs and caches).
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", "force", "--encoder-precision", "f32x3", "--model", args.model,
                   "--seconds", str(args.seconds), "--steps", str(max(3, min(args.steps, 10))), "--warmup", "2", "--no-cpu-baseline",
                   "--no-exact-line"] + (["--batch", str(args.batch)] if args.batch else []) + (["--no-regularisers"] if args.no_regularisers else [])
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                ex = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                res["index_exact"] = {"encoder_precision": "f32x3", "value": ex["value"], "unit": ex["unit"], "ms_per_step": ex["ms_per_step"],
                                      "steps": ex["steps"], "note": "same step with the frozen encoder in the exact-index mode (csrc/exact.hip): "
                                      "pred_frame_phns equal to the reference's on every frame; the figure north_star's 'alignment indices "
                                      "bit-exact' refers to.  The headline value of this record is the bf16-operand encoder (margin-exact)."}
            except Exception as e:          # noqa: BLE001 - the headline measurement stands; say what is missing
                res["index_exact"] = {"encoder_precision": "f32x3", "value": None, "note": f"second measurement failed: {e!r}"}
        if world == 1 and not args.no_cpu_baseline:
            if wl == "aptai":
                res["ema_rmse_vs_ref"] = ema_rmse_check(model, cfg, args, device)
            res["cpu_baseline"] = cpu_baseline(args, cfg)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
