"""GPU, two ranks on one card (gloo rendezvous, both ranks compute on cuda:0): SURVEY.md 8(e)'s parity definition with the
product model — the all-reduced data-parallel gradients of a sharded batch equal the single-process gradients of the whole
batch.  Utterance lengths differ between the shards, so the masked-mean losses (models/aptai.py:89-100) only agree with
the single-process run when the backward normalises by the GLOBAL valid counts (aptai_amd.dp.GlobalLossNorm).

Per-utterance arithmetic does not depend on the batch it sits in (every GEMM output element reduces over K in the same
order, GroupNorm / LayerNorm are per sample), so the two sides differ only by fp32 summation order over utterances:
tolerance 2e-3 relative L2 per parameter, against ~3e-2 without the global normalisation (asserted on the tv head weight).
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
B_GLOBAL, SAMPLES = 4, 16000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup():
    from aptai_amd.config import W2V2Config
    from oracle import synth
    from test_gpu_aptai import _build
    cfg = W2V2Config.base(num_hidden_layers=2, hidden_dropout=0., activation_dropout=0., attention_dropout=0.,
                          feat_proj_dropout=0., final_dropout=0., layerdrop=0., apply_spec_augment=False, vocab_size=46)
    sd = synth.make_state_dict(synth.aptai_param_shapes(cfg), 0)
    model = _build(cfg, sd, tv_drop=0.0, phn_drop=0.0)
    model.train()
    batch = synth.synth_aptai_batch(cfg, B_GLOBAL, SAMPLES, seed=5)
    batch["audio_lengths"] = torch.tensor([16000, 9000, 12000, 16000])       # shard 0: 49 + 27 frames, shard 1: 37 + 49
    batch = synth.apply_lengths(cfg, batch) if hasattr(synth, "apply_lengths") else _apply_lengths(cfg, batch)
    return cfg, model, batch


def _apply_lengths(cfg, batch):
    """Re-pads the synthetic batch to the lengths set above (audio zeros, TV -100, frame labels 0)."""
    from aptai_amd import hostlogic
    lens = batch["audio_lengths"]
    S = batch["audio_inputs"].shape[1]
    batch["audio_inputs"] = batch["audio_inputs"] * (torch.arange(S)[None, :] < lens[:, None])
    fl = hostlogic.feat_extract_output_lengths(lens, cfg.conv_kernel, cfg.conv_stride)
    T = batch["phn_frames_49hz"].shape[1]
    valid = torch.arange(T)[None, :] < fl[:, None]
    g = torch.Generator().manual_seed(11)
    batch["phn_frames_49hz"] = torch.where(valid, torch.randint(1, 46, (len(lens), T), generator=g), torch.zeros((), dtype=torch.long))
    for k in list(batch):
        if k not in ("audio_inputs", "audio_lengths", "phn_frames_49hz"):
            tv = torch.randn(len(lens), T, generator=g, dtype=torch.float64)
            batch[k] = torch.where(valid, tv, torch.full_like(tv, -100.0))
    return batch


def _worker(rank, world, port, mode, use_norm, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from aptai_amd.dp import GlobalLossNorm, GradBucketReducer, shard_batch
    cfg, model, batch = _setup()
    mine = {k: v.cuda() for k, v in shard_batch(batch, rank, world).items()}
    if use_norm:
        model.dp_loss_norm = GlobalLossNorm()
    params = [p for p in model.parameters() if p.requires_grad]
    red = GradBucketReducer(params, bucket_mb=4.0, comm_dtype=None)
    if mode == "eager":
        model(0, **mine)["loss"].backward()
        red.finish()
    else:
        from aptai_amd.graphed import GraphedAPTAIStep
        red.remove()
        runner = GraphedAPTAIStep(model, torch.optim.SGD(params, lr=0.0), mine, reducer=red)
        runner.step()
        runner.step()                                            # replayed counts / collectives, not the capture-time ones
        torch.cuda.synchronize()
    q.put((rank, {n: p.grad.detach().float().cpu().numpy().copy() for n, p in model.named_parameters() if p.grad is not None}))
    dist.barrier()
    dist.destroy_process_group()


def _run(mode, use_norm):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, use_norm, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue, time
    res, t0 = {}, time.time()
    while len(res) < world:
        try:
            r, g = q.get(timeout=2)
            res[r] = g
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > 240:                   # a rank died (its traceback is on stderr) or hangs
                for p in procs:
                    p.kill()
                raise AssertionError(f"data-parallel worker failed: exit codes {[p.exitcode for p in procs]}")
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _reference():
    cfg, model, batch = _setup()
    model(0, **{k: v.cuda() for k, v in batch.items()})["loss"].backward()
    return {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}


def _rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_dp_gradients_equal_single_process_global_batch(mode):
    ref = _reference()
    res = _run(mode, True)
    assert set(ref) <= set(res[0])
    for n in set(res[0]) - set(ref):                             # graph mode keeps static (zero) gradients for unused parameters
        assert not res[0][n].any(), n
    worst = max((_rel(torch.from_numpy(res[0][n]), ref[n]), n) for n in ref)
    assert worst[0] <= 2e-3, worst
    for n in res[0]:
        assert (res[0][n] == res[1][n]).all(), n                 # replicas hold identical averaged gradients


def test_per_rank_means_differ_from_the_global_batch():
    """The sensitivity of the test above: without GlobalLossNorm the shards' own valid counts weight the frames unevenly."""
    ref = _reference()
    res = _run("eager", False)
    n = "tv_head.2.weight"
    assert _rel(torch.from_numpy(res[0][n]), ref[n]) > 1e-2
