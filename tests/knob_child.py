"""Child process of tests/test_gpu_env_knobs.py: the library reads its environment knobs ONCE (function-local statics at the first call), so
every setting gets a process of its own.  Modes:
    splitn  <unused>   the column-split launch of aptai_gemm_bf16 (APTAI_GEMM_SPLITN=1 in the environment) against the forced single launch
    step    <out.pt>   APTAI (wav2vec2-base shape, 2 layers, regularisers off): one eager train step and two hipGraph-replayed steps
    pr      <out.pt>   Wav2Vec2_PR with its trainable GroupNorm conv stack: one eager train step
    force   <out.pt>   Force_APTAI: two steps of GraphedForceStep (encoder graph on the side stream) from a fixed state
and writes losses / predictions / gradients for the parent to compare."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def splitn():
    from aptai_amd import ops
    M, N, K = 8192, 3072, 768
    g = torch.Generator(device="cuda").manual_seed(5)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
    for km in (False, True):
        a = rnd(M, K)
        b = rnd(K, N) if km else rnd(N, K)
        bias = torch.randn(N, device="cuda", generator=g)
        if km:                                                  # the FFN2-dgrad form: x aux
            aux = rnd(M, N)
            one = ops.gemm(a, b, M, N, K, b_kmajor=True, mul_aux=aux, tile=128)
            two = ops.gemm(a, b, M, N, K, b_kmajor=True, mul_aux=aux)
            assert torch.equal(one.view(torch.int16), two.view(torch.int16))
            continue
        pre1 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        pre2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        kw = dict(bias=bias, gelu=True, pre_dgelu=True, dropout_p=0.1, seed=7)
        one = ops.gemm(a, b, M, N, K, out_pre=pre1, tile=128, **kw)
        two = ops.gemm(a, b, M, N, K, out_pre=pre2, **kw)
        torch.cuda.synchronize()
        assert torch.equal((one == 0), (two == 0)) and 0.08 < (two == 0).float().mean().item() < 0.13
        assert torch.equal(one.view(torch.int16), two.view(torch.int16))
        assert torch.equal(pre1.view(torch.int16), pre2.view(torch.int16))
    print("SPLITN-OK")


def _quiet(vocab):
    from aptai_amd.config import W2V2Config
    return W2V2Config.base(num_hidden_layers=2, hidden_dropout=0., activation_dropout=0., attention_dropout=0., feat_proj_dropout=0.,
                           final_dropout=0., layerdrop=0., apply_spec_augment=False, vocab_size=vocab, ctc_loss_reduction="mean",
                           ctc_zero_infinity=True)


def step(path):
    from aptai_amd.graphed import GraphedAPTAIStep
    from aptai_amd.optim import Adam
    from oracle import synth
    from test_gpu_aptai import _build
    cfg = _quiet(46)
    sd = synth.make_state_dict(synth.aptai_param_shapes(cfg), 0)
    model = _build(cfg, sd, tv_drop=0.0, phn_drop=0.0)
    model.train()
    batch = {k: v.cuda() for k, v in synth.synth_aptai_batch(cfg, 4, 48000, seed=11).items()}
    out = model(0, **batch)
    out["loss"].backward()
    rec = {"eager_loss": out["loss"].detach().cpu(), "eager_tvs": out["tvs_pred"].detach().float().cpu()}
    for n, p in model.named_parameters():
        if p.grad is not None:
            rec["eager_grad/" + n] = p.grad.detach().float().cpu()
    model.zero_grad(set_to_none=True)
    opt = Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4).publish_to(model)
    with GraphedAPTAIStep(model, opt, batch) as runner:
        for i in range(2):
            o = runner.step(batch)
            rec[f"graph_loss{i}"] = o["loss"].detach().cpu().clone()
        for n, p in model.named_parameters():
            if p.grad is not None:
                rec["graph_grad/" + n] = p.grad.detach().float().cpu().clone()
    torch.save(rec, path)


def pr(path):
    from oracle import synth
    from test_gpu_ctc_pr import _build_pr
    cfg = _quiet(40)
    sd = synth.make_state_dict(synth.pr_param_shapes(cfg), 0)
    model = _build_pr(cfg, sd)
    model.train()
    sb = synth.synth_aptai_batch(cfg, 2, 32000, seed=3)
    g = torch.Generator().manual_seed(9)
    lab = torch.full((2, 20), -100, dtype=torch.int64)
    for b, n in enumerate((20, 11)):
        lab[b, :n] = torch.randint(1, 40, (n,), generator=g)
    out = model(input_values=sb["audio_inputs"].cuda(), input_lengths=sb["audio_lengths"].reshape(-1).cuda(), phoneme_labels=lab.cuda())
    out["loss"].backward()
    rec = {"loss": out["loss"].detach().cpu(), "logits": out["phoneme_logits"].detach().float().cpu()}
    for n, p in model.named_parameters():
        if p.grad is not None:
            rec["grad/" + n] = p.grad.detach().float().cpu()
    torch.save(rec, path)


def force(path):
    from aptai_amd.config import W2V2Config
    from aptai_amd.graphed import GraphedForceStep
    from aptai_amd.optim import Adam
    from conftest import load_golden
    from oracle import synth
    from test_gpu_force import _build
    z, meta = load_golden("force_aptai_1x2s")
    pr_cfg = W2V2Config.from_any(meta["pr_cfg"])
    sd = synth.make_state_dict(synth.force_aptai_param_shapes(pr_cfg, meta["vocab_len"]), meta["seed"])
    sd["w2v2_pr.pr_head.bias"][0] += meta["blank_bias"]
    model, _ = _build(meta, sd)
    model.train()
    model.hidden_drop = 0.0
    model.rnn_drop = 0.0
    B = 4
    bt = {k: v.cuda() for k, v in synth.synth_aptai_batch(pr_cfg, B, 32000, seed=8, n_phn=40).items()}
    bt["phoneme_labels"] = torch.zeros(B, 4, dtype=torch.int32).cuda()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = Adam(params, lr=1e-4)
    rec = {}
    runner = GraphedForceStep(model, opt, bt)
    for i in range(2):
        o = runner.step(bt, next_batch=bt)
        rec[f"loss{i}"] = o["loss"].detach().cpu().clone()
        rec[f"tvs{i}"] = o["tvs_pred"].detach().float().cpu().clone()
        rec[f"n_ids{i}"] = o["n_ids"].detach().cpu().clone()
    runner.close()
    torch.save(rec, path)


if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    {"splitn": lambda p: splitn(), "step": step, "pr": pr, "force": force}[mode](path)
    print("CHILD-OK")
