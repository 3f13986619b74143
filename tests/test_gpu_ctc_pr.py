"""GPU parity: the CTC kernels against the golden vectors torch's F.ctc_loss produced (the call the reference makes),
and Wav2Vec2_PR.forward (BASELINE config 1 backbone) against the reference's own outputs / the oracle."""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_ctc_against_torch_golden():
    from aptai_amd import ops
    z, _ = load_golden("ops_small")
    logits = torch.from_numpy(z["ctc/logits"])                       # (T,B,V)
    T, B, V = logits.shape
    rows = logits.permute(1, 0, 2).contiguous().view(B * T, V).cuda()
    tg = torch.from_numpy(z["ctc/targets"]).to(torch.int32).cuda()
    il = torch.from_numpy(z["ctc/input_lengths"]).to(torch.int32).cuda()
    tl = torch.from_numpy(z["ctc/target_lengths"]).to(torch.int32).cuda()
    for red in ("mean", "sum"):
        for zi in (True, False):
            loss, nll, lp, alpha = ops.ctc_fwd(rows, V, T, tg, il, tl, B, T, V, reduction=red, zero_infinity=zi)
            ref = z[f"ctc/loss_{red}_zi{int(zi)}"]
            assert np.allclose(loss.item(), ref, rtol=2e-5, atol=1e-5, equal_nan=True) or (np.isinf(ref) and np.isinf(loss.item())), (red, zi, loss.item(), ref)
    loss, nll, lp, alpha = ops.ctc_fwd(rows, V, T, tg, il, tl, B, T, V, reduction="none", zero_infinity=False)
    assert np.allclose(nll.cpu().numpy(), z["ctc/loss_none_zi0"], rtol=2e-5, atol=1e-5)
    ref_lp = torch.log_softmax(logits, -1).numpy()
    assert np.abs(lp.cpu().numpy() - ref_lp).max() < 1e-5
    loss, nll, lp, alpha = ops.ctc_fwd(rows, V, T, tg, il, tl, B, T, V, reduction="mean", zero_infinity=True)
    g = ops.ctc_bwd(rows, V, T, tg, il, tl, B, T, V, alpha, nll, None, reduction="mean", zero_infinity=True, ldd=V,
                    out_dtype=torch.float32)
    got = g.view(B, T, V).permute(1, 0, 2).cpu().numpy()
    assert np.abs(got - z["ctc/grad_logits_mean_zi1"]).max() < 2e-6


def test_ctc_long_targets_and_oracle():
    """S = 2L+1 up to 401 states (NS = 8 path), T = 300: against the oracle's alpha recursion."""
    from aptai_amd import ops
    from oracle.heads_ref import ctc_loss_ref
    g = torch.Generator().manual_seed(3)
    T, B, V = 300, 3, 46
    logits = torch.randn(T, B, V, generator=g).requires_grad_(True)
    tl = [200, 57, 1]
    tg = torch.randint(1, V, (B, 200), generator=g, dtype=torch.int32)
    il = [300, 250, 40]
    ref = ctc_loss_ref(torch.log_softmax(logits, -1), tg, il, tl, 0, "mean", True)
    ref.backward()
    rows = logits.detach().permute(1, 0, 2).contiguous().view(B * T, V).cuda()
    loss, nll, lp, alpha = ops.ctc_fwd(rows, V, T, tg.cuda(), torch.tensor(il, dtype=torch.int32).cuda(),
                                       torch.tensor(tl, dtype=torch.int32).cuda(), B, T, V)
    assert abs(loss.item() - ref.item()) < 2e-4 * abs(ref.item())
    gr = ops.ctc_bwd(rows, V, T, tg.cuda(), torch.tensor(il, dtype=torch.int32).cuda(), torch.tensor(tl, dtype=torch.int32).cuda(),
                     B, T, V, alpha, nll, None, ldd=V, out_dtype=torch.float32)
    got = gr.view(B, T, V).permute(1, 0, 2).cpu()
    assert (got - logits.grad).abs().max().item() < 1e-4 * logits.grad.abs().max().item()      # fast exp/log over 300 steps


def _build_pr(cfg, sd, vocab_n=40):
    from aptai_amd.w2v2_pr import Wav2Vec2_PR
    from safetensors.torch import save_file
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "config.json"), "w") as f:
            json.dump(cfg.to_dict(), f)
        save_file({k[len("wav2vec2."):]: v.contiguous() for k, v in sd.items() if k.startswith("wav2vec2.")},
                  os.path.join(tmp, "model.safetensors"))
        model = Wav2Vec2_PR(cfg, None, tmp, {f"p{i}": i for i in range(vocab_n)})
    model.load_state_dict(sd)
    return model.cuda()


@pytest.mark.parametrize("name", ["pr_base_mini_2x1s", "pr_base_2x4s"])
def test_pr_forward_against_reference_golden(name):
    """Wav2Vec2_PR.forward on the MI355X vs the reference run on CPU (pr_base_2x4s = BASELINE config 1)."""
    from aptai_amd.config import W2V2Config
    from oracle import synth
    z, meta = load_golden(name)
    cfg = W2V2Config.from_any(meta["cfg"])
    sd = synth.make_state_dict(synth.pr_param_shapes(cfg), meta["seed"])
    model = _build_pr(cfg, sd)
    model.train()                           # everything trainable, conv feature encoder included (reference default for PR)
    batch = {k: torch.from_numpy(z["in/" + k]).cuda() for k in ("input_values", "input_lengths", "phoneme_labels")}
    out = model(**batch)
    out["loss"].backward()
    torch.cuda.synchronize()
    ref_logits = z["train/phoneme_logits"]
    got = out["phoneme_logits"].float().cpu().numpy()
    scale = np.abs(ref_logits).max()
    assert np.abs(got - ref_logits).max() < 4e-2 * scale
    assert np.linalg.norm(got - ref_logits) < 1.5e-2 * np.linalg.norm(ref_logits)
    assert abs(out["loss"].item() - float(z["train/loss"])) < 1e-2 * abs(float(z["train/loss"]))
    lp = out["log_probs"].cpu().numpy()
    assert lp.shape == z["train/log_probs"].shape and np.abs(lp - z["train/log_probs"]).max() < 4e-2 * scale
    named = dict(model.named_parameters())
    bad = []
    worst = {"qk": 0.0, "other": 0.0}
    for key in z.files:
        if key.startswith("gnorm/"):
            n = key[6:]
            if "feature_extractor" in n or n.endswith("k_proj.bias"):
                continue
            got_n = named[n].grad.double().norm().item()
            ref_n = float(z[key])
            # q/k projections see the softmax Jacobian P*(dP - delta): a small difference of bf16-rounded terms, so
            # with random weights their (tiny) gradients carry the most bf16 noise -> wider band for those two.
            # Measured on MI355X (printed below): q/k <= 0.020, all others <= 0.0045; the bands are 3-4x that.
            tol = 0.06 if ("q_proj" in n or "k_proj" in n) else 2e-2
            worst["qk" if ("q_proj" in n or "k_proj" in n) else "other"] = max(worst["qk" if ("q_proj" in n or "k_proj" in n) else "other"],
                                                                               abs(got_n - ref_n) / (ref_n + 1e-30))
            if abs(got_n - ref_n) > tol * ref_n + 1e-9:
                bad.append((n, got_n, ref_n))
    print(f"[bands] {name}: worst gradient-norm deviation q/k {worst['qk']:.4f}, others {worst['other']:.4f}")
    assert not bad, bad[:10]
    # gradient slices of the conv stack (layer 0 incl. GroupNorm, layer 3) and a few others, element-wise
    for key in z.files:
        if key.startswith("gslice/"):
            n = key[7:]
            flat = named[n].grad.float().flatten().cpu()
            step = max(1, flat.numel() // 512)
            got_s, ref_s = flat[::step][:512].numpy(), z[key]
            rel = np.linalg.norm(got_s - ref_s) / (np.linalg.norm(ref_s) + 1e-30)
            print(f"[bands] {name}: slice rel-L2 {n}: {rel:.4f}")
            assert rel < 0.08, (n, rel)                 # measured <= 0.030 (layer-11 output_dense at 2 x 4 s)
    # eval helpers run and agree with the training logits (no dropout in the fixture)
    emb = model.get_embeddings(batch["input_values"], batch["input_lengths"])
    assert emb["last_transf_hidden"].shape[1] == cfg.hidden_size
    assert len(emb["phn_pred_seq_idx"]) == 2 and emb["frame_seq_lens"].tolist() == out["hidden_states"].new_tensor(0).new_tensor(
        model.wav2vec2._get_feat_extract_output_lengths(batch["input_lengths"]).tolist()).long().tolist()
