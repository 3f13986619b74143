"""GPU: BASELINE configs[4] - Force_APTAI on the wav2vec2-LARGE shape with 30-second utterances (S = 480 000 samples, T = 1 499
frames; K + V of one head no longer fit the LDS).  The reference has no counterpart of this configuration beyond
"the same code on longer input", so parity is the oracle at reduced depth (2 layers, what the CPU finishes in seconds) and
size-independent properties at full depth (24 layers)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_force import _build
from test_gpu_parity2 import _att_scores, margin_exact

pytestmark = pytest.mark.gpu
TV = ("LA", "LP", "JA", "TTCL", "TTCD", "TMCL", "TMCD", "TBCL", "TBCD")
S30 = 480000


def _lists(B, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(2, 40, (int(torch.randint(20, 56, (1,), generator=g)),), generator=g).numpy() for _ in range(B)]


def _setup(layers):
    from aptai_amd.config import W2V2Config
    from oracle import synth
    z, meta = load_golden("force_aptai_1x2s")
    pr_cfg = W2V2Config.from_any(dict(meta["pr_cfg"], num_hidden_layers=layers))
    assert pr_cfg.hidden_size == 1024 and pr_cfg.do_stable_layer_norm and pr_cfg.feat_extract_norm == "layer"
    sd = synth.make_state_dict(synth.force_aptai_param_shapes(pr_cfg, meta["vocab_len"]), meta["seed"])
    model, _ = _build(dict(meta, pr_cfg=pr_cfg.to_dict()), sd)
    return model, pr_cfg, sd


def test_force_aptai_large_30s_against_the_oracle_at_reduced_depth():
    from oracle import heads_ref, synth
    model, pr_cfg, sd = _setup(2)
    model.train()
    model.hidden_drop = model.rnn_drop = 0.0
    batch = synth.synth_aptai_batch(pr_cfg, 2, S30, seed=11, n_phn=40)
    lists = _lists(2, 3)
    with torch.no_grad():
        ref = heads_ref.force_aptai_forward(sd, pr_cfg, batch["audio_inputs"], batch["audio_lengths"], [batch[n] for n in TV],
                                            phn_pred_list=lists)
    cb = {k: v.cuda() for k, v in batch.items()}
    cb["phoneme_labels"] = torch.zeros(2, 4, dtype=torch.int32).cuda()
    out = model(0, **cb, _phn_pred_list=lists)
    out["loss"].backward()
    torch.cuda.synchronize()
    assert out["tvs_pred"].shape == (2, 1499, 9) and ref["tvs_pred"].shape == (2, 1499, 9)
    for k in ("loss", "tv_loss", "align_loss"):
        assert abs(out[k].item() - ref[k].item()) <= 2e-2 * abs(ref[k].item()), (k, out[k].item(), ref[k].item())
    assert (out["tvs_pred"].cpu() - ref["tvs_pred"]).abs().max().item() <= 4e-2 * ref["tvs_pred"].abs().max().item()
    # alignment indices over 2 x ~1 400 frames: exact outside the measured noise band
    with torch.no_grad():
        res, g, dec = model._run(cb["audio_inputs"], cb["audio_lengths"], phn_pred_list=lists)
        _, frame_lens, phn_lens, _ = model._lists(dec)
    att_gpu = res[5].view(g.B, g.Tp, 60)[:, :g.T].float().cpu().numpy()
    sg, sr = _att_scores(att_gpu, frame_lens, phn_lens), _att_scores(ref["att"].numpy(), frame_lens, phn_lens)
    ig = np.concatenate([res[8].view(g.B, g.Tp)[b, :t].cpu().numpy() for b, t in enumerate(frame_lens)])
    ir = np.concatenate([ref["align_idx"][b, :t].numpy() for b, t in enumerate(frame_lens)])
    eps, frac = margin_exact("force alignment, large, 30 s, 2 layers", ig, ir, sr, sg, max_under=0.2, max_dev=1.5)        # 12.5 % under eps, deviation 0.98 measured
    got_ids = np.concatenate([np.asarray(out["pred_frame_phns"][b]) for b in range(2)])
    ref_ids = np.concatenate([np.asarray(ref["pred_frame_phns"][b]) for b in range(2)])
    top2 = np.sort(sr, -1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > eps
    assert (got_ids[clear] == ref_ids[clear]).all()
    heads = [(n, p) for n, p in model.named_parameters() if not n.startswith("w2v2_pr.") and p.requires_grad]
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in heads)
    # the read-out above ran with Force_APTAI's default, the fp32 residual stream in the frozen encoder; the plain bf16 stream
    # must not do better (it carries 2 more bf16 roundings per layer on the residual path and a bf16 last hidden state)
    model.set_encoder_precision("bf16")
    with torch.no_grad():
        res2, g2, dec2 = model._run(cb["audio_inputs"], cb["audio_lengths"], phn_pred_list=lists)
    model.set_encoder_precision("bf16_f32res")
    sg2 = _att_scores(res2[5].view(g.B, g.Tp, 60)[:, :g.T].float().cpu().numpy(), frame_lens, phn_lens)
    ig2 = np.concatenate([res2[8].view(g.B, g.Tp)[b, :t].cpu().numpy() for b, t in enumerate(frame_lens)])
    eps2, frac2 = margin_exact("force alignment, large, 30 s, 2 layers, all-bf16 residual stream", ig2, ir, sr, sg2, max_under=0.2, max_dev=1.6)
    assert frac <= frac2 + 0.02, (frac, frac2)


def test_force_aptai_large_30s_full_depth_properties():
    """24 layers, 2 x 30 s: shapes, finite losses and head gradients, aligned ids drawn from each utterance's own phoneme list,
    and utterance independence - the first utterance alone gives the same alignment and trajectories (no cross-utterance coupling
    anywhere on the path, padding included: the second utterance is shorter)."""
    from oracle import synth
    model, pr_cfg, _ = _setup(24)
    model.eval()
    batch = synth.synth_aptai_batch(pr_cfg, 2, S30, seed=12, n_phn=40)
    lists = _lists(2, 4)
    cb = {k: v.cuda() for k, v in batch.items()}
    cb["phoneme_labels"] = torch.zeros(2, 4, dtype=torch.int32).cuda()
    with torch.no_grad():
        out = model(0, **cb, _phn_pred_list=lists)
        one = model(0, **{k: v[:1] for k, v in cb.items()}, _phn_pred_list=lists[:1])
    assert out["tvs_pred"].shape == (2, 1499, 9)
    for k in ("loss", "tv_loss", "align_loss"):
        assert np.isfinite(out[k].item()), k
    for b in range(2):
        assert set(int(v) for v in out["pred_frame_phns"][b]) <= set(int(v) for v in lists[b])
    n0 = len(out["pred_frame_phns"][0])
    assert out["pred_frame_phns"][0] == one["pred_frame_phns"][0]
    # batch 1 runs its LSTM over all frames (models/modules.py:209-212); utterance 0 is full length, so the two runs compute the
    # same thing - up to fp32 summation order: the fp32 head GEMMs pick their K split from the batch size
    assert n0 == 1499 and (out["tvs_pred"][0] - one["tvs_pred"][0]).abs().max().item() < 1e-4
    model.train()
    out = model(0, **cb, _phn_pred_list=lists)
    out["loss"].backward()
    torch.cuda.synchronize()
    heads = [(n, p) for n, p in model.named_parameters() if not n.startswith("w2v2_pr.") and p.requires_grad]
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in heads)
    assert all(p.grad is None for n, p in model.named_parameters() if n.startswith("w2v2_pr."))


def test_force_aptai_large_30s_full_depth_with_the_mxfp8_encoder():
    """BASELINE configs[4] as written: 24 layers, 30 s, the frozen encoder's Linear layers on MX block-scaled FP8 operands.  No CPU
    oracle finishes 24 large layers on 30 s in test time, so the gate is relative to the bf16 run of the SAME model (itself gated
    against the oracle at reduced depth above): trajectories within the E4M3 error level of the bf16 ones, alignment indices equal
    on every frame whose bf16 top-2 margin exceeds 2 x the score deviation between the two runs (capped absolutely), finite losses."""
    from oracle import synth
    model, pr_cfg, _ = _setup(24)
    model.eval()
    batch = synth.synth_aptai_batch(pr_cfg, 2, S30, seed=14, n_phn=40)
    lists = _lists(2, 6)
    cb = {k: v.cuda() for k, v in batch.items()}
    runs = {}
    for mode in ("bf16_f32res", "mxfp8"):
        model.set_encoder_precision(mode)
        with torch.no_grad():
            res, g, dec = model._run(cb["audio_inputs"], cb["audio_lengths"], torch.stack([cb[n] for n in TV], dim=-1).float(),
                                     phn_pred_list=lists)
            _, frame_lens, phn_lens, _ = model._lists(dec)
        att = _att_scores(res[5].view(g.B, g.Tp, 60)[:, :g.T].float().cpu().numpy(), frame_lens, phn_lens)
        idx = np.concatenate([res[8].view(g.B, g.Tp)[b, :t].cpu().numpy() for b, t in enumerate(frame_lens)])
        runs[mode] = (att, idx, res[3].float().cpu(), res[0].item())
    model.set_encoder_precision("bf16_f32res")
    a16, i16, tv16, l16 = runs["bf16_f32res"]
    a8, i8, tv8, l8 = runs["mxfp8"]
    assert np.isfinite(l8) and abs(l8 - l16) <= 0.15 * abs(l16), (l8, l16)
    e = (tv8 - tv16).abs().max().item() / tv16.abs().max().item()
    e2 = ((tv8 - tv16).norm() / tv16.norm()).item()
    print(f"[mxfp8] 24 layers, 30 s: max |tvs(mxfp8) - tvs(bf16)| / max|tvs| = {e:.4f}, rel-L2 {e2:.4f}")
    # a real fp8 run (not the bf16 kernels) at the E4M3 error level: 24 layers of 3-bit-mantissa operands on random weights
    # (measured: max deviation 0.36 of the largest trajectory value over 2 x 1499 x 9 outputs)
    assert 1e-4 < e <= 0.6 and e2 <= 0.25
    # Finding, recorded rather than hidden: at full depth on random weights the E4M3 operand noise (score deviation 4.7 on energies
    # of O(40)) leaves only ~7 % of the 2 757 decisions with a margin above their row's noise; those agree, 439 of the rest differ.
    # The MX-fp8 encoder is an approximate mode - index-exact work belongs to set_encoder_precision("f32x3" | "f32x6").
    margin_exact("force alignment, large, 30 s, 24 layers, mxfp8 vs bf16 encoder", i8, i16, a16, a8, max_under=0.97, max_dev=7.0)
