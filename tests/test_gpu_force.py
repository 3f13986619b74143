"""GPU parity of the Force_APTAI path: fp32 head kernels (LSTM, cross-attention softmax/argmax, sgemm) against torch on
the CPU, and the full model against the reference fixture (B=1, the only batch size the shipped reference can run) and
against the oracle at B=2 (intent of models/modules.py:203-208).  The encoder runs in bf16, the heads in fp32."""
import json
import os
import pickle
import tempfile

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TV = ("LA", "LP", "JA", "TTCL", "TTCD", "TMCL", "TMCD", "TBCL", "TBCD")


def _sgemm_ref(a, b):
    return (a.double() @ b.double()).float()


def test_sgemm_f32_on_the_matrix_cores_matches_fp64():
    """aptai_sgemm_f32 (v_mfma_f32_32x32x2_f32): every operand orientation, the bf16 A operand, batches, ragged edges, bias /
    alpha / accumulate and the split-K form, against an fp64 product (fp32 rounding only: 1e-6 relative)."""
    from aptai_amd import ops
    g = torch.Generator().manual_seed(3)

    def close(got, ref, tol=2e-6):
        err = (got.cpu().double() - ref.double()).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()) * (ref.shape[-1] ** 0 ) + tol, (err, ref.abs().max().item())

    # A row-major [M][K], B as nn.Linear weight [N][K]  (x W^T + b)
    for (M, N, K) in ((300, 128, 768), (64, 9, 256), (1000, 60, 128), (256, 2048, 256)):
        a, w, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
        got = ops.linear_f32(a.cuda(), w.cuda(), bias.cuda())
        close(got, _sgemm_ref(a, w.t()) + bias, 3e-6)
    # bf16 A (encoder output), strided rows
    a = torch.randn(512, 800, generator=g).to(torch.bfloat16)
    w = torch.randn(128, 768, generator=g) / 28
    got = ops.linear_f32(a.cuda()[:, :768], w.cuda(), None, rows=512, ldx=800)
    close(got, _sgemm_ref(a[:, :768].float(), w.t()), 3e-6)
    # A^T B with a long K (gradient shape): split-K and single pass agree with fp64
    M, N, K = 256, 512, 8192
    a, b = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    ref = _sgemm_ref(a.t(), b)
    for sk in (1, None, 7):
        got = ops.sgemm(a.cuda(), 1, M, b.cuda(), N, 1, M, N, K, split_k=sk)
        assert ((got.cpu().double() - ref.double()).norm() / ref.double().norm()).item() < 2e-6, sk
    # batched, alpha, accumulate into a strided C
    Bn, M, N, K = 3, 100, 60, 128
    a, b = torch.randn(Bn, M, K, generator=g), torch.randn(Bn, N, K, generator=g)
    c0 = torch.randn(Bn, M, 64, generator=g)
    c = c0.clone().cuda()
    ops.sgemm(a.cuda(), K, 1, b.cuda(), 1, K, M, N, K, out=c, ldc=64, alpha=0.5, accumulate=True, batch=Bn, bsa=M * K, bsb=N * K,
              bsc=M * 64)
    ref = c0.clone()
    ref[:, :, :N] += 0.5 * torch.einsum("bmk,bnk->bmn", a.double(), b.double()).float()
    close(c, ref, 3e-6)
    assert torch.equal(c.cpu()[:, :, N:], c0[:, :, N:])                       # columns beyond N untouched


def test_colsum_f32_two_stage():
    from aptai_amd import ops
    x = torch.randn(8192, 300, generator=torch.Generator().manual_seed(1))
    got = ops.colsum_f32(x.cuda()[:, :290], 8192, 290, ld=300).cpu()
    ref = x[:, :290].double().sum(0)
    assert (got.double() - ref).abs().max().item() < 2e-5 * 90


@pytest.mark.parametrize("B,lens", [(3, [50, 37, 5]), (16, [50, 50, 49, 48, 40, 33, 32, 31, 17, 16, 15, 9, 3, 2, 1, 50]),
                                    (20, [50] * 4 + [44, 30, 12] * 5 + [1])])
def test_lstm_kernels_match_torch_packed_lstm(B, lens):
    """Cooperating-workgroup BiLSTM (csrc/lstm.hip) against torch.nn.LSTM over packed sequences, forward and backward, incl. a batch
    that spans two 16-utterance groups, and against the serial one-block-per-utterance kernels."""
    from aptai_amd import ops
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    torch.manual_seed(0)
    T, Tp = 50, 128
    lstm = torch.nn.LSTM(256, 256, bidirectional=True, num_layers=1, batch_first=True)
    x = torch.randn(B, T, 256, requires_grad=True)
    out, _ = pad_packed_sequence(lstm(pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=False))[0], batch_first=True,
                                 total_length=T)
    gout = torch.randn_like(out)
    for b, L in enumerate(lens):
        gout[b, L:] = 0
    out.backward(gout)
    xp = torch.zeros(B, Tp, 256)
    xp[:, :T] = x.detach()
    wih = torch.cat([lstm.weight_ih_l0, lstm.weight_ih_l0_reverse]).detach()
    bsum = torch.cat([lstm.bias_ih_l0 + lstm.bias_hh_l0, lstm.bias_ih_l0_reverse + lstm.bias_hh_l0_reverse]).detach()
    # the cluster kernels take / return the gate axis interleaved (column dir * 1024 + unit * 4 + gate): W_ih's rows permuted going in,
    # everything with a gate axis permuted back coming out; the serial cross-check kernels keep torch's gate-major order
    perm, inv = ops.lstm_gate_perm("cuda")
    xproj_gm = ops.linear_f32(xp.view(B * Tp, 256).cuda(), wih.cuda(), bsum.cuda())
    xproj = ops.linear_f32(xp.view(B * Tp, 256).cuda(), wih.cuda()[perm].contiguous(), bsum.cuda()[perm].contiguous())
    assert torch.equal(xproj[:, inv], xproj_gm)
    whh = torch.stack([lstm.weight_hh_l0, lstm.weight_hh_l0_reverse]).detach().contiguous()
    lens_t = torch.tensor(lens, dtype=torch.int32).cuda()
    hout, gates, cst = ops.lstm_fwd(xproj, whh.cuda(), lens_t, B, Tp, T)
    torch.cuda.synchronize()
    assert ops.lstm_status("cuda:0") == 0
    got = hout.view(B, Tp, 512)[:, :T].cpu()
    assert (got - out.detach()).abs().max().item() < 2e-5
    for b, L in enumerate(lens):
        assert hout.view(B, Tp, 512)[b, L:].abs().max().item() == 0.0
    # the serial kernels see the same inputs: same recurrence, different summation order
    h2, g2, c2 = ops.lstm_fwd_serial(xproj_gm, whh.transpose(1, 2).contiguous().cuda(), lens_t, B, Tp, T)
    assert (h2 - hout).abs().max().item() < 1e-5
    for b, L in enumerate(lens):                                   # saved gates: the same numbers, permuted
        assert (gates.view(B, Tp, 2048)[b, :L][:, inv] - g2.view(B, Tp, 2048)[b, :L]).abs().max().item() < 1e-5
    dh = torch.zeros(B, Tp, 512)
    dh[:, :T] = gout
    dgates = ops.lstm_bwd(dh.view(B * Tp, 512).cuda(), whh.cuda(), lens_t, gates, cst, B, Tp, T)
    torch.cuda.synchronize()
    assert ops.lstm_status("cuda:0") == 0
    dg2 = ops.lstm_bwd_serial(dh.view(B * Tp, 512).cuda(), whh.cuda(), lens_t, g2, c2, B, Tp, T)
    dgates = dgates[:, inv].contiguous()                           # gate-major from here on, like torch's gradients
    assert (dg2 - dgates).abs().max().item() < 2e-5 * max(1.0, dg2.abs().max().item())
    dx = ops.sgemm(dgates, 2048, 1, wih.cuda(), 256, 1, B * Tp, 256, 2048).view(B, Tp, 256)[:, :T].cpu()
    assert (dx - x.grad).abs().max().item() < 5e-5 * max(1.0, x.grad.abs().max().item())
    dwih = ops.sgemm(dgates, 1, 2048, xp.view(B * Tp, 256).cuda(), 256, 1, 2048, 256, B * Tp).cpu()
    assert (dwih[:1024] - lstm.weight_ih_l0.grad).abs().max().item() < 2e-4
    assert (dwih[1024:] - lstm.weight_ih_l0_reverse.grad).abs().max().item() < 2e-4
    M = B * Tp
    dwhh0 = ops.sgemm(dgates[1:], 1, 2048, hout, 512, 1, 1024, 256, M - 1).cpu()
    dwhh1 = ops.sgemm(dgates[:, 1024:], 1, 2048, hout[1:, 256:], 512, 1, 1024, 256, M - 1).cpu()
    assert (dwhh0 - lstm.weight_hh_l0.grad).abs().max().item() < 2e-4
    assert (dwhh1 - lstm.weight_hh_l0_reverse.grad).abs().max().item() < 2e-4
    db = ops.colsum_f32(dgates, M, 2048).cpu()
    assert (db[:1024] - lstm.bias_ih_l0.grad).abs().max().item() < 2e-4


def test_xattn_softmax_and_alignment_bit_exact_on_golden_energy():
    """Fed the reference's own energies, the alignment argmax is bit-exact and the log-softmax matches to fp32 rounding."""
    from aptai_amd import ops
    z, _ = load_golden("force_aptai_1x2s")
    energy, ids = torch.from_numpy(z["b2/energy"]), torch.from_numpy(z["b2/phn_ids"])
    B, T, N = energy.shape
    mask = (ids != 0)
    raw = energy - (~mask).float()[:, None, :] * -1000.0            # undo the first mask: kernel input is q.k^T
    e, att, att_log, align = ops.xattn_softmax_fwd(raw.reshape(B * T, N).contiguous().cuda(), ids.int().cuda(), B, T, N)
    assert np.array_equal(align.cpu().numpy(), z["b2/align_idx"])
    assert np.allclose(att_log.view(B, T, N).cpu().numpy(), z["b2/att"], rtol=3e-7, atol=2e-5)     # values reach -2000
    assert np.allclose(e.view(B, T, N).cpu().numpy(), z["b2/energy"], rtol=3e-7, atol=2e-5)


def _pr_ckpt(tmp, pr_cfg, sd, vocab):
    from aptai_amd.w2v2_pr import Wav2Vec2_PR
    from safetensors.torch import save_file
    mdir = os.path.join(tmp, "w2v2")
    os.makedirs(mdir)
    with open(os.path.join(mdir, "config.json"), "w") as f:
        json.dump(pr_cfg.to_dict(), f)
    save_file({k[len("w2v2_pr.wav2vec2."):]: v.contiguous() for k, v in sd.items() if k.startswith("w2v2_pr.wav2vec2.")},
              os.path.join(mdir, "model.safetensors"))
    ck = os.path.join(tmp, "pr", "best-model-ckpt")
    os.makedirs(ck)
    torch.save({k[len("w2v2_pr."):]: v for k, v in sd.items() if k.startswith("w2v2_pr.")}, os.path.join(ck, "pytorch_model.bin"))
    pickle.dump({"pretrain_cfg": pr_cfg.to_dict(), "cache_dir": None, "huggingface_model_id": mdir},
                open(os.path.join(ck, "model_cfg.pkl"), "wb"))
    return os.path.join(tmp, "pr")


def _build(meta, sd):
    from aptai_amd.config import W2V2Config
    from aptai_amd.force_aptai import Force_APTAI
    pr_cfg = W2V2Config.from_any(meta["pr_cfg"])
    vocab = {"(blank)": 0, "(...)": 1}
    vocab.update({f"p{i}": i for i in range(2, 40)})
    with tempfile.TemporaryDirectory() as tmp:
        model = Force_APTAI(_pr_ckpt(tmp, pr_cfg, sd, vocab), "cuda", vocab)
    model.load_state_dict(sd)
    return model.cuda(), pr_cfg


def test_force_aptai_golden_b1():
    from oracle import synth
    from aptai_amd.config import W2V2Config
    z, meta = load_golden("force_aptai_1x2s")
    pr_cfg = W2V2Config.from_any(meta["pr_cfg"])
    sd = synth.make_state_dict(synth.force_aptai_param_shapes(pr_cfg, meta["vocab_len"]), meta["seed"])
    sd["w2v2_pr.pr_head.bias"][0] += meta["blank_bias"]
    model, _ = _build(meta, sd)
    model.train()
    model.hidden_drop = 0.0
    model.rnn_drop = 0.0
    batch = {k[len("b1/in/"):]: torch.from_numpy(z[k]).cuda() for k in z.files if k.startswith("b1/in/")}
    out = model(0, **batch, _phn_pred_list=[z["b1/pred_ctc_phn_seq"]])
    out["loss"].backward()
    torch.cuda.synchronize()
    for k in ("loss", "tv_loss", "align_loss"):
        assert abs(out[k].item() - float(z["b1/" + k])) <= 2e-2 * abs(float(z["b1/" + k])), (k, out[k].item(), z["b1/" + k])
    ref_tv = z["b1/tvs_pred"]
    assert np.abs(out["tvs_pred"].cpu().numpy() - ref_tv).max() <= 4e-2 * np.abs(ref_tv).max()
    # (frame-level alignment indices and the decode: tests/test_gpu_parity2.py::test_force_alignment_indices_exact_outside_the_measured_noise
    #  compares them exactly on every clear-margin frame of this fixture, with an absolute cap on the score deviation)
    named = dict(model.named_parameters())
    bad = []
    for key in z.files:
        if key.startswith("b1/gnorm/"):
            n = key[len("b1/gnorm/"):]
            got, ref = named[n].grad.double().norm().item(), float(z[key])
            # alignment-path gradients (xatt, frame_lin, phn_emb) sit behind softmaxes over random-weight energies of O(40):
            # flipping the bf16 rounding of 27 of the 4.7 M positional-conv weights moved them by 3 % (measured), so this
            # is a noise band around the reference; the exact pin is test_force_aptai_b2_against_oracle (fp32 encoder output
            # fed to the same head kernels: 2e-3)
            print(f"[bands] force b1 gradient norm {n}: deviation {abs(got - ref) / (ref + 1e-30):.4f}")
            # Measured: phn_emb_layer 0.082, xatt / frame_lin <= 0.015, everything behind the LSTM <= 0.004; bands = 1.5-3 x those.
            band = 0.125 if n.startswith("phn_emb_layer") else 0.04 if n.startswith(("xatt", "frame_lin")) else 0.012
            if abs(got - ref) > band * ref + 1e-7:
                bad.append((n, got, ref))
    assert not bad, bad
    assert all(p.grad is None for n, p in model.named_parameters() if n.startswith("w2v2_pr."))


def test_force_aptai_b2_against_oracle():
    from oracle import heads_ref, synth
    from aptai_amd.config import W2V2Config
    z, meta = load_golden("force_aptai_1x2s")
    pr_cfg = W2V2Config.from_any(meta["pr_cfg"])
    sd = synth.make_state_dict(synth.force_aptai_param_shapes(pr_cfg, meta["vocab_len"]), meta["seed"])
    sd["w2v2_pr.pr_head.bias"][0] += meta["blank_bias"]
    batch = synth.synth_aptai_batch(pr_cfg, 2, 24000, seed=5, n_phn=40)
    sdo = {k: v.clone() for k, v in sd.items()}
    for k, v in sdo.items():
        if v.dtype == torch.float32 and not k.startswith("w2v2_pr.") and k != "pe_phn.pe":
            v.requires_grad_(True)
    ref = heads_ref.force_aptai_forward(sdo, pr_cfg, batch["audio_inputs"], batch["audio_lengths"], [batch[n] for n in TV])
    ref["loss"].backward()
    model, _ = _build(meta, sd)
    model.train()
    model.hidden_drop = 0.0
    model.rnn_drop = 0.0
    cb = {k: v.cuda() for k, v in batch.items()}
    cb["phoneme_labels"] = torch.zeros(2, 4, dtype=torch.int32).cuda()
    out = model(0, **cb, _phn_pred_list=ref["pred_ctc_phn_seq"])
    out["loss"].backward()
    torch.cuda.synchronize()
    for k in ("loss", "tv_loss", "align_loss"):
        assert abs(out[k].item() - ref[k].item()) <= 2e-2 * abs(ref[k].item()), (k, out[k].item(), ref[k].item())
    assert (out["tvs_pred"].cpu() - ref["tvs_pred"]).abs().max().item() <= 4e-2 * ref["tvs_pred"].abs().max().item()
    named = dict(model.named_parameters())

    def check(tol, tag):
        bad, worst = [], {}
        for k, v in sdo.items():
            if v.grad is None:
                continue
            rel = ((named[k].grad.cpu().double() - v.grad.double()).norm() / (v.grad.double().norm() + 1e-30)).item()
            fam = k.split(".")[0]
            worst[fam] = max(worst.get(fam, 0.0), rel)
            if rel > (tol[fam] if isinstance(tol, dict) else tol):
                bad.append((k, round(rel, 5)))
        print(f"[bands] force b2 head gradients, {tag}: " + ", ".join(f"{f} {r:.4f}" for f, r in worst.items()))
        assert not bad, bad
    # bf16 encoder in front: the random-weight energies are O(40), so the peaky softmaxes amplify its 2^-9 noise on the
    # alignment path (per-family bands = ~1.5 x the rel-L2 deviations measured on MI355X, printed as [bands])
    check({"xatt": 0.085, "frame_lin": 0.085, "phn_emb_layer": 0.09, "rnn": 0.015}, "bf16 encoder")      # measured 0.056 / 0.055 / 0.058 / 0.010
    # same heads fed the ORACLE's fp32 encoder output: only fp32 summation order differs -> tight agreement, which
    # pins the head kernels' forward AND backward (CrossAttention, forward-sum/CTC, BiLSTM, MLP, FIR) exactly
    with torch.no_grad():
        e = heads_ref.pr_get_embeddings(sd, pr_cfg, batch["audio_inputs"], batch["audio_lengths"], prefix="w2v2_pr.")
    g = model.w2v2_pr.wav2vec2._geometry(2, 24000)
    ac = torch.zeros(2, g.Tp, pr_cfg.hidden_size)
    ac[:, :g.T] = e["last_transf_hidden"].permute(0, 2, 1)
    model.zero_grad(set_to_none=True)
    out = model(0, **cb, _phn_pred_list=ref["pred_ctc_phn_seq"], _ac_override=ac.view(2 * g.Tp, -1).cuda().contiguous())
    out["loss"].backward()
    for k in ("loss", "tv_loss", "align_loss"):
        assert abs(out[k].item() - ref[k].item()) <= 2e-4 * abs(ref[k].item()), (k, out[k].item(), ref[k].item())
    assert (out["tvs_pred"].cpu() - ref["tvs_pred"]).abs().max().item() <= 2e-4
    for b in range(2):
        assert out["pred_frame_phns"][b] == ref["pred_frame_phns"][b]                  # alignment indices bit-exact
    check(2e-3, "oracle fp32 embeddings")


def test_force_aptai_config3_size_step():
    """The forced-alignment step at 16 x 10 s on the REFERENCE's recogniser width (hidden 1024 / pre-LN / LayerNorm conv stack, the
    only shape models/force_aptai.py:43 can build) at 2 layers: one training step through the frozen encoder (inference),
    CrossAttention, forward-sum CTC, BiLSTM and the TV regression - shapes, padding conventions, finite gradients on the head
    parameters only, alignment rows that stay inside each utterance's phoneme list.  BASELINE configs[2] as written (a 12-layer
    wav2vec2-BASE recogniser) is tests/test_gpu_force_base.py."""
    from oracle import synth
    from aptai_amd.config import W2V2Config
    z, meta = load_golden("force_aptai_1x2s")
    pr_cfg = W2V2Config.from_any(meta["pr_cfg"])
    sd = synth.make_state_dict(synth.force_aptai_param_shapes(pr_cfg, meta["vocab_len"]), meta["seed"])
    model, _ = _build(meta, sd)
    model.train()
    B, S = 16, 160000
    batch = {k: v.cuda() for k, v in synth.synth_aptai_batch(pr_cfg, B, S, seed=8, n_phn=40).items()}
    g = torch.Generator().manual_seed(5)
    lists = [torch.randint(2, 40, (int(torch.randint(20, 56, (1,), generator=g)),), generator=g).numpy() for _ in range(B)]
    batch["phoneme_labels"] = torch.zeros(B, 4, dtype=torch.int32).cuda()
    out = model(0, **batch, _phn_pred_list=lists)
    out["loss"].backward()
    torch.cuda.synchronize()
    assert out["tvs_pred"].shape == (B, 499, 9)
    for k in ("loss", "tv_loss", "align_loss"):
        assert np.isfinite(out[k].item()), k
    assert len(out["pred_frame_phns"]) == B
    for b in range(B):
        frames = out["pred_frame_phns"][b]
        assert set(int(v) for v in frames) <= set(int(v) for v in lists[b])          # aligned ids come from the utterance's own list
    heads = [(n, p) for n, p in model.named_parameters() if not n.startswith("w2v2_pr.") and p.requires_grad]
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in heads), [n for n, p in heads if p.grad is None]
    assert all(p.grad is None for n, p in model.named_parameters() if n.startswith("w2v2_pr."))


def test_force_aptai_prefetched_encoder_is_bit_identical_to_inline():
    """Force_APTAI.prefetch: the frozen recogniser of batch i+1 on a side stream beside the heads of batch i.  Three training
    steps over two alternating batches, once inline and once pipelined, from the same initial state: every loss, every TV
    prediction and every head gradient must be EQUAL (the same kernels on the same inputs, only issued earlier), and a
    prefetch for tensors that are not the ones passed next must be ignored."""
    from oracle import synth
    from aptai_amd.config import W2V2Config
    from aptai_amd.optim import Adam
    z, meta = load_golden("force_aptai_1x2s")
    pr_cfg = W2V2Config.from_any(meta["pr_cfg"])
    sd = synth.make_state_dict(synth.force_aptai_param_shapes(pr_cfg, meta["vocab_len"]), meta["seed"])
    B, S = 4, 48000
    batches = []
    for seed in (8, 9):
        bt = {k: v.cuda() for k, v in synth.synth_aptai_batch(pr_cfg, B, S, seed=seed, n_phn=40).items()}
        bt["phoneme_labels"] = torch.zeros(B, 4, dtype=torch.int32).cuda()
        batches.append(bt)

    def run(pipelined):
        model, _ = _build(meta, sd)
        model.train()
        # the random-weight recogniser must decode 1..59 phonemes: push the blank up until it does (as bench.py does)
        with torch.no_grad():
            blank = model.w2v2_pr._blank()
            for _ in range(40):
                n = [len(l) for bt in batches for l in model.w2v2_pr._decode(model.w2v2_pr._logits_eval(bt["audio_inputs"], bt["audio_lengths"].reshape(-1)[:, None])[0])]
                if max(n) < 60 and min(n) >= 1:
                    break
                model.w2v2_pr.pr_head.bias[blank] += 0.25 if max(n) >= 60 else -0.25
            else:
                pytest.skip("no blank bias gives 1..59 phonemes on both batches")
        model.w2v2_pr.wav2vec2._step = 100
        params = [p for p in model.parameters() if p.requires_grad]
        opt = Adam(params, lr=1e-4)
        rec = []
        for i in range(3):
            bt, nb = batches[i % 2], batches[(i + 1) % 2]
            opt.zero_grad(set_to_none=True)
            ahead = (nb["audio_inputs"], nb["audio_lengths"]) if pipelined else None
            out = model(0, **bt, _prefetch_next=ahead)
            out["loss"].backward()
            rec.append((out["loss"].detach().clone(), out["tvs_pred"].detach().clone(),
                        [p.grad.detach().clone() for p in params], out["pred_ctc_phn_seq"]))
            opt.step()
        torch.cuda.synchronize()
        return rec, model

    inline, m0 = run(False)
    names = [n for n, p in m0.named_parameters() if p.requires_grad]
    piped, model = run(True)
    for step_i, ((l0, tv0, g0, s0), (l1, tv1, g1, s1)) in enumerate(zip(inline, piped)):
        # the same kernels on the same inputs, only issued earlier: EQUAL on every step.  (Round 3 had to allow rtol 1e-3 from the second
        # step on: the embedding scatter-add and the forward-sum occupancy sums used float atomics whose order depended on what else ran
        # on the chip.  Both are order-fixed now - csrc/force.hip embed_bwd_kernel, csrc/ctc.hip ctc_grad_kernel - so a race between the
        # side-stream encoder and the heads can no longer hide behind a tolerance.)
        assert torch.equal(l0, l1) and torch.equal(tv0, tv1), (step_i, (tv0 - tv1).abs().max().item(), (l0 - l1).abs().item())
        diff = [(n, (a - b).abs().max().item(), a.abs().max().item()) for n, a, b in zip(names, g0, g1) if not torch.equal(a, b)]
        assert not diff, (step_i, diff)
        assert all(np.array_equal(a, b) for a, b in zip(s0, s1))
    # a stale prefetch (other tensor objects) is dropped, not used
    other = {k: v.clone() for k, v in batches[0].items()}
    model.prefetch(batches[1]["audio_inputs"], batches[1]["audio_lengths"])
    out = model(0, **other)
    assert model._prefetched is None and np.isfinite(out["loss"].item())


def test_prefetch_graph_survives_forwards_of_other_batch_shapes():
    """The captured encoder pass holds the addresses of the recogniser's persistent scratch buffers.  A forward with ANOTHER batch
    shape in between (validation at batch 1 between training epochs) used to replace those buffers and the next replays read
    freed memory (NaN after a few steps): scratch buffers are now kept per size.  Replay == eager, bit for bit, before and
    after interleaved batch-1 and batch-2 forwards."""
    from oracle import synth
    from aptai_amd.config import W2V2Config
    z, meta = load_golden("force_aptai_1x2s")
    pr_cfg = W2V2Config.from_any(meta["pr_cfg"])
    sd = synth.make_state_dict(synth.force_aptai_param_shapes(pr_cfg, meta["vocab_len"]), meta["seed"])
    model, _ = _build(meta, sd)
    model.train()
    bs = [{k: v.cuda() for k, v in synth.synth_aptai_batch(pr_cfg, 2, 16000, seed=s, n_phn=40).items()} for s in (1, 2, 3)]
    b1 = {k: v.cuda() for k, v in synth.synth_aptai_batch(pr_cfg, 1, 16000, seed=9, n_phn=40).items()}

    def check(i):
        b = bs[i]
        model.prefetch(b["audio_inputs"], b["audio_lengths"])
        enc = model._take_prefetched(b["audio_inputs"], b["audio_lengths"])
        with torch.no_grad():
            ref = model._encode(b["audio_inputs"], b["audio_lengths"])
        torch.cuda.synchronize()
        assert torch.isfinite(enc.ac).all() and torch.equal(enc.ac, ref.ac) and torch.equal(enc.ids, ref.ids), i
    for i in (0, 1, 2, 0):                       # eager, capture, replays
        check(i)
    assert any(g is not None for g in model._enc_graphs.values())
    model.eval()
    with torch.no_grad():
        for _ in range(2):
            model._encode(b1["audio_inputs"], b1["audio_lengths"])     # another shape through the same persistent scratch keys
    model.train()
    for i in (1, 2, 0, 1):
        check(i)


def test_lstm_timeout_status_raises_on_the_product_path():
    """csrc/lstm.hip bounds every cross-workgroup wait and raises a status word on timeout; Force_APTAI reads it with the lengths
    (one transfer) and must refuse to return lists for a step whose BiLSTM output is incomplete.  Also: the exchange workspace is
    keyed by (device, stream), so two streams never share one."""
    from aptai_amd import ops
    from aptai_amd._lib import AptaiHipError
    from oracle import synth
    from aptai_amd.config import W2V2Config
    z, meta = load_golden("force_aptai_1x2s")
    pr_cfg = W2V2Config.from_any(meta["pr_cfg"])
    sd = synth.make_state_dict(synth.force_aptai_param_shapes(pr_cfg, meta["vocab_len"]), meta["seed"])
    model, _ = _build(meta, sd)
    model.eval()
    b = {k: v.cuda() for k, v in synth.synth_aptai_batch(pr_cfg, 2, 16000, seed=1, n_phn=40).items()}
    b["phoneme_labels"] = torch.zeros(2, 4, dtype=torch.int32).cuda()
    lists = [np.arange(2, 12), np.arange(5, 20)]
    with torch.no_grad():
        out = model(0, **b, _phn_pred_list=lists)                      # creates this stream's workspace; status clear
        assert np.isfinite(out["loss"].item())
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            model(0, **b, _phn_pred_list=lists)
        side.synchronize()
        k_cur, k_side = ("cuda:0", cur.cuda_stream), ("cuda:0", side.cuda_stream)
        assert k_cur in ops._LSTM_WS and k_side in ops._LSTM_WS        # the second stream got its own exchange area
        assert ops._LSTM_WS[k_cur].data_ptr() != ops._LSTM_WS[k_side].data_ptr()
        ws = ops._LSTM_WS[k_cur]
        ws[:4].view(torch.int32).fill_(1)                              # what a timed-out wait leaves behind
        with pytest.raises(AptaiHipError, match="timed out"):
            model(0, **b, _phn_pred_list=lists)
        assert ops.lstm_status("cuda:0") == 0                          # cleared by the raise: the next step is judged on its own
        model(0, **b, _phn_pred_list=lists)


def test_force_aptai_with_the_flashlight_framed_decode():
    """Wav2Vec2_PR.decoder = "flashlight" inside Force_APTAI (opt-in, PARITY UNPINNED - INTEGRATION.md "The CTC decoder"): the phoneme sequence
    the aligner embeds is the best path framed by the silence token, produced on the device (no host synchronisation in the step); the step
    runs, its alignment targets are ids of the framed sequence, and the default decode is what it was."""
    from oracle import synth
    from aptai_amd.config import W2V2Config
    z, meta = load_golden("force_aptai_1x2s")
    pr_cfg = W2V2Config.from_any(meta["pr_cfg"])
    sd = synth.make_state_dict(synth.force_aptai_param_shapes(pr_cfg, meta["vocab_len"]), meta["seed"])
    sd["w2v2_pr.pr_head.bias"][0] += meta["blank_bias"]
    model, _ = _build(meta, sd)
    model.train()
    batch = {k: v.cuda() for k, v in synth.synth_aptai_batch(pr_cfg, 2, 24000, seed=5, n_phn=40).items()}
    batch["phoneme_labels"] = torch.zeros(2, 4, dtype=torch.int32).cuda()
    sil = 1
    out0 = model(0, **batch)
    base = [list(map(int, q)) for q in out0["pred_ctc_phn_seq"]]
    model.w2v2_pr.decoder = "flashlight"
    try:
        out = model(0, **batch)
        out["loss"].backward()
        torch.cuda.synchronize()
        assert np.isfinite(out["loss"].item())
        framed = [list(map(int, q)) for q in out["pred_ctc_phn_seq"]]
        for f, q in zip(framed, base):
            assert f[0] == sil and f[-1] == sil and f in ([sil] + q + [sil], q + [sil], [sil] + q, q), (f, q)
            assert len(f) < model.max_phn_seq_len
        for frames, f in zip(out["pred_frame_phns"], framed):
            assert set(map(int, frames)) <= set(f)            # every frame is aligned to a phoneme of the framed sequence
    finally:
        model.w2v2_pr.decoder = "best_path"
    again = [list(map(int, q)) for q in model(0, **batch)["pred_ctc_phn_seq"]]
    assert again == base
