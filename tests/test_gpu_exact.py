"""GPU: the exact-index mode of the frozen encoder (Wav2Vec2Model.set_encoder_precision("f32x3" | "f32x6"), csrc/exact.hip).

BASELINE's north star: "alignment paths must match the reference ... alignment indices bit-exact".  With bf16 GEMM operands in front
of a 60-way argmax a few per cent of the decisions sit inside the arithmetic noise (tests/test_gpu_parity2.py measures the band).
In the exact mode every matrix product of the encoder is evaluated at fp32-class accuracy (bf16 split-operand products, fp32
accumulation, fp32 element-wise math), and the tests below demand EQUALITY ON EVERY FRAME with the reference fixture / the oracle,
with an absolute cap on the deviation of the deciding scores."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_parity2 import _att_scores, _force_setup

pytestmark = pytest.mark.gpu
TV = ("LA", "LP", "JA", "TTCL", "TTCD", "TMCL", "TMCD", "TBCL", "TBCD")


@pytest.mark.parametrize("pieces", [3, 6])
def test_split_operand_gemm_reaches_fp32_class_accuracy(pieces):
    """aptai_split_f32 + the NT kernel at K' = pieces * K against an fp64 product of the SAME fp32 operands: 3 pieces ~2^-17,
    6 pieces ~2^-23 per product (the plain bf16 kernel: 2^-9); overlapping rows (conv layers) and the erf GELU before the split."""
    from aptai_amd import ops
    g = torch.Generator().manual_seed(pieces)
    M, N, K = 512, 768, 1024
    a = torch.randn(M, K, generator=g) * 2.0
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().t() + bias.double() + res.double()
    a_s = ops.split_f32(a.cuda(), pieces)
    w_s = ops.split_f32(w.cuda(), pieces, weight_side=True)
    out = ops.gemm_split(a_s, w_s, M, N, K, pieces, bias=bias.cuda(), residual_f32=res.cuda())
    rel = ((out.cpu().double() - ref).norm() / ref.norm()).item()
    err = (out.cpu().double() - ref).abs().max().item()
    print(f"[exact] split GEMM, {pieces} pieces: rel-L2 {rel:.2e}, max abs {err:.2e}")
    assert rel < (8e-6 if pieces == 3 else 1e-6), rel
    bf = ops.gemm(a.cuda().to(torch.bfloat16), w.cuda().to(torch.bfloat16), M, N, K, out_f32=True).cpu().double() + bias.double() + res.double()
    assert ((bf - ref).norm() / ref.norm()).item() > 50 * rel                     # what the mode buys over bf16 operands
    # erf GELU in front of the split, and the strided overlapping-row view of a conv layer (k = 3, s = 2, C = 128)
    C, T = 128, 200
    x = torch.randn(2 * T + 2, C, generator=g)
    wc = torch.randn(256, 3 * C, generator=g) * 0.1
    xs = ops.split_f32(x.cuda(), pieces, gelu=True)
    o = ops.gemm_split(xs, ops.split_f32(wc.cuda(), pieces, weight_side=True), T, 256, 3 * C, pieces, lda=2 * C)
    gx = torch.nn.functional.gelu(x.double())
    A = torch.stack([gx[2 * t:2 * t + 3].reshape(-1) for t in range(T)])
    refc = A @ wc.double().t()
    assert ((o.cpu().double() - refc).norm() / refc.norm()).item() < (1e-5 if pieces == 3 else 1e-6)


def test_exact_elementwise_ops():
    from aptai_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1024, 768, generator=g) * 3
    bias, res = torch.randn(768, generator=g), torch.randn(1024, 768, generator=g)
    lens = torch.tensor([400, 512], dtype=torch.int32)
    y = ops.bias_act_res_f32(x.cuda(), bias=bias.cuda(), res=res.cuda(), gelu=True, lens_i32=lens.cuda(), rows_per_b=512).cpu()
    ref = res + torch.nn.functional.gelu(x + bias)
    ref[400:512] = 0
    assert (y - ref).abs().max().item() < 2e-6 * ref.abs().max().item() + 1e-6
    B, heads, Tp = 2, 3, 128
    s = torch.randn(B, heads, Tp, Tp, generator=g) * 4
    lens = torch.tensor([100, 128], dtype=torch.int32)
    p = ops.softmax_rows_f32(s.clone().cuda(), lens.cuda(), B, heads, Tp).cpu()
    for b in range(B):
        r = torch.softmax(s[b, :, :, :lens[b]], -1)
        assert (p[b, :, :, :lens[b]] - r).abs().max().item() < 2e-7 and p[b, :, :, lens[b]:].abs().sum().item() == 0.0


@pytest.mark.parametrize("prec,max_dev", [("f32x6", 4e-4), ("f32x3", 2e-3)])      # measured 7.7e-5 / 4.7e-4 (bf16 operands: 0.46)
def test_force_alignment_indices_equal_on_every_frame(prec, max_dev):
    """Force_APTAI with the exact-mode encoder: `pred_frame_phns` and the alignment slots EQUAL the reference's on every frame of the
    `force_aptai_1x2s` fixture (B = 1, the shipped reference's own output) and the oracle's at B = 2; log-attention scores within
    `max_dev` (bf16 operands: 0.46), trajectories within 1e-3 of the oracle (bf16: 1.4e-2)."""
    from oracle import heads_ref, synth
    model, pr_cfg, sd, z, meta = _force_setup()
    model.set_encoder_precision(prec)
    model.eval()
    # ---- B = 2 against the oracle
    batch = synth.synth_aptai_batch(pr_cfg, 2, 24000, seed=5, n_phn=40)
    with torch.no_grad():
        ref = heads_ref.force_aptai_forward(sd, pr_cfg, batch["audio_inputs"], batch["audio_lengths"], [batch[n] for n in TV])
        res, g, dec = model._run(batch["audio_inputs"].cuda(), batch["audio_lengths"].cuda(), phn_pred_list=ref["pred_ctc_phn_seq"])
        _, frame_lens, phn_lens, _ = model._lists(dec)
    sg = _att_scores(res[5].view(g.B, g.Tp, 60)[:, :g.T].float().cpu().numpy(), frame_lens, phn_lens)
    sr = _att_scores(ref["att"].numpy(), frame_lens, phn_lens)
    ig = np.concatenate([res[8].view(g.B, g.Tp)[b, :t].cpu().numpy() for b, t in enumerate(frame_lens)])
    ir = np.concatenate([ref["align_idx"][b, :t].numpy() for b, t in enumerate(frame_lens)])
    dev = float(np.abs(sg - sr).max())
    tv = (res[3].cpu() - ref["tvs_pred"]).abs().max().item() / ref["tvs_pred"].abs().max().item()
    top2 = np.sort(sr, -1)[:, -2:]
    print(f"[exact] {prec} B=2 vs oracle: max log-attention deviation {dev:.2e}, tvs deviation {tv:.2e}, smallest oracle top-2 margin "
          f"{(top2[:, 1] - top2[:, 0]).min():.3e}, {int((ig != ir).sum())} of {ig.size} alignment indices differ")
    assert dev <= max_dev and tv <= 1e-3
    assert np.array_equal(ig, ir)                                                  # every frame, no margin rule
    fp = res[4].cpu().numpy()
    for b, t in enumerate(frame_lens):
        assert fp[b, :t].tolist() == [int(v) for v in ref["pred_frame_phns"][b]]
    # the decode the model makes on its own (fp32 logits) equals the oracle's best path on every frame
    with torch.no_grad():
        res2, _, dec2 = model._run(batch["audio_inputs"].cuda(), batch["audio_lengths"].cuda())
        lists, _, _, _ = model._lists(dec2)
    assert all(list(a) == list(b) for a, b in zip(lists, ref["pred_ctc_phn_seq"]))
    # ---- B = 1: the reference's own output
    b1 = {k[len("b1/in/"):]: torch.from_numpy(z[k]).cuda() for k in z.files if k.startswith("b1/in/")}
    with torch.no_grad():
        out = model(0, **b1, _phn_pred_list=[z["b1/pred_ctc_phn_seq"]])
    assert [int(v) for v in out["pred_frame_phns"][0]] == [int(v) for v in z["b1/pred_frame_phns"]]
    assert np.abs(out["tvs_pred"].cpu().numpy() - z["b1/tvs_pred"]).max() <= 1e-3 * np.abs(z["b1/tvs_pred"]).max()
    for k in ("loss", "tv_loss", "align_loss"):
        assert abs(out[k].item() - float(z["b1/" + k])) <= 2e-4 * abs(float(z["b1/" + k])), (k, out[k].item(), float(z["b1/" + k]))
    model.set_encoder_precision("bf16_f32res")


def test_exact_encoder_hidden_states_base_and_large_against_the_oracle():
    """Every hidden state of the exact pass against the oracle, wav2vec2-base (GroupNorm conv0, post-LN) and -large (LayerNorm conv
    stack, conv bias, pre-LN) shapes at reduced depth, with a padded second utterance: max deviation 2e-4 of the state's scale
    (bf16 path: 4e-2)."""
    from aptai_amd.config import W2V2Config
    from aptai_amd.wav2vec2 import Wav2Vec2Model
    from oracle import synth, w2v2_ref
    for cfg in (W2V2Config.base(num_hidden_layers=3), W2V2Config.large(num_hidden_layers=3)):
        sd = synth.make_state_dict(synth.w2v2_param_shapes(cfg, "wav2vec2."), 2)
        model = Wav2Vec2Model(cfg)
        model.load_state_dict({k[len("wav2vec2."):]: v for k, v in sd.items()})
        model = model.cuda().eval().set_encoder_precision("f32x6")
        x = torch.randn(2, 16000, generator=torch.Generator().manual_seed(4))
        lens = torch.tensor([16000, 12345])
        with torch.no_grad():
            ref = w2v2_ref.wav2vec2_forward(sd, cfg, x, lens, "wav2vec2.")
            out = model(x.cuda(), attention_mask=lens[:, None].cuda(), output_hidden_states=True)
        fl = w2v2_ref.feat_extract_output_lengths(lens, cfg).tolist()
        assert len(out.hidden_states) == len(ref["hidden_states"]) == cfg.num_hidden_layers + 1
        worst = 0.0
        for got, want in zip(out.hidden_states, ref["hidden_states"]):
            for b in range(2):                                                     # valid frames (the reference's padded frames carry no meaning)
                d = (got[b, :fl[b]].float().cpu() - want[b, :fl[b]]).abs().max().item() / want[b, :fl[b]].abs().max().item()
                worst = max(worst, d)
        print(f"[exact] hidden states, H={cfg.hidden_size}: worst relative max deviation {worst:.2e}")
        assert worst < 2e-4


@pytest.mark.parametrize("prec,max_dev", [("f32x6", 8e-4), ("f32x3", 4e-3)])      # measured 1.5e-4 with 6 pieces (bf16 operands: 0.98)
def test_exact_mode_large_30s_alignment_equal_on_every_frame(prec, max_dev):
    """configs[4] shape (large, 30 s, 2 layers as the oracle allows): all 2 x ~1 400 alignment indices equal."""
    from oracle import heads_ref, synth
    from test_gpu_config5 import S30, _lists, _setup
    model, pr_cfg, sd = _setup(2)
    model.set_encoder_precision(prec)
    model.eval()
    batch = synth.synth_aptai_batch(pr_cfg, 2, S30, seed=11, n_phn=40)
    lists = _lists(2, 3)
    with torch.no_grad():
        ref = heads_ref.force_aptai_forward(sd, pr_cfg, batch["audio_inputs"], batch["audio_lengths"], [batch[n] for n in TV],
                                            phn_pred_list=lists)
        res, g, dec = model._run(batch["audio_inputs"].cuda(), batch["audio_lengths"].cuda(), phn_pred_list=lists)
        _, frame_lens, phn_lens, _ = model._lists(dec)
    sg = _att_scores(res[5].view(g.B, g.Tp, 60)[:, :g.T].float().cpu().numpy(), frame_lens, phn_lens)
    sr = _att_scores(ref["att"].numpy(), frame_lens, phn_lens)
    ig = np.concatenate([res[8].view(g.B, g.Tp)[b, :t].cpu().numpy() for b, t in enumerate(frame_lens)])
    ir = np.concatenate([ref["align_idx"][b, :t].numpy() for b, t in enumerate(frame_lens)])
    dev = float(np.abs(sg - sr).max())
    diff = np.nonzero(ig != ir)[0]
    top2 = np.sort(sr, -1)[:, -2:]
    print(f"[exact] {prec} large 30 s: max log-attention deviation {dev:.2e}, {diff.size} of {ig.size} indices differ"
          + "".join(f"; frame {i}: oracle margin {top2[i, 1] - top2[i, 0]:.2e}" for i in diff[:8]))
    assert dev <= max_dev
    assert diff.size == 0


@pytest.mark.parametrize("P", [3, 6])
def test_split_out_epilogue_equals_gemm_then_split(P):
    """APTAI_EPI_SPLIT_OUT (round 4): a split-operand GEMM whose fp32 result [-> erf GELU] leaves as the NEXT product's split A operand must
    equal the two-pass form it replaces (fp32 store, aptai_split_f32) bit for bit - plain, with bias + GELU (FFN1), and batched over
    (utterance, head) with the head's 64 columns as one K-tile of the output layout (attention context)."""
    from aptai_amd import ops
    g = torch.Generator().manual_seed(3)
    M, N, K = 300, 256, 128
    a32, w32 = torch.randn(M, K, generator=g).cuda(), (torch.randn(N, K, generator=g) * 0.2).cuda()
    bias = torch.randn(N, generator=g).cuda()
    a_s, w_s = ops.split_f32(a32, P), ops.split_f32(w32, P, weight_side=True)
    for gelu in (False, True):
        two = ops.split_f32(ops.gemm_split(a_s, w_s, M, N, K, P, bias=bias), P, gelu=gelu)
        one = ops.gemm_split(a_s, w_s, M, N, K, P, bias=bias, split_out=True, gelu=gelu)
        assert one.dtype == torch.bfloat16 and one.shape == (M, P * N)
        assert torch.equal(one.view(torch.int16), two.view(torch.int16)), gelu
    # batched: B x heads problems of [T][64] outputs written into [B T][P heads 64]
    B, heads, T, d = 2, 3, 128, 64
    p32 = torch.rand(B * heads * T, T, generator=g).cuda()
    vt32 = torch.randn(B * heads * d, T, generator=g).cuda()
    ps, vts = ops.split_f32(p32, P), ops.split_f32(vt32, P, weight_side=True)
    H = heads * d
    batch = lambda cs: dict(outer=B, inner=heads, a=(heads * T * P * T, T * P * T), b=(heads * d * P * T, d * P * T), c=cs)
    ctx32 = torch.empty(B * T, H, device="cuda")
    ops.gemm(ps, vts, T, d, P * T, lda=P * T, ldb=P * T, out=ctx32, ldc=H, out_f32=True, tile=128, batch=batch((T * H, d)))
    ctx_s = torch.empty(B * T, P * H, device="cuda", dtype=torch.bfloat16)
    ops.gemm(ps, vts, T, d, P * T, lda=P * T, ldb=P * T, out=ctx_s, ldc=P * H, out_f32=True, tile=128, split_out=P, batch=batch((T * P * H, P * d)))
    assert torch.equal(ctx_s.view(torch.int16), ops.split_f32(ctx32, P).view(torch.int16))
    # the 192- and 256-row kernels carry the same epilogue (conv stack / single-round shapes of the exact mode); 64-row tiles refuse it
    M2, N2 = 512, 384
    a2, w2 = ops.split_f32(torch.randn(M2, K, generator=g).cuda(), P), ops.split_f32((torch.randn(N2, K, generator=g) * 0.2).cuda(), P, weight_side=True)
    ref2 = ops.split_f32(ops.gemm(a2, w2, M2, N2, P * K, out_f32=True, tile=128), P, gelu=True)
    for tile in (192, 256):
        got = ops.gemm(a2, w2, M2, N2, P * K, out_f32=True, tile=tile, split_out=P, gelu=True)
        assert torch.equal(got.view(torch.int16), ref2.view(torch.int16)), tile
    with pytest.raises(Exception):
        ops.gemm(a2, w2, M2, N2, P * K, out_f32=True, tile=64, split_out=P)


def _unsplit(xs: torch.Tensor, P: int, weight_side: bool = False) -> torch.Tensor:
    """Sum of the distinct pieces of a split tensor [rows][cols P] (layout [cols / 64][slot][64]) in float64."""
    rows = xs.shape[0]
    t = xs.view(rows, -1, P, 64).double()
    slots = ((0, 1, 4) if weight_side else (0, 2, 5))[: (2 if P == 3 else 3)]
    return sum(t[:, :, s] for s in slots).reshape(rows, -1)


@pytest.mark.parametrize("P,tol", [(3, 1e-4), (6, 8e-6)])
def test_fused_exact_attention_against_float64_and_the_three_launch_form(P, tol):
    """aptai_attention_exact_fwd (round 4): scores, key-masked softmax, the split of the probabilities and P . V in one kernel.  Checked
    (a) against softmax(Q K^T / 8 + mask) V in float64 computed from the SAME split operands (what the pieces represent), ragged lengths
    incl. one utterance shorter than a key tile and one that fills Tp, scores with a spread of e^+-12 so the reference point of the
    running softmax is exercised; (b) against the three launches it replaces (aptai_gemm_bf16 scores -> aptai_softmax_split_f32 ->
    aptai_gemm_bf16 P . V) - same accuracy class; (c) slot duplication of the output layout bit for bit."""
    from aptai_amd import ops
    g = torch.Generator().manual_seed(11)
    B, heads, Tp, d = 3, 2, 256, 64
    H, M = heads * d, B * Tp
    lens = torch.tensor([256, 37, 190], dtype=torch.int32).cuda()
    q32 = (torch.randn(M, H, generator=g) * 3.0).cuda()
    k32 = (torch.randn(M, H, generator=g) * 3.0).cuda()
    v32 = torch.randn(M, H, generator=g).cuda()
    k32[5] *= 4.0                                           # one dominant key per utterance-0 head: a sharp row
    qkvs = torch.cat([ops.split_f32(q32, P), ops.split_f32(k32, P, weight_side=True), ops.split_f32(v32, P, weight_side=True)], dim=1).contiguous()
    ctx_s = ops.attention_exact_fwd(qkvs, lens, B, Tp, H, heads, P, d ** -0.5)
    torch.cuda.synchronize()
    assert ctx_s.shape == (M, P * H) and ctx_s.dtype == torch.bfloat16
    got = _unsplit(ctx_s, P).view(B, Tp, heads, d)
    q64 = _unsplit(qkvs[:, :P * H].contiguous(), P).view(B, Tp, heads, d)
    k64 = _unsplit(qkvs[:, P * H:2 * P * H].contiguous(), P, True).view(B, Tp, heads, d)
    v64 = _unsplit(qkvs[:, 2 * P * H:].contiguous(), P, True).view(B, Tp, heads, d)
    s = torch.einsum("bqhd,bkhd->bhqk", q64, k64) * d ** -0.5
    key = torch.arange(Tp, device="cuda")[None, None, None, :] >= lens.long()[:, None, None, None]
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s.masked_fill(key, float("-inf")), -1), v64)
    err = ((got - ref).abs().amax() / ref.abs().amax()).item()
    row_err = ((got - ref).abs().amax(-1) / ref.abs().amax(-1).clamp_min(1e-3)).amax().item()
    print(f"[exact] fused attention, {P} pieces: max deviation {err:.2e} of the largest context value, worst row {row_err:.2e}")
    # (b) the three-launch form on the same operands
    s32 = torch.empty((B, heads, Tp, Tp), device="cuda", dtype=torch.float32)
    ld = 3 * P * H
    ops.gemm(qkvs, qkvs[:, P * H:], Tp, Tp, P * d, lda=ld, ldb=ld, out=s32, ldc=Tp, out_f32=True, alpha=d ** -0.5, tile=128,
             batch=dict(outer=B, inner=heads, a=(Tp * ld, P * d), b=(Tp * ld, P * d), c=(heads * Tp * Tp, Tp * Tp)))
    ps = ops.softmax_split_f32(s32, lens, B, heads, Tp, P)
    vts = ops.split_f32(v32.view(B, Tp, heads, d).permute(2, 3, 0, 1).reshape(H, M).contiguous(), P, weight_side=True)     # V^T, split along the keys
    old_s = torch.empty((M, P * H), device="cuda", dtype=torch.bfloat16)
    ops.gemm(ps, vts, Tp, d, P * Tp, lda=P * Tp, ldb=P * M, out=old_s, ldc=P * H, out_f32=True, tile=128, split_out=P,
             batch=dict(outer=B, inner=heads, a=(heads * Tp * P * Tp, Tp * P * Tp), b=(Tp * P, d * P * M), c=(Tp * P * H, P * d)))
    old = _unsplit(old_s, P).view(B, Tp, heads, d)
    err_old = ((old - ref).abs().amax() / ref.abs().amax()).item()
    print(f"[exact] three-launch form on the same operands: {err_old:.2e}")
    assert err <= tol and row_err <= 20 * tol
    assert err <= max(2 * err_old, tol / 8)
    # (c) the duplicated slots of the activation-side order (h h m | h h m m h l) carry identical pieces
    t = ctx_s.view(M, heads, P, 64).view(torch.int16)
    assert torch.equal(t[:, :, 0], t[:, :, 1])
    if P == 6:
        assert torch.equal(t[:, :, 0], t[:, :, 4]) and torch.equal(t[:, :, 2], t[:, :, 3])
    # refusals
    with pytest.raises(Exception):
        ops.attention_exact_fwd(qkvs[:, :-8].contiguous(), lens, B, Tp, H, heads, P, d ** -0.5)
    with pytest.raises(Exception):
        _ = __import__("aptai_amd")._lib.call("aptai_attention_exact_fwd", qkvs.data_ptr(), ld, lens.data_ptr(), ctx_s.data_ptr(), P * H, B, 200, H,
                                              heads, P, 0.125, 0)
