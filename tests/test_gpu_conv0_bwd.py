"""GPU: backward of the first conv layer in GroupNorm mode (wav2vec2-base, the recogniser's trainable stack) - aptai_conv0_bwd, whose two
passes run their contractions on the fp32 matrix pipe - against torch autograd in fp64 with the kernels' own GELU (the logistic fit of
csrc/common.h: the comparison is about the conv / GroupNorm / reduction arithmetic, models: HF:260-266,317-323)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
A1, A3, A5 = 1.59499531, 7.40885562e-2, -7.23764583e-4


@pytest.mark.parametrize("B,S,ragged", [(2, 16000, False), (3, 20483, True)])
def test_conv0_backward_group_mode_against_fp64_autograd(B, S, ragged):
    from aptai_amd import ops
    g = torch.Generator().manual_seed(11)
    audio = torch.randn(B, S, generator=g)
    w = torch.randn(512, 1, 10, generator=g) * 0.3
    gamma = 1.0 + 0.1 * torch.randn(512, generator=g)
    beta = 0.1 * torch.randn(512, generator=g)
    T = (S - 10) // 5 + 1                                   # ragged: not a multiple of 16 (the matrix-pipe block) or 256 (a chunk)
    Ta = (T + 63) // 64 * 64
    dy = (torch.randn(B, Ta, 512, generator=g) * 0.5).to(torch.bfloat16)
    dy[:, T:] = 7.0 if ragged else 0.0                      # rows beyond T_real must not be read into any sum
    dev = "cuda"
    out = torch.empty(B, Ta, 512, device=dev, dtype=torch.bfloat16)
    stats = ops.conv0_fwd(audio.to(dev), w.to(dev), None, gamma.to(dev), beta.to(dev), 0, out, T, Ta, want_stats=True)
    dw, _, dg, db = ops.conv0_bwd(audio.to(dev), w.to(dev), None, gamma.to(dev), beta.to(dev), 0, dy.to(dev), T, Ta, stats)
    wd, gd, bd = w.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    v = torch.nn.functional.conv1d(audio.double()[:, None], wd, stride=5)
    xh = (v - v.mean(-1, keepdim=True)) / torch.sqrt(v.var(-1, unbiased=False, keepdim=True) + 1e-5)
    z = xh * gd[None, :, None] + bd[None, :, None]
    y = z * torch.sigmoid(z * (A1 + A3 * z * z + A5 * z ** 4))
    (y * dy[:, :T].double().transpose(1, 2)).sum().backward()
    for name, got, ref in (("dweight", dw, wd.grad), ("dgamma", dg, gd.grad), ("dbeta", db, bd.grad)):
        err = (got.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-5, (name, err)                      # measured 2-4e-7: fp32 sums of ~3 000 frames per channel


@pytest.mark.parametrize("S,Sb", [(19200, 32000), (23337, 24000)])
def test_conv0_group_statistics_follow_the_collated_batch_inside_a_bucket(S, Sb):
    """Bucketed hipGraphs (graphed.BucketedGraphedStep) run the first conv layer on a waveform zero-padded from the batch's own length S
    to a bucket Sb with per-step frame bounds (aptai_set_frame_bounds).  Frame T(S) of the padded waveform starts at sample 5 T(S) < S,
    so its window still covers 5..9 REAL samples of an utterance that fills the batch - a frame the reference (GroupNorm over the frames of
    the batch as collated, HF:317-323) does not have.  Round 3 summed the window moments and the backward's sums over the bucket's frames
    (advice r3, conv.hip:133): statistics and gradients must be BIT-IDENTICAL to the un-bucketed shape, with nonzero samples in the tail."""
    from aptai_amd import graphed, ops
    g = torch.Generator().manual_seed(5)
    B = 2
    audio = torch.randn(B, S, generator=g)                 # nonzero up to the very last sample
    w = torch.randn(512, 1, 10, generator=g) * 0.3
    gamma = 1.0 + 0.1 * torch.randn(512, generator=g)
    beta = 0.1 * torch.randn(512, generator=g)
    T, Tb = (S - 10) // 5 + 1, (Sb - 10) // 5 + 1
    assert 5 * T < S                                        # the frame that must not be counted does see real samples
    Ta, Tba = (T + 63) // 64 * 64, (Tb + 63) // 64 * 64
    dy = (torch.randn(B, Ta, 512, generator=g) * 0.5).to(torch.bfloat16)
    dy[:, T:] = 0
    dyb = torch.zeros(B, Tba, 512, dtype=torch.bfloat16)
    dyb[:, :Ta] = dy
    dev = "cuda"
    P = [t.to(dev) for t in (w, gamma, beta)]
    out = torch.empty(B, Ta, 512, device=dev, dtype=torch.bfloat16)
    stats = ops.conv0_fwd(audio.to(dev), P[0], None, P[1], P[2], 0, out, T, Ta, want_stats=True)
    ref = ops.conv0_bwd(audio.to(dev), P[0], None, P[1], P[2], 0, dy.to(dev), T, Ta, stats)
    audio_b = torch.nn.functional.pad(audio, (0, Sb - S)).to(dev)
    outb = torch.empty(B, Tba, 512, device=dev, dtype=torch.bfloat16)
    stream = torch.cuda.current_stream().cuda_stream
    bounds = torch.tensor([T, 1], device=dev, dtype=torch.int32)
    graphed._bind_bounds(stream, bounds)
    try:
        stats_b = ops.conv0_fwd(audio_b, P[0], None, P[1], P[2], 0, outb, Tb, Tba, want_stats=True)
        got = ops.conv0_bwd(audio_b, P[0], None, P[1], P[2], 0, dyb.to(dev), Tb, Tba, stats_b)
        torch.cuda.synchronize()
    finally:
        graphed._unbind_salt(stream)
    assert torch.equal(stats, stats_b)
    assert torch.equal(out[:, :T], outb[:, :T])
    for name, a, b in zip(("dweight", "dbias", "dgamma", "dbeta"), ref, got):
        if a is not None:
            assert torch.equal(a, b), (name, (a - b).abs().max().item())
    # and without the bounds the bucket's statistics are NOT the batch's (the test has teeth)
    stats_n = ops.conv0_fwd(audio_b, P[0], None, P[1], P[2], 0, outb, Tb, Tba, want_stats=True)
    assert not torch.equal(stats, stats_n)
