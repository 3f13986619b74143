"""GPU parity: aptai_gemm_bf16 (NT / NN / TN, epilogues, split-K) against fp32 torch on the CPU.

Inputs are rounded to bf16 first, so the only differences are fp32 accumulation order and the final
bf16 rounding of the output (rel 2^-8): tolerance 1e-2 * scale.  Integer-valued cases must be exact.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16)


def _rand(shape, g, scale=1.0):
    return _bf(torch.randn(shape, generator=g) * scale)


def _cmp(got, ref, tol=1e-2):
    got = got.float().cpu()
    scale = ref.abs().max().item() + 1e-6
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"max err {err} vs scale {scale}"


class _TileOps:
    """aptai_amd.ops with the GEMM tile forced (0 = the dispatcher's own choice)."""

    def __init__(self, ops, tile):
        self._ops, self._tile = ops, tile

    def __getattr__(self, k):
        return getattr(self._ops, k)

    def gemm(self, *a, **kw):
        tile = self._tile
        if tile == 257 and (a[2] < 256 or a[3] < 256 or kw.get("batch") is not None or kw.get("split_k", 1) > 1 or kw.get("accumulate")):
            tile = 0                                   # the stream-K form takes un-batched, un-split problems of >= one 256 x 256 tile
        if tile == 448 and (kw.get("a_kmajor") or kw.get("out_f32") or kw.get("batch") is not None or kw.get("split_k", 1) > 1
                            or kw.get("accumulate") or kw.get("residual_f32") is not None):
            tile = 0                                   # the 256 x 192 kernel: bf16 output, K-contiguous A, one un-batched un-split problem
        kw.setdefault("tile", tile)
        return self._ops.gemm(*a, **kw)


@pytest.fixture(scope="module", params=[0, 64, 128, 192, 256, 257, 448], ids=["auto", "t64", "t128", "t192", "t256", "streamk", "t256x192"])
def ops(request):
    from aptai_amd import ops
    return _TileOps(ops, request.param)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 768, 512), (1000, 768, 768), (499, 2304, 768),
                                   (130, 136, 192), (8192, 768, 3072)])
def test_nt_plain(ops, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a, b = _rand((M, K), g), _rand((N, K), g)
    out = ops.gemm(a.cuda(), b.cuda(), M, N, K)
    _cmp(out, a.float() @ b.float().t())


def test_nt_exact_identity_asymmetric():
    from aptai_amd import ops
    M = N = K = 256
    a = torch.eye(M)
    b = (torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] % 7).float() % 251   # asymmetric, exact in bf16
    out = ops.gemm(_bf(a).cuda(), _bf(b).cuda(), M, N, K)
    assert torch.equal(out.float().cpu(), _bf(b).float().t())


def test_nt_strided_rows_conv_like(ops):
    """Implicit-GEMM conv: overlapping A rows, lda = 2*C < K = 3*C (HF conv layer k=3, s=2)."""
    g = torch.Generator().manual_seed(3)
    C, Tout = 128, 300
    x = _rand((2 * Tout + 2, C), g)
    w = _rand((256, 3 * C), g, 0.1)
    out = ops.gemm(x.cuda(), w.cuda(), Tout, 256, 3 * C, lda=2 * C, gelu=True)
    A = torch.stack([x[2 * t:2 * t + 3].reshape(-1) for t in range(Tout)]).float()
    _cmp(out, torch.nn.functional.gelu(A @ w.float().t()))


def test_nt_epilogues(ops):
    g = torch.Generator().manual_seed(5)
    M, N, K = 777, 768, 256
    a, b = _rand((M, K), g), _rand((N, K), g, 0.1)
    bias = torch.randn(N, generator=g)
    res = _rand((M, N), g)
    aux = _rand((M, N), g)
    A, B = a.float(), b.float()
    base = A @ B.t() + bias
    out_pre = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    out = ops.gemm(a.cuda(), b.cuda(), M, N, K, bias=bias.cuda(), gelu=True, out_pre=out_pre)
    _cmp(out_pre, base)
    _cmp(out, torch.nn.functional.gelu(base))
    out = ops.gemm(a.cuda(), b.cuda(), M, N, K, bias=bias.cuda(), residual=res.cuda())
    _cmp(out, base + res.float())
    out = ops.gemm(a.cuda(), b.cuda(), M, N, K, dgelu_aux=aux.cuda(), residual=res.cuda())
    x = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    _cmp(out, (A @ B.t()) * x.grad + res.float())
    out = ops.gemm(a.cuda(), b.cuda(), M, N, K, alpha=0.125)
    _cmp(out, 0.125 * (A @ B.t()))


def test_nt_dropout_mask_is_reproducible_and_unbiased(ops):
    g = torch.Generator().manual_seed(6)
    M, N, K = 1024, 768, 64
    a, b = _rand((M, K), g), _rand((N, K), g)
    ref = a.float() @ b.float().t()
    o1 = ops.gemm(a.cuda(), b.cuda(), M, N, K, dropout_p=0.1, seed=1234).float().cpu()
    o2 = ops.gemm(a.cuda(), b.cuda(), M, N, K, dropout_p=0.1, seed=1234).float().cpu()
    o3 = ops.gemm(a.cuda(), b.cuda(), M, N, K, dropout_p=0.1, seed=99).float().cpu()
    assert torch.equal(o1, o2)
    keep = o1 != 0
    rate = 1.0 - keep.float().mean().item()
    assert abs(rate - 0.1) < 0.003, rate
    assert (keep != (o3 != 0)).float().mean().item() > 0.1          # different seed, different mask
    scale = 65536.0 / (65536.0 - round(0.1 * 65536))
    _cmp(o1[keep], ref[keep] * scale)
    # rows / columns are not correlated: per-column keep rate stays near 0.9
    assert (keep.float().mean(0) - 0.9).abs().max().item() < 0.05


@pytest.mark.parametrize("M,N,K", [(256, 768, 3072), (1000, 512, 128), (499, 3072, 768)])
def test_nn_dgrad(ops, M, N, K):
    """dX[M,N] = dY[M,K] . W[K,N]   (B stored K-major)."""
    g = torch.Generator().manual_seed(7 + M)
    dy, w = _rand((M, K), g), _rand((K, N), g, 0.1)
    res = _rand((M, N), g)
    out = ops.gemm(dy.cuda(), w.cuda(), M, N, K, b_kmajor=True, residual=res.cuda())
    _cmp(out, dy.float() @ w.float() + res.float())


def test_nn_exact_integer(ops):
    M, N, K = 128, 256, 128
    dy = (torch.arange(M)[:, None] == torch.arange(K)[None, :]).float() * 2.0     # 2*I
    w = ((torch.arange(K)[:, None] * 5 + torch.arange(N)[None, :] * 3) % 127).float()
    out = ops.gemm(_bf(dy).cuda(), _bf(w).cuda(), M, N, K, b_kmajor=True)
    assert torch.equal(out.float().cpu(), (2.0 * w))


@pytest.mark.parametrize("M,N,K,S", [(768, 768, 1024, 1), (2304, 768, 2048, 4), (768, 3072, 8192, 2),
                                     (512, 1536, 640, 3), (40, 768, 512, 2)])
def test_tn_wgrad(ops, M, N, K, S):
    """dW[M,N] = dY[K,M]^T . X[K,N] (both K-major), fp32 out, split-K slabs, accumulate."""
    g = torch.Generator().manual_seed(11 + M + S)
    dy, x = _rand((K, M), g), _rand((K, N), g)
    ref = dy.float().t() @ x.float()
    out = ops.gemm(dy.cuda(), x.cuda(), M, N, K, a_kmajor=True, b_kmajor=True, out_f32=True, split_k=S)
    _cmp(out, ref, tol=2e-5 * K ** 0.5)
    prev = torch.randn(M, N, generator=g)
    acc = prev.clone().cuda()
    ops.gemm(dy.cuda(), x.cuda(), M, N, K, a_kmajor=True, b_kmajor=True, out_f32=True, split_k=S, out=acc,
             accumulate=True)
    _cmp(acc, ref + prev, tol=2e-5 * K ** 0.5)


def test_tn_strided_operand(ops):
    """conv wgrad: X operand has overlapping rows (ldb = 2*C, N = 3*C)."""
    g = torch.Generator().manual_seed(13)
    C, T = 128, 512
    x = _rand((2 * T + 2, C), g)
    dy = _rand((T, 256), g)
    out = ops.gemm(dy.cuda(), x.cuda(), 256, 3 * C, T, a_kmajor=True, b_kmajor=True, out_f32=True, ldb=2 * C)
    A = torch.stack([x[2 * t:2 * t + 3].reshape(-1) for t in range(T)]).float()
    _cmp(out, dy.float().t() @ A, tol=1e-3)


def test_grouped_wgrad_and_bias_grads_match_separate_launches():
    """aptai_gemm_bf16_grouped: weight gradients dY^T X of different shapes plus column sums as (ones^T dY) problems in one
    launch; must equal the separate launches bit for bit (same tiles, same K order) and fp32 torch within bf16-input noise."""
    from aptai_amd import ops
    g = torch.Generator().manual_seed(11)
    K = 1024
    dy1, x1 = _rand((K, 768), g).cuda(), _rand((K, 256), g).cuda()
    dy2, x2 = _rand((K, 136), g).cuda(), _rand((K, 768), g).cuda()
    dy3, x3 = _rand((K, 2304), g).cuda(), _rand((K, 128), g).cuda()
    tn = dict(a_kmajor=True, b_kmajor=True, out_f32=True)
    ones = ops.ones_kmajor(K, dy1.device)
    outs = ops.gemm_grouped([(dy1, x1, 768, 256, K, tn), (dy2, x2, 136, 768, K, tn), (dy3, x3, 2304, 128, K, tn),
                             (ones, dy1, 8, 768, K, tn), (ones, dy3, 8, 2304, K, tn)])
    torch.cuda.synchronize()
    for (dy, x), got in zip(((dy1, x1), (dy2, x2), (dy3, x3)), outs[:3]):
        sep = ops.gemm(dy, x, dy.shape[1], x.shape[1], K, tile=128, **tn)
        assert torch.equal(got, sep)
        _cmp(got, dy.float().cpu().t() @ x.float().cpu(), tol=2e-3)
    for dy, got in ((dy1, outs[3]), (dy3, outs[4])):
        assert got.shape[0] == 8 and torch.equal(got[0], got[7])
        _cmp(got[0], dy.float().cpu().sum(0), tol=1e-3)


def test_grouped_rejects_mixed_layouts():
    from aptai_amd import ops, _lib
    g = torch.Generator().manual_seed(12)
    a, b = _rand((128, 64), g).cuda(), _rand((128, 64), g).cuda()
    with pytest.raises(_lib.AptaiHipError):
        ops.gemm_grouped([(a, b, 128, 128, 64, {}), (a, b, 64, 128, 128, dict(a_kmajor=True, b_kmajor=True, out_f32=True))])


def test_bad_arguments_fail_loudly(ops):
    from aptai_amd._lib import AptaiHipError
    a = torch.zeros(128, 100, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(AptaiHipError):
        ops.gemm(a, a, 128, 128, 100)              # K not a multiple of 64
    with pytest.raises(AptaiHipError):
        ops.gemm(a.cpu(), a.cpu(), 128, 128, 64)   # CPU tensors: no fallback


@pytest.mark.parametrize("M,N,K", [(8192, 3072, 768), (8192, 2304, 768), (8192, 768, 3072), (8192, 768, 768), (4096, 4096, 1024),
                                   (1000, 777 // 8 * 8, 320), (512, 256, 64), (300, 264, 6400), (2048, 2048, 128)])
def test_stream_k_256_tiles_against_fp32_and_the_tile_kernels(M, N, K):
    """tile 257: the persistent stream-K form of the 256 x 256 kernel.  Every (tile, K-tile) iteration is computed exactly once
    whatever the cut (partial tiles published as fp32 slabs and added by the tile's owner), edges included; launches repeat
    bit-identically (fixed summation order, self-cleaning flags); the status word stays clear."""
    from aptai_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a, b = _rand((M, K), g), _rand((N, K), g, 0.2)
    ref = a.float() @ b.float().t()
    ac, bc = a.cuda(), b.cuda()
    out = ops.gemm(ac, bc, M, N, K, tile=257)
    _cmp(out, ref)
    out32 = ops.gemm(ac, bc, M, N, K, tile=257, out_f32=True)
    assert (out32.cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() * (K ** 0.5)
    for _ in range(3):                                                   # replays: flags were re-zeroed by the owners
        assert torch.equal(ops.gemm(ac, bc, M, N, K, tile=257), out)
    # dgrad layout (B K-major) and the wgrad layout (both K-major, fp32 out)
    bk = b.t().contiguous()
    _cmp(ops.gemm(ac, bk.cuda(), M, N, K, b_kmajor=True, tile=257), ref)
    if M % 8 == 0:
        ak = a.t().contiguous()
        o = ops.gemm(ak.cuda(), bk.cuda(), M, N, K, a_kmajor=True, b_kmajor=True, out_f32=True, tile=257)
        assert (o.cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() * (K ** 0.5)
    # heavy epilogue through the owner / whole-tile paths alike
    bias = torch.randn(N, generator=g)
    res = _rand((M, N), g)
    o = ops.gemm(ac, bc, M, N, K, bias=bias.cuda(), residual=res.cuda(), tile=257)
    _cmp(o, ref + bias + res.float())
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    o = ops.gemm(ac, bc, M, N, K, bias=bias.cuda(), gelu=True, out_pre=pre, tile=257)
    _cmp(pre, ref + bias)
    o128 = ops.gemm(ac, bc, M, N, K, bias=bias.cuda(), gelu=True, tile=128)
    assert (o.float() - o128.float()).abs().max().item() <= 2e-2 * o128.float().abs().max().item()
    torch.cuda.synchronize()
    ops.gemm_sk_check()


@pytest.mark.parametrize("km", [False, True], ids=["nt", "nn"])
@pytest.mark.parametrize("M,N,K", [(8192, 3072, 768), (4096, 3072, 1024), (1000, 2304, 128), (300, 200, 64)])
def test_256x192_tile_equals_the_128_tile_bit_for_bit(M, N, K, km):
    """tile = 448 (csrc/gemm_t4.hip: 256 x 192 x 64, A double- / B triple-buffered, two wave groups one barrier apart).  A tile's K walk
    does not depend on its size (same 32-deep MFMA chain per output element), so plain and heavy-epilogue results - dropout masks and
    second outputs included - must EQUAL the 128-row kernel's, at whole-round shapes, ragged edges and a single K-tile; repeated launches
    must repeat (a hazard in the staging schedule would not)."""
    from aptai_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = _rand((M, K), g).cuda()
    b = (_rand((K, N), g, 0.1) if km else _rand((N, K), g, 0.1)).cuda()
    kw = dict(b_kmajor=True) if km else {}
    ref = ops.gemm(a, b, M, N, K, tile=128, **kw)
    _cmp(ref, a.float().cpu() @ (b.float().cpu() if km else b.float().cpu().t()))
    for _ in range(3):
        assert torch.equal(ops.gemm(a, b, M, N, K, tile=448, **kw), ref)
    bias = torch.randn(N, generator=g).cuda()
    res, aux = _rand((M, N), g).cuda(), _rand((M, N), g).cuda()
    p1, p2 = (torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for _ in range(2))
    heavy = [dict(bias=bias, gelu=True, out_pre=p1, pre_dgelu=True, dropout_p=0.1, seed=77), dict(bias=bias, residual=res, dropout_p=0.1, seed=78),
             dict(mul_aux=aux), dict(dgelu_aux=aux, residual=res), dict(bias=bias, colscale=(N // 8 * 8 // 2 // 8 * 8, 0.18))]
    for h in heavy:
        h2 = dict(h)
        if "out_pre" in h2:
            h2["out_pre"] = p2
        o1 = ops.gemm(a, b, M, N, K, tile=128, **kw, **h)
        o2 = ops.gemm(a, b, M, N, K, tile=448, **kw, **h2)
        assert torch.equal(o1, o2), (sorted(h), (o1.float() - o2.float()).abs().max().item())
        if "out_pre" in h:
            assert torch.equal(p1, p2)
