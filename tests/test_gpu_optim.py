"""GPU: aptai_amd.optim.Adam (one multi-tensor HIP kernel, csrc/optim.hip) against torch.optim.Adam - the optimiser the
reference builds at train/train_aptai.py:350-356 - on the same parameters and gradients: identical update rule (bias
corrections per parameter step count, eps outside the square root, L2 weight decay), fp32 agreement to 1e-6 relative over
several steps, parameters that skip steps (LayerDrop), odd sizes, interchangeable state dicts, fused bf16 copies."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(768, 768), (3072,), (46, 768), (13,), (1, 1, 128), (5, 7)]


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in SHAPES]


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_matches_torch_adam(wd):
    from aptai_amd.optim import Adam
    a, b = _params(1), _params(1)
    oa = Adam(a, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    ob = torch.optim.Adam(b, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    g = torch.Generator().manual_seed(2)
    for step in range(6):
        for i, (pa, pb) in enumerate(zip(a, b)):
            if i == 1 and step in (1, 2):                 # a LayerDrop'd parameter: no gradient on some steps
                pa.grad = pb.grad = None
                continue
            gr = torch.randn(pa.shape, generator=g).cuda()
            pa.grad, pb.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    for pa, pb in zip(a, b):
        assert (pa - pb).abs().max().item() <= 1e-6 * pb.abs().max().item() + 1e-7
    for pa, pb in zip(a, b):
        sa, sb = oa.state[pa], ob.state[pb]
        assert int(sa["step"]) == int(sb["step"].item())
        assert (sa["exp_avg"] - sb["exp_avg"]).abs().max().item() <= 1e-6
        assert (sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max().item() <= 1e-6


def test_state_dict_interchanges_with_torch():
    from aptai_amd.optim import Adam
    a, b = _params(3), _params(3)
    oa, ob = Adam(a, lr=1e-3), torch.optim.Adam(b, lr=1e-3)
    g = torch.Generator().manual_seed(4)
    for p, q in zip(a, b):
        gr = torch.randn(p.shape, generator=g).cuda()
        p.grad, q.grad = gr.clone(), gr.clone()
    oa.step()
    ob.step()
    sd = oa.state_dict()
    assert torch.is_tensor(sd["state"][0]["step"])
    ob2 = torch.optim.Adam(_params(3), lr=1e-3)
    ob2.load_state_dict(sd)                               # ours -> torch
    oa2 = Adam(_params(3), lr=1e-3)
    oa2.load_state_dict(ob.state_dict())                  # torch -> ours
    assert all(int(st["step"]) == 1 for st in oa2.state.values())


def test_refuses_cpu_parameters():
    from aptai_amd import _lib
    from aptai_amd.optim import Adam
    p = torch.nn.Parameter(torch.randn(8))
    p.grad = torch.randn(8)
    with pytest.raises(_lib.AptaiHipError):
        Adam([p]).step()


def test_publishes_fresh_bf16_copies_to_the_model():
    """publish_to: after optimizer.step() the layer weight copies equal bf16(parameter) without any cast launch, and the
    training forward does not re-run the cast plan."""
    from aptai_amd.config import W2V2Config
    from aptai_amd.optim import Adam
    from oracle import synth
    from test_gpu_aptai import _build
    cfg = W2V2Config.base(num_hidden_layers=2, layerdrop=0.0, vocab_size=46)
    sd = synth.make_state_dict(synth.aptai_param_shapes(cfg), 0)
    model = _build(cfg, sd)
    model.train()
    batch = {k: v.cuda() for k, v in synth.synth_aptai_batch(cfg, 2, 16000, seed=3).items()}
    opt = Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3).publish_to(model)
    plan = model.wav2vec2._layer_plan()
    runs = []
    orig = plan.run
    plan.run = lambda: (runs.append(1), orig())[1]
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        model(0, **batch)["loss"].backward()
        opt.step()
    assert len(runs) == 1                                  # only the very first forward casts
    l0 = model.wav2vec2.encoder.layers[0]
    e = plan.entries[0]
    H = cfg.hidden_size
    torch.cuda.synchronize()
    assert torch.equal(e.w1, l0.feed_forward.intermediate_dense.weight.detach().to(torch.bfloat16))
    assert torch.equal(e.wqkv[H:2 * H], l0.attention.k_proj.weight.detach().to(torch.bfloat16))
    assert torch.equal(e.bqkv[0:H], l0.attention.q_proj.bias.detach())
