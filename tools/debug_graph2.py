import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from aptai_amd.config import W2V2Config
from aptai_amd.graphed import GraphedAPTAIStep
from oracle import synth
from test_gpu_aptai import _build
cfg = W2V2Config.base(num_hidden_layers=3, hidden_dropout=0., activation_dropout=0., attention_dropout=0.,
                      feat_proj_dropout=0., final_dropout=0., layerdrop=0., apply_spec_augment=False, vocab_size=46)
sd = synth.make_state_dict(synth.aptai_param_shapes(cfg), 0)
batch = {k: v.cuda() for k, v in synth.synth_aptai_batch(cfg, 2, 16000, seed=3).items()}
model = _build(cfg, sd, tv_drop=0.0, phn_drop=0.0); model.train()
model.zero_grad(set_to_none=True)
model(0, **batch)["loss"].backward()
ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=0.0, fused=True)
r = GraphedAPTAIStep(model, opt, batch)
for it in range(2):
    r.step(); torch.cuda.synchronize()
    bad = []
    for n, p in model.named_parameters():
        if n in ref:
            if p.grad is None: bad.append((n, "None")); continue
            d = (p.grad - ref[n]).abs().max().item(); sc = ref[n].abs().max().item()
            if d > 1e-3 * sc + 1e-12: bad.append((n, round(d / (sc + 1e-30), 4)))
    print("iter", it, "mismatching grads:", len(bad), bad[:12])
