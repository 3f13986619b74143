import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from aptai_amd.config import W2V2Config
from aptai_amd.graphed import GraphedAPTAIStep
from aptai_amd import ops
from oracle import synth
from test_gpu_aptai import _build
cfg = W2V2Config.base(num_hidden_layers=3, hidden_dropout=0., activation_dropout=0., attention_dropout=0.,
                      feat_proj_dropout=0., final_dropout=0., layerdrop=0., apply_spec_augment=False, vocab_size=46)
sd = synth.make_state_dict(synth.aptai_param_shapes(cfg), 0)
batch = {k: v.cuda() for k, v in synth.synth_aptai_batch(cfg, 2, 16000, seed=3).items()}
mA = _build(cfg, sd, tv_drop=0.0, phn_drop=0.0); mA.train()
oA = torch.optim.Adam([p for p in mA.parameters() if p.requires_grad], lr=1e-4, fused=True)
oA.zero_grad(set_to_none=True); mA(0, **batch)["loss"].backward(); oA.step()
mB = _build(cfg, sd, tv_drop=0.0, phn_drop=0.0); mB.train()
oB = torch.optim.Adam([p for p in mB.parameters() if p.requires_grad], lr=1e-4, fused=True)
r = GraphedAPTAIStep(mB, oB, batch)
r.step(); torch.cuda.synchronize()
pa, pb = dict(mA.named_parameters()), dict(mB.named_parameters())
bad = [(n, (pa[n] - pb[n]).abs().max().item()) for n in pa if (pa[n] - pb[n]).abs().max().item() > 1e-7]
print("params differing after 1 step:", len(bad), bad[:10])
print("state steps:", [int(oB.state[p]["step"]) for p in list(oB.state)[:3]], "n state", len(oB.state), "eager n state", len(oA.state))
# now second forward in graph mode: check caches
r.g_prep.replay(); torch.cuda.synchronize()
l0 = mB.wav2vec2.encoder.layers[0]
wt = r.lw[0][0]
print("wqkv cache vs param:", (wt.wqkv[:768].float() - l0.attention.q_proj.weight.to(torch.bfloat16).float()).abs().max().item())
print("w1 cache vs param:", (wt.w1.float() - l0.feed_forward.intermediate_dense.weight.to(torch.bfloat16).float()).abs().max().item())
print("b1 alias:", wt.b1.data_ptr() == l0.feed_forward.intermediate_dense.bias.data_ptr(), "bqkv ok:", (wt.bqkv[:768] - l0.attention.q_proj.bias).abs().max().item())
lossA = mA(0, **batch)["loss"].item()
lossB = r.step()["loss"].item()
print("second loss eager", lossA, "graph", lossB)
out_w = mA.wav2vec2(batch["audio_inputs"], attention_mask=batch["audio_lengths"][:, None], output_hidden_states=True)
g = r.g
for i, x in enumerate(r.X):
    hx = x.view(g.B, g.Tp, -1)[:, :g.T].float()
    print("X", i, (hx - out_w.hidden_states[i].float()).abs().max().item())
mA.eval(); 
from oracle import heads_ref
TV = ("LA", "LP", "JA", "TTCL", "TTCD", "TMCL", "TMCD", "TBCL", "TBCD")
sdA = {k: v.detach().cpu() for k, v in mB.state_dict().items()}
cb = {k: v.cpu() for k, v in batch.items()}
ref = heads_ref.aptai_forward(sdA, cfg, cb["audio_inputs"], cb["audio_lengths"], cb["phn_frames_49hz"], [cb[n] for n in TV], training=False)
print("oracle loss with mB's current params (after 2 graph steps):", ref["loss"].item())
mB.eval()
with torch.no_grad():
    print("mB eager eval loss (fresh caches by version):", mB(0, **batch)["loss"].item())
r2 = r.step()["loss"].item()
print("graph third-step loss (computed from the same params before its update):", r2)
