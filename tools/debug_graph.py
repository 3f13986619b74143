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
out_w = model.wav2vec2(batch["audio_inputs"], attention_mask=batch["audio_lengths"][:, None], output_hidden_states=True)
ref_h = [h.float().clone() for h in out_w.hidden_states]
ref = model(0, **batch)
print("eager loss", ref["loss"].item())
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, fused=True)
r = GraphedAPTAIStep(model, opt, batch)
out = r.step()
torch.cuda.synchronize()
print("graph loss", out["loss"].item(), out["mse_loss"].item(), out["ce_loss"].item(), "eager", ref["mse_loss"].item(), ref["ce_loss"].item())
g = r.g
for i, x in enumerate(r.X):
    hx = x.view(g.B, g.Tp, -1)[:, :g.T].float()
    print("X", i, (hx - ref_h[i]).abs().max().item(), ref_h[i].abs().max().item())
print("tvs", (out["tvs_pred"] - ref["tvs_pred"]).abs().max().item())
print("lens", r.lens_i32.tolist(), "tv_tgt eq", torch.equal(r.tv_tgt, torch.stack([batch[k] for k in batch if k not in ("audio_inputs","audio_lengths","phn_frames_49hz")], -1).float()))
print("phn eq", torch.equal(r.phn_tgt, batch["phn_frames_49hz"]))
out2 = r.step(); torch.cuda.synchronize(); print("graph loss 2", out2["loss"].item())
print("graph losses cont:", [r.step()["loss"].item() for _ in range(3)])
model2 = _build(cfg, sd, tv_drop=0.0, phn_drop=0.0); model2.train()
opt2 = torch.optim.Adam([p for p in model2.parameters() if p.requires_grad], lr=1e-4, fused=True)
ls = []
for _ in range(5):
    opt2.zero_grad(set_to_none=True); o = model2(0, **batch); o["loss"].backward(); opt2.step(); ls.append(o["loss"].item())
print("eager losses:", ls)
