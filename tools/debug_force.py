import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_golden
from oracle import heads_ref, synth
from aptai_amd.config import W2V2Config
import test_gpu_force as tf
TV = tf.TV
z, meta = load_golden("force_aptai_1x2s")
pr_cfg = W2V2Config.from_any(meta["pr_cfg"])
sd = synth.make_state_dict(synth.force_aptai_param_shapes(pr_cfg, meta["vocab_len"]), meta["seed"])
sd["w2v2_pr.pr_head.bias"][0] += meta["blank_bias"]
for B, S, seed in ((1, 32000, 99), (2, 24000, 5)):
    batch = synth.synth_aptai_batch(pr_cfg, B, S, seed=seed, n_phn=40)
    sdo = {k: v.clone() for k, v in sd.items()}
    for k, v in sdo.items():
        if v.dtype == torch.float32 and not k.startswith("w2v2_pr.") and k != "pe_phn.pe":
            v.requires_grad_(True)
    ref = heads_ref.force_aptai_forward(sdo, pr_cfg, batch["audio_inputs"], batch["audio_lengths"], [batch[n] for n in TV])
    ref["loss"].backward()
    model, _ = tf._build(meta, sd)
    model.train(); model.hidden_drop = 0.0; model.rnn_drop = 0.0
    cb = {k: v.cuda() for k, v in batch.items()}
    cb["phoneme_labels"] = torch.zeros(B, 4, dtype=torch.int32).cuda()
    out = model(0, **cb, _phn_pred_list=ref["pred_ctc_phn_seq"])
    out["loss"].backward()
    print("B", B, "loss", out["loss"].item(), ref["loss"].item(), "seq lens", [len(x) for x in ref["pred_ctc_phn_seq"]])
    named = dict(model.named_parameters())
    for k, v in sdo.items():
        if v.grad is None: continue
        gp = named[k].grad.cpu().double()
        rel = ((gp - v.grad.double()).norm() / (v.grad.double().norm() + 1e-30)).item()
        print(f"  {k:40s} rel {rel:.4f} norm {gp.norm().item():.5f} ref {v.grad.double().norm().item():.5f}")
    if True:
        ge = named["phn_emb_layer.weight"].grad.cpu(); gr = sdo["phn_emb_layer.weight"].grad
        rows = (gr.abs().sum(1) > 0).nonzero().flatten().tolist()
        print("  emb rows with grad:", rows[:20])
        for r in rows[:8]:
            print("   row", r, ge[r].norm().item(), gr[r].norm().item())
