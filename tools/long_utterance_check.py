import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from types import SimpleNamespace
args = SimpleNamespace(model="large", no_regularisers=False, n_tv=9)
dev = torch.device("cuda", 0)
model, cfg = bench.build_model(args, dev)
model.train()
batch = bench.synth_batch(cfg, 2, 480000, 9, 0, dev)
from aptai_amd.optim import Adam
opt = Adam([p for p in model.parameters() if p.requires_grad], lr=1e-5).publish_to(model)
for i in range(3):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = model(0, **batch)
    out["loss"].backward()
    opt.step()
    torch.cuda.synchronize()
    print(f"30 s x 2, large: step {i} loss {float(out['loss'].detach()):.4f} tvs {tuple(out['tvs_pred'].shape)} {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
gn = sum(float(p.grad.float().norm()) ** 2 for p in model.parameters() if p.grad is not None) ** 0.5
print("grad norm finite:", gn == gn and gn < 1e9, gn)
