"""cProfile of the eager (autograd) train step: where the host time of the drop-in loop goes."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from types import SimpleNamespace
args = SimpleNamespace(model="base", no_regularisers=False, n_tv=12)
dev = torch.device("cuda", 0)
model, cfg = bench.build_model(args, dev)
model.train()
params = [p for p in model.parameters() if p.requires_grad]
from aptai_amd.optim import Adam
opt = Adam(params, lr=1e-5).publish_to(model)
batch = bench.synth_batch(cfg, 16, 160000, 12, 0, dev)
def step():
    opt.zero_grad(set_to_none=True)
    out = model(0, **batch)
    out["loss"].backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
