"""Host-side (Python) cost of the optimiser step inside the graph runner, torch fused Adam vs aptai_amd.optim.Adam."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from types import SimpleNamespace
args = SimpleNamespace(model="base", no_regularisers=False, n_tv=12)
dev = torch.device("cuda", 0)
for which in ("torch", "ours"):
    model, cfg = bench.build_model(args, dev)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    if which == "torch":
        opt = torch.optim.Adam(params, lr=1e-5, fused=True)
    else:
        from aptai_amd.optim import Adam
        opt = Adam(params, lr=1e-5).publish_to(model)
    batch = bench.synth_batch(cfg, 16, 160000, 12, 0, dev)
    from aptai_amd.graphed import GraphedAPTAIStep
    runner = GraphedAPTAIStep(model, opt, batch)
    orig = opt.step
    acc = []
    def timed(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k); acc.append(time.perf_counter() - t0); return r
    opt.step = timed
    for _ in range(5): runner.step()
    torch.cuda.synchronize(); acc.clear()
    t0 = time.perf_counter()
    for _ in range(20): runner.step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{which}: opt.step host {1e3*sum(acc)/len(acc):.2f} ms/step; host loop {1e3*(t1-t0)/20:.2f} ms/step; wall {1e3*(t2-t0)/20:.2f} ms/step")
    runner.close()
