"""CPU, world_size 2 over gloo: the data-parallel gradient path (aptai_amd/dp.py) — averaged DP gradients of the
sharded batch equal the single-process gradients of the whole batch, with incomplete buckets (LayerDrop'd parameters
that received no gradient) and a bf16 communication dtype."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(),
                               torch.nn.Linear(32, 4))


def _mock_rccl_semantics(world):
    """gloo has neither ReduceOp.AVG nor reduce_scatter_tensor.  To reach the branches dp.py takes ONLY under the "nccl" (RCCL)
    backend without RCCL hardware, report the backend as averaging-capable and emulate exactly those two calls on top of gloo:
    same call signatures, same results, so a typo or a wrong argument in the RCCL-only code surfaces here and not first on the
    8-GPU run."""
    import aptai_amd.dp as dp
    calls = {"avg_all_reduce": 0, "reduce_scatter": 0}
    real_all_reduce = dist.all_reduce
    dp._backend_has_avg = lambda group: True

    def all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if op == dist.ReduceOp.AVG:
            calls["avg_all_reduce"] += 1
            w = real_all_reduce(tensor, op=dist.ReduceOp.SUM, group=group, async_op=False)
            tensor.div_(world)

            class _Done:
                def wait(self):
                    return True
            return _Done() if async_op else w
        return real_all_reduce(tensor, op=op, group=group, async_op=async_op)

    def reduce_scatter_tensor(output, input, op=dist.ReduceOp.SUM, group=None, async_op=False):
        assert op == dist.ReduceOp.AVG and not async_op and input.numel() == world * output.numel()
        calls["reduce_scatter"] += 1
        tmp = input.clone()
        real_all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
        output.copy_(tmp.view(world, -1)[dist.get_rank(group)] / world)
    dist.all_reduce = all_reduce
    dist.reduce_scatter_tensor = reduce_scatter_tensor
    return calls


def _worker(rank, world, port, comm_bf16, q, algo="allreduce", mock_rccl=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["APTAI_DP_ALGO"] = algo
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aptai_amd.dp import GradBucketReducer, shard_batch
    calls = _mock_rccl_semantics(world) if mock_rccl else None
    model = _model()
    g = torch.Generator().manual_seed(1)
    batch = {"x": torch.randn(8, 16, generator=g), "y": torch.randn(8, 4, generator=g)}
    mine = shard_batch(batch, rank, world)
    red = GradBucketReducer(model.parameters(), bucket_mb=0.002, comm_dtype=torch.bfloat16 if comm_bf16 else None)
    assert len(red.buckets) >= 2
    for step in range(2):                                   # two steps: buckets reset correctly
        model.zero_grad(set_to_none=True)
        h = model[1](model[0](mine["x"]))
        if step == 1:
            out = model[4](h)                               # "LayerDrop": skip model[2] -> its params get no grad
        else:
            out = model[4](model[3](model[2](h)))
        # per-rank SUM scaled by the GLOBAL count, so the all-reduce AVERAGE times world == global-mean gradient
        loss = ((out - mine["y"]) ** 2).sum() / (batch["x"].shape[0] * 4) * world
        loss.backward()
        red.finish()
    # by value (the sender may exit first); a parameter no rank touched keeps grad None, like the single-process run
    q.put((rank, {n: (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy() for n, p in model.named_parameters()}))
    assert model[2].weight.grad is None
    if calls is not None:                                   # the RCCL-only branch of this algorithm really ran
        assert calls["avg_all_reduce" if algo == "allreduce" else "reduce_scatter"] >= 2, calls
    dist.destroy_process_group()


@pytest.mark.parametrize("comm_bf16,algo,mock_rccl", [(False, "allreduce", False), (True, "allreduce", False), (False, "rs_ag", False),
                                                      (True, "a2a", False), (False, "a2a", False),
                                                      (False, "allreduce", True), (False, "rs_ag", True)])
def test_dp_gradients_equal_single_process(comm_bf16, algo, mock_rccl):
    """Every bucket-averaging pattern of aptai_amd.dp (all-reduce, reduce-scatter + all-gather, direct all-to-all) gives the
    single-process gradient of the whole batch.  mock_rccl: the branches only the "nccl" backend takes (ReduceOp.AVG all-reduce,
    reduce_scatter_tensor with AVG), reached through an emulation of those two calls over gloo."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, comm_bf16, q, algo, mock_rccl)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = _model()
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g)
    out = model[4](model[1](model[0](x)))                   # step-1 graph (model[2] skipped)
    (((out - y) ** 2).sum() / (8 * 4)).backward()
    tol = 2e-2 if comm_bf16 else 1e-6
    for n, p in model.named_parameters():
        ref = p.grad if p.grad is not None else torch.zeros_like(p)
        for r in range(world):
            got = torch.from_numpy(res[r][n])
            assert (got - ref).abs().max().item() <= tol * (ref.abs().max().item() + 1e-6) + 1e-7, (n, r)
    for n in res[0]:
        assert (res[0][n] == res[1][n]).all()            # replicas stay bit-identical


def _group_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aptai_amd.dp import GradGroupReducer
    red = GradGroupReducer(comm_dtype=None)
    g = torch.Generator().manual_seed(100 + rank)
    base = torch.randn(6, 5, generator=g)
    groups = {"a": [torch.randn(4, 3, generator=g), torch.randn(7, generator=g)], "b": [base]}
    for step in range(2):                                    # flat buffers are reused across steps
        for k, ts in groups.items():
            red.launch(k, ts)
        red.finish()
    q.put((rank, {k: [t.numpy().copy() for t in ts] for k, ts in groups.items()}))
    dist.destroy_process_group()


def test_group_reducer_averages_explicit_tensor_groups():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_group_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # two averaging rounds of values that are already equal after the first: result = mean of the two ranks' tensors
    import numpy as np
    ref = {}
    for rank in range(world):
        g = torch.Generator().manual_seed(100 + rank)
        base = torch.randn(6, 5, generator=g)
        ref[rank] = {"a": [torch.randn(4, 3, generator=g), torch.randn(7, generator=g)], "b": [base]}
    for k in ("a", "b"):
        for j in range(len(ref[0][k])):
            want = ((ref[0][k][j] + ref[1][k][j]) / 2).numpy()
            for r in range(world):
                np.testing.assert_allclose(res[r][k][j], want, rtol=1e-6, atol=1e-7)


def test_shard_batch_is_contiguous():
    from aptai_amd.dp import shard_batch
    b = {"a": torch.arange(8), "b": torch.arange(16).view(8, 2)}
    assert shard_batch(b, 1, 4)["a"].tolist() == [2, 3] and shard_batch(b, 3, 4)["b"].tolist() == [[12, 13], [14, 15]]
    with pytest.raises(ValueError):
        shard_batch(b, 0, 3)                                # 8 utterances over 3 ranks: refuse, do not drop two


def test_second_backward_before_finish_is_refused():
    """GradBucketReducer's protocol is one backward per finish(): a second one must not silently reduce stale buckets."""
    from aptai_amd.dp import GradBucketReducer
    lin = torch.nn.Linear(4, 4)
    red = GradBucketReducer(lin.parameters(), bucket_mb=1.0)
    red.world = 2                                           # exercise the hook bookkeeping without a process group
    red._launch = lambda bi: None
    for p in red.params:
        red._on_grad(p)
    with pytest.raises(RuntimeError):
        red._on_grad(red.params[0])


def _norm_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aptai_amd.dp import GlobalLossNorm
    norm = GlobalLossNorm()
    out = []
    for step in range(2):                                       # the buffer is reused step after step
        tv = torch.full((2, 10, 3), -100.0)
        tv[:, :4 + 2 * rank + step] = 0.5                       # 2 * (4 + 2 rank + step) * 3 valid elements
        phn = torch.zeros(2, 10, dtype=torch.long)
        phn[:, :3 + rank] = 7                                   # 2 * (3 + rank) valid frames
        norm.begin(tv, phn)
        out.append(norm.scalars().clone().numpy())
    q.put((rank, out))
    dist.destroy_process_group()


def test_global_loss_norm_counts_are_global_over_world():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_norm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step in range(2):
        n_tv = sum(2 * (4 + 2 * r + step) * 3 for r in range(world)) / world
        n_ph = sum(2 * (3 + r) for r in range(world)) / world
        for r in range(world):
            assert res[r][step][3] == n_tv and res[r][step][4] == n_ph, (r, step, res[r][step])


# ---------------------------------------------------------------------------------------------------------------- world_size 4
def _bf16_ring_avg(world):
    """Pessimistic stand-in for what RCCL's ReduceOp.AVG does to a bf16 bucket: the W contributions are added ONE AFTER THE OTHER in
    bf16 (one rounding per addition, ring order starting at rank 0), then divided by W in bf16.  (DESIGN section 6 "known and accepted":
    the product's default bucket average with comm_dtype = bf16 accumulates inside the collective.)"""
    import aptai_amd.dp as dp
    real_all_reduce = dist.all_reduce
    dp._backend_has_avg = lambda group: True

    def all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if op != dist.ReduceOp.AVG:
            return real_all_reduce(tensor, op=op, group=group, async_op=async_op)
        parts = [torch.empty_like(tensor) for _ in range(world)]
        dist.all_gather(parts, tensor, group=group)
        acc = parts[0].clone()
        for p in parts[1:]:
            acc = (acc + p)                                   # bf16 + bf16 -> bf16: rounds every time
        tensor.copy_(acc / world)

        class _Done:
            def wait(self):
                return True
        return _Done() if async_op else None
    dist.all_reduce = all_reduce


def _w4_worker(rank, world, port, q, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["APTAI_DP_ALGO"] = "a2a" if path == "a2a" else "allreduce"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aptai_amd.dp import GradBucketReducer, shard_batch
    if path == "bf16_avg":
        _bf16_ring_avg(world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.Tanh(), torch.nn.Linear(128, 128), torch.nn.Tanh(), torch.nn.Linear(128, 8))
    g = torch.Generator().manual_seed(1)
    batch = {"x": torch.randn(32, 64, generator=g), "y": torch.randn(32, 8, generator=g)}
    mine = shard_batch(batch, rank, world)
    red = GradBucketReducer(model.parameters(), bucket_mb=0.012, comm_dtype=torch.bfloat16)
    assert len(red.buckets) >= 2
    out = model(mine["x"])
    (((out - mine["y"]) ** 2).sum() / (32 * 8) * world).backward()
    red.finish()
    q.put((rank, {n: p.grad.numpy().copy() for n, p in model.named_parameters()}))
    dist.destroy_process_group()


def test_bf16_avg_gradients_are_bounded_against_the_fp32_sum_path_at_world_size_4():
    """Round-3 review, item 8 / weak 15: with the bf16 communication dtype the default bucket average (`ReduceOp.AVG` under RCCL) adds the
    W contributions in bf16 inside the collective, while the direct all-to-all pattern (`APTAI_DP_ALGO=a2a`) sums them in fp32.  World
    size 4 over gloo, the RCCL call emulated by a sequential bf16 ring sum (the pessimistic order): per-parameter rel-L2 error of both
    paths against the exact single-process gradient.  Numbers, not adjectives: a2a stays at the bf16 rounding of its inputs and output
    (~2^-9), in-collective bf16 accumulation within 2^-8 sqrt(W) - and every replica holds the same bits either way."""
    world = 4
    ctx = mp.get_context("spawn")
    res = {}
    for path in ("bf16_avg", "a2a"):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_w4_worker, args=(r, world, port, q, path)) for r in range(world)]
        for p in procs:
            p.start()
        res[path] = dict(q.get(timeout=180) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        for r in range(1, world):
            for n in res[path][0]:
                assert (res[path][0][n] == res[path][r][n]).all(), (path, n)          # replicas bit-identical
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.Tanh(), torch.nn.Linear(128, 128), torch.nn.Tanh(), torch.nn.Linear(128, 8))
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(32, 64, generator=g), torch.randn(32, 8, generator=g)
    (((model(x) - y) ** 2).sum() / (32 * 8)).backward()
    worst = {"bf16_avg": 0.0, "a2a": 0.0}
    for n, p in model.named_parameters():
        ref = p.grad.double()
        for path in worst:
            got = torch.from_numpy(res[path][0][n]).double()
            rel = ((got - ref).norm() / ref.norm()).item()
            worst[path] = max(worst[path], rel)
    print(f"[bands] W = 4, bf16 buckets: worst per-parameter rel-L2 vs the exact gradient: in-collective bf16 sum {worst['bf16_avg']:.2e}, "
          f"a2a (fp32 sum) {worst['a2a']:.2e}")
    assert worst["a2a"] <= 2.0 ** -8                                   # one rounding of each input, one of the result
    assert worst["bf16_avg"] <= 2.0 ** -8 * world ** 0.5 * 1.5         # W - 1 further roundings inside the sum
    assert worst["a2a"] <= worst["bf16_avg"] * 1.05


def test_collective_plan_describes_the_gradient_exchange():
    """dp.collective_plan: what bench.py writes into its JSON line (also as a dry plan on one GPU) - per-group payload bytes in launch
    order and per-link xGMI times for the ring and the direct pattern (2 (W - 1) / W vs 2 / W of the payload per link)."""
    from aptai_amd.dp import GradBucketReducer, collective_plan
    m = torch.nn.Sequential(torch.nn.Linear(1000, 1000), torch.nn.Linear(1000, 10))
    red = GradBucketReducer(m.parameters(), bucket_mb=1.0, comm_dtype=torch.bfloat16)
    plan = collective_plan(red.plan_groups(), 8)
    assert plan["world"] == 8 and plan["comm_dtype"] == "bfloat16" and plan["collectives_per_step"] == len(red.buckets) + 1
    assert plan["gradient_bytes_per_step"] == sum(g["bytes"] for g in plan["groups"]) >= 2 * (1000 * 1000 + 1000 + 10 * 1000 + 10)
    g0 = plan["groups"][0]
    assert abs(g0["ring_us"] / g0["direct_us"] - 7.0) < 0.1          # (W - 1) x: the ring drives one link, the direct pattern seven
    assert collective_plan(red.plan_groups(), 1)["ring_us_total"] == 0.0
