"""Pure data parallelism for the APTAI hot path: one process per GPU, utterances sharded across ranks, one
bucketed gradient all-reduce per step over RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm), overlapped
with the rest of backward.  The reference is single-device: this is new functionality whose
parity definition is "averaged DP gradients == single-process gradients of the concatenated batch".

Design for xGMI (7 point-to-point links x ~153 GB/s per GPU, no switch): few, large buckets (default 48 MB
bf16 => ~13 all-reduces for wav2vec2-large) so each collective is bandwidth- not latency-bound; buckets are
filled in reverse parameter order (the order backward produces gradients) and launched from
post-accumulate-grad hooks on a side stream so the collective of bucket k overlaps the backward of bucket k+1.
Works unchanged on CPU tensors with the gloo backend (used by the world_size-2 tests).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

# How one flat gradient bucket is averaged over the ranks (APTAI_DP_ALGO, or the `algo=` argument of the reducers):
#   "allreduce" : one dist.all_reduce (RCCL picks ring / tree and its channel count)
#   "rs_ag"     : dist.reduce_scatter_tensor + dist.all_gather_into_tensor on the same flat buffer: each rank
#                 owns 1/W of the bucket, so the two halves can be pipelined bucket against bucket by RCCL
#   "a2a"       : the DIRECT pattern written out - all_to_all_single hands chunk j of every rank to rank j over the pairwise xGMI
#                 link (all 7 links of a GPU busy at once, each with 1/W of the bucket), fp32 sum of the W received chunks,
#                 all_gather_into_tensor of the reduced shards
# All three produce the same averaged bucket (tests/test_cpu_dp.py runs each at world_size 2 over gloo).
DP_ALGOS = ("allreduce", "rs_ag", "a2a")


def default_algo() -> str:
    a = os.environ.get("APTAI_DP_ALGO", "allreduce")
    if a not in DP_ALGOS:
        raise ValueError(f"APTAI_DP_ALGO={a!r}: expected one of {DP_ALGOS}")
    return a


def _backend_has_avg(group) -> bool:
    try:
        return dist.get_backend(group) == "nccl"
    except Exception:                   # noqa: BLE001
        return False


class _BucketWork:
    """Handle of one in-flight bucket average; wait() leaves the AVERAGE in `flat[:n]` (on the side stream for device buffers)."""

    # staging buffers of the rs_ag / a2a forms, one set per flat bucket (keyed by its storage): allocated on first use and reused
    # every step - a bucket is never in flight twice, and nothing is allocated on the side stream inside the step
    _STAGE = {}

    def __init__(self, flat, n, world, group, algo):
        self.flat, self.n, self.world, self.group, self.algo = flat, n, world, group, algo
        self.scale = 1.0 / world
        self.handles = []
        self.stage = None

    def _staging(self, kind, numel, dtype):
        key = (kind, self.flat.data_ptr(), self.flat.numel(), str(dtype))
        t = _BucketWork._STAGE.get(key)
        if t is None or t.numel() != numel:
            t = torch.empty(numel, device=self.flat.device, dtype=dtype)
            _BucketWork._STAGE[key] = t
        return t

    def start(self):
        flat, W, g = self.flat, self.world, self.group
        if self.algo == "allreduce":
            if _backend_has_avg(g):
                self.handles.append(dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=g, async_op=True))
                self.scale = None
            else:
                self.handles.append(dist.all_reduce(flat, group=g, async_op=True))
        elif self.algo == "rs_ag":
            shard = self._staging("shard", flat.numel() // W, flat.dtype)
            if _backend_has_avg(g):
                dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.AVG, group=g)
                self.scale = None
            else:                       # gloo has no reduce_scatter: the rehearsal sums the whole bucket and keeps its shard
                tmp = flat.clone()
                dist.all_reduce(tmp, group=g)
                r = dist.get_rank(g)
                shard.copy_(tmp.view(W, -1)[r])
            self.handles.append(dist.all_gather_into_tensor(flat, shard, group=g, async_op=True))
            self.stage = shard
        else:                           # "a2a"
            recv = self._staging("recv", flat.numel(), flat.dtype)
            dist.all_to_all_single(recv, flat, group=g)
            # sum of the W received chunks in fp32 (a bf16 running sum would round W - 1 times), averaged, rounded once
            acc = self._staging("acc32", flat.numel() // W, torch.float32)
            torch.sum(recv.view(W, -1), dim=0, dtype=torch.float32, out=acc)
            shard = self._staging("shard", flat.numel() // W, flat.dtype)
            shard.copy_(acc.mul_(1.0 / W))
            self.scale = None
            self.handles.append(dist.all_gather_into_tensor(flat, shard, group=g, async_op=True))
            self.stage = (recv, shard)
        return self

    def wait(self):
        for h in self.handles:
            h.wait()
        self.handles = []


def _padded(n: int, world: int) -> int:
    return (n + world - 1) // world * world


class GradBucketReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 48.0,
                 comm_dtype: Optional[torch.dtype] = None, process_group=None, algo: Optional[str] = None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.comm_dtype = comm_dtype
        self.algo = algo or default_algo()
        cap = int(bucket_mb * 1024 * 1024)
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, cur_bytes = [], 0
        esize = torch.empty(0, dtype=comm_dtype).element_size() if comm_dtype else 4
        for p in reversed(self.params):
            cur.append(p)
            cur_bytes += p.numel() * esize
            if cur_bytes >= cap:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): bi for bi, b in enumerate(self.buckets) for p in b}
        self._flat: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._pending = [0] * len(self.buckets)
        self._handles = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._had_grad = [None] * len(self.buckets)
        self._hooks = []
        self._stream = None
        if self.world > 1:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.reset()

    # ------------------------------------------------------------------ per-step protocol
    def reset(self):
        for bi, b in enumerate(self.buckets):
            self._pending[bi] = len(b)
            self._handles[bi] = None
            self._launched[bi] = False

    def _on_grad(self, p):
        bi = self._bucket_of[id(p)]
        self._pending[bi] -= 1
        # protocol: ONE backward pass per finish().  A second backward before finish() (gradient accumulation) would find the
        # bucket already in flight without its second contribution - refuse instead of reducing stale sums.
        if self._pending[bi] < 0:
            raise RuntimeError("GradBucketReducer: a parameter received a second gradient before finish(); call finish() "
                               "after every backward (gradient accumulation across backwards is not supported)")
        if self._pending[bi] == 0:
            self._launch(bi)

    def _flat_for(self, bi):
        b = self.buckets[bi]
        n = _padded(sum(p.numel() for p in b), self.world)
        dt = self.comm_dtype or b[0].dtype
        f = self._flat[bi]
        if f is None or f.numel() != n or f.device != b[0].device:
            f = torch.zeros(n, device=b[0].device, dtype=dt)
            self._flat[bi] = f
        return f

    def _launch(self, bi):
        b = self.buckets[bi]
        flat = self._flat_for(bi)
        views, off = [], 0
        for p in b:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        had = [p.grad is not None for p in b]
        self._had_grad[bi] = had
        src = [p.grad for p, h in zip(b, had) if h]
        dst = [v for v, h in zip(views, had) if h]
        empty = [v for v, h in zip(views, had) if not h]

        def fill_and_start():
            # parameters without a gradient this step (LayerDrop) contribute zeros, written straight into their slots of the
            # flat buffer ON the communication stream (no temporaries whose memory another stream could recycle)
            if empty:
                torch._foreach_zero_(empty)
            if dst:
                torch._foreach_copy_(dst, src)
            self._handles[bi] = _BucketWork(flat, off, self.world, self.group, self.algo).start()
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                fill_and_start()
        else:
            fill_and_start()
        self._launched[bi] = True

    def finish(self):
        """Call after ``loss.backward()``: launches incomplete buckets (parameters that received no gradient this
        step, e.g. LayerDrop'd layers, contribute zeros), waits, averages and writes the result back to ``.grad``."""
        if self.world == 1:
            return
        for bi in range(len(self.buckets)):
            if not self._launched[bi]:
                self._launch(bi)
        for bi, b in enumerate(self.buckets):
            flat = self._flat[bi]
            work = self._handles[bi]
            if flat.is_cuda:
                with torch.cuda.stream(self._stream):
                    work.wait()
                torch.cuda.current_stream().wait_stream(self._stream)
            else:
                work.wait()
            inv = work.scale
            off = 0
            outs, srcs = [], []
            for p, had in zip(b, self._had_grad[bi]):
                # a parameter without a gradient on this rank has none on any rank (LayerDrop coins are shared): keep
                # it None so the optimiser skips it exactly like the single-process reference
                if had:
                    outs.append(p.grad)
                    srcs.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            if outs:
                torch._foreach_copy_(outs, srcs)
                if inv is not None:
                    torch._foreach_mul_(outs, inv)
        self.reset()

    def reduce_existing_grads(self):
        """For manually driven backward passes (aptai_amd.graphed): all-reduce whatever is in ``.grad`` right now."""
        if self.world == 1:
            return
        self.reset()
        for bi in range(len(self.buckets)):
            self._launch(bi)
        self.finish()

    def plan_groups(self):
        """[(name, elements)] of the buckets in launch order (reverse parameter order): input of collective_plan."""
        return [(f"bucket{bi}", sum(p.numel() for p in b)) for bi, b in enumerate(self.buckets)]

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


class GradGroupReducer:
    """All-reduce (average) of explicit groups of gradient tensors, for manually driven backward passes
    (aptai_amd.graphed): `launch(key, tensors)` right after the graph segment that produced them — the collective runs
    on a side stream under the next segment's kernels — and `finish()` before the optimiser step.  One flat
    communication buffer per key (bf16 by default: a 12-layer base model is 12 x 14 MB + front/heads), so a layer is one
    collective of a size xGMI moves in ~0.2 ms while the next layer's backward takes ~0.6 ms."""

    def __init__(self, comm_dtype: Optional[torch.dtype] = torch.bfloat16, process_group=None, algo: Optional[str] = None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.comm_dtype = comm_dtype
        self.algo = algo or default_algo()
        self._flat = {}
        self._pending = []
        self._stream = None

    def launch(self, key, tensors: List[torch.Tensor]) -> None:
        if self.world == 1 or not tensors:
            return
        n = _padded(sum(t.numel() for t in tensors), self.world)
        dt = self.comm_dtype or tensors[0].dtype
        flat = self._flat.get(key)
        if flat is None or flat.numel() != n or flat.device != tensors[0].device or flat.dtype != dt:
            flat = torch.zeros(n, device=tensors[0].device, dtype=dt)
            self._flat[key] = flat
        views, off = [], 0
        for t in tensors:
            views.append(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                torch._foreach_copy_(views, tensors)
                h = _BucketWork(flat, off, self.world, self.group, self.algo).start()
        else:
            torch._foreach_copy_(views, tensors)
            h = _BucketWork(flat, off, self.world, self.group, self.algo).start()
        self._pending.append((h, views, tensors))

    def finish(self) -> None:
        for h, views, tensors in self._pending:
            if views[0].is_cuda:
                with torch.cuda.stream(self._stream):
                    h.wait()
                torch.cuda.current_stream().wait_stream(self._stream)
            else:
                h.wait()
            torch._foreach_copy_(tensors, views)
            if h.scale is not None:
                torch._foreach_mul_(tensors, h.scale)
        self._pending = []


class GlobalLossNorm:
    """Masked-mean losses over the GLOBAL batch under data parallelism ("loss-mean subtlety").

    models/aptai.py:89-100 divides the squared error by the number of valid TV elements of the batch and the cross-entropy
    by the number of valid frames.  With utterances sharded over W ranks each rank sees only its own counts n_r, and the
    average of per-rank means weights a frame of a short shard more than a frame of a long one.  This object makes the
    averaged DP gradient equal the single-process gradient at the global batch: the backward normalises by n_global / W
    instead of n_r (so the all-reduce AVERAGE of the W shard gradients is sum_r (shard sums) / n_global).  One 2-scalar
    all-reduce per step, launched as soon as the targets are on the device and waited for just before the loss backward.

    `begin(tv_tgt, phn_tgt)` -> launches; `scalars()` -> float32[5] device buffer in the layout of the loss kernels'
    `scalars` (only [3] = TV elements and [4] = frames are read by aptai_aptai_loss_bwd)."""

    def __init__(self, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._buf = None
        self._work = None
        self._stream = None

    def buffer(self, device) -> torch.Tensor:
        if self._buf is None or self._buf.device != torch.device(device):
            self._buf = torch.ones(5, device=device, dtype=torch.float32)
        return self._buf

    def begin(self, tv_tgt: torch.Tensor, phn_tgt: Optional[torch.Tensor]) -> None:
        buf = self.buffer(tv_tgt.device)
        cur = torch.cuda.current_stream() if buf.is_cuda else None
        if buf.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            self._stream.wait_stream(cur)
        ctx = torch.cuda.stream(self._stream) if buf.is_cuda else _Null()
        with ctx:
            buf[3] = (tv_tgt != -100.0).sum()
            buf[4] = (phn_tgt != 0).sum() if phn_tgt is not None else 1.0
            if self.world > 1:
                self._work = dist.all_reduce(buf, group=self.group, async_op=True)

    def scalars(self) -> torch.Tensor:
        """Waits for the counts and returns the buffer (n_global / W in [3], [4])."""
        if self._work is not None:
            # wait() orders the stream that is current when it is called behind the collective (NCCL/RCCL run it on their
            # own stream): call it on the side stream, scale there, then let the caller's stream wait for the side stream
            if self._buf.is_cuda:
                with torch.cuda.stream(self._stream):
                    self._work.wait()
                    self._buf.mul_(1.0 / self.world)
            else:
                self._work.wait()
                self._buf.mul_(1.0 / self.world)
            self._work = None
        if self._buf.is_cuda:
            torch.cuda.current_stream().wait_stream(self._stream)
        return self._buf


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def collective_plan(groups, world: int, comm_dtype: Optional[torch.dtype] = torch.bfloat16, algo: Optional[str] = None) -> dict:
    """The gradient exchange of ONE train step as data (bench.py prints it into its JSON line, also as a dry plan on one GPU, so that the
    first record from a real 8-GPU node describes itself): `groups` = [(name, number of gradient elements)] in launch order - the
    48 MB buckets of GradBucketReducer or the per-segment groups of GradGroupReducer.  Per collective: payload bytes, and the time its
    slowest xGMI link needs at 153 GB/s per direction (7 point-to-point links per GPU) for a ring (2 (W - 1) / W of the
    payload over ONE link) and for the direct reduce-scatter + all-gather pattern (2 / W of the payload over EACH of the W - 1 links)."""
    algo = algo or default_algo()
    esize = torch.empty(0, dtype=comm_dtype).element_size() if comm_dtype else 4
    link = 153e9
    items, total = [], 0
    for name, numel in groups:
        b = _padded(int(numel), max(world, 1)) * esize
        total += b
        items.append({"group": str(name), "bytes": b,
                      "ring_us": round(2.0 * (world - 1) / world * b / link * 1e6, 1) if world > 1 else 0.0,
                      "direct_us": round(2.0 / world * b / link * 1e6, 1) if world > 1 else 0.0})
    return {"world": world, "algo": algo, "comm_dtype": str(comm_dtype).replace("torch.", "") if comm_dtype else "float32",
            "collectives_per_step": len(items) + 1, "gradient_bytes_per_step": total,
            "loss_norm": "one 5-float all-reduce at the top of the step (global valid counts, dp.GlobalLossNorm)",
            "ring_us_total": round(sum(i["ring_us"] for i in items), 1), "direct_us_total": round(sum(i["direct_us"] for i in items), 1),
            "groups": items}


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """Contiguous utterance shards of a global batch dict."""
    out = {}
    for k, v in batch.items():
        n = v.shape[0]
        if n % world:
            raise ValueError(f"shard_batch: {k!r} holds {n} utterances, not a multiple of world_size {world} "
                             f"(the remainder would be dropped silently)")
        per = n // world
        out[k] = v[rank * per:(rank + 1) * per]
    return out
