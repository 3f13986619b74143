"""Adam for the train step (train/train_aptai.py:350-356: betas, eps, weight_decay; no amsgrad) as ONE hand-written
multi-tensor kernel per parameter group (csrc/optim.hip) — a drop-in for ``torch.optim.Adam(model.parameters(), ...)`` with the
same constructor arguments, ``param_groups`` (LambdaLR works unchanged) and ``state`` keys (``step``, ``exp_avg``,
``exp_avg_sq``), so optimizer checkpoints interchange with torch's.

It can also refresh the model's compute copies in the same pass (``publish_to(model)``): the updated fp32 parameter is written
back AND converted into the persistent bf16 buffer the GEMMs read, which removes the separate per-step cast launch.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._copies: Dict[int, tuple] = {}         # id(param) -> (destination tensor, kind)
        self._tables: Dict[int, tuple] = {}         # group index -> (key, table tensor, max_n)
        self._plans = []

    # ------------------------------------------------------------------ compute copies
    def publish_to(self, model) -> "Adam":
        """Refresh the wav2vec2 layer compute copies (ops.CastPlan destinations) inside the optimiser kernel."""
        w2v = getattr(model, "wav2vec2", None) or getattr(getattr(model, "w2v2_pr", None), "wav2vec2", None) or model
        plan = w2v._layer_plan()
        by_ptr = {p.data_ptr(): p for g in self.param_groups for p in g["params"]}
        for src, dst in plan.jobs:
            p = by_ptr.get(src.data_ptr())
            if p is not None and p.numel() == dst.numel():
                self._copies[id(p)] = (dst, 0 if dst.dtype == torch.bfloat16 else 1)
        self._plans.append(plan)
        self._tables.clear()
        return self

    # ------------------------------------------------------------------ step
    def _group_tables(self, gi: int, group):
        """Static job table of a parameter group (all its parameters, whether or not they get a gradient this step) and a
        small ring of pinned host buffers for the per-step columns {grad pointer, step count} (two planes: prepare())."""
        key = tuple(p.data_ptr() for p in group["params"])
        cached = self._tables.get(gi)
        if cached is not None and cached["key"] == key:
            return cached
        # parameters the kernel cannot update (e.g. the float64 low-pass taps, which are never trained) stay out of the table;
        # step() raises if one of them ever shows up with a gradient
        params = [p for p in group["params"] if p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()]
        others = [p for p in group["params"] if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous())]
        rows = []
        for p in params:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = 0                        # python int here; state_dict() emits torch's tensor form
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            dst, kind = self._copies.get(id(p), (None, 0))
            rows.append([p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), 0 if dst is None else dst.data_ptr(),
                         p.numel(), kind])
        dev = params[0].device
        cached = dict(key=key, table=torch.tensor(rows, dtype=torch.int64).to(dev), max_n=max(r[4] for r in rows),
                      numels=[r[4] for r in rows], dyn_dev=torch.zeros((2, len(params), 2), dtype=torch.int64, device=dev),
                      ring=[torch.zeros((2, len(params), 2), dtype=torch.int64).pin_memory() for _ in range(4)],
                      events=[None] * 4, turn=0, states=[self.state[p] for p in params], params=params, others=others)
        self._tables[gi] = cached
        return cached

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.prepare()
        self.finish()
        return loss

    # ------------------------------------------------------------------ the step in pieces (aptai_amd.graphed: optimiser under the backward pass)
    @torch.no_grad()
    def prepare(self, early=()) -> None:
        """Host half of step(): step counts and the per-step {gradient pointer, step} column of every group, copied to the device
        on the current stream.  `early` names parameters whose update will be launched by launch_early() (once their gradient is
        final, on any stream ordered behind this call) before finish() updates the rest.  Every parameter's `.grad` must already
        be the tensor its gradient WILL be in (static gradient buffers): the kernel reads it only when it is launched."""
        stream = torch.cuda.current_stream()
        early_ids = {id(p) for p in early}
        self._pending = []
        for gi, group in enumerate(self.param_groups):
            if not any(p.is_cuda for p in group["params"]):
                if any(p.grad is not None for p in group["params"]):
                    raise _lib.AptaiHipError("aptai_amd.optim.Adam needs parameters on the MI355X (no CPU fallback)")
                continue
            t = self._group_tables(gi, group)
            params = t["params"]
            if any(p.grad is not None for p in t["others"]):
                raise _lib.AptaiHipError("aptai_amd.optim.Adam updates contiguous fp32 parameters on the MI355X only")
            slot = t["turn"]
            t["turn"] = (slot + 1) % len(t["ring"])
            if t["events"][slot] is not None:
                t["events"][slot].synchronize()        # the copy that last used this pinned buffer has been consumed
            host = t["ring"][slot].numpy()             # [2][n][2]: plane 0 = rows finish() updates, plane 1 = rows launch_early() updates
            host[:, :, 0] = 0
            any_late = any_early = False
            for j, (p, st) in enumerate(zip(params, t["states"])):
                g = p.grad
                if g is None:
                    continue
                if g.dtype != torch.float32 or not g.is_contiguous():
                    raise _lib.AptaiHipError("aptai_amd.optim.Adam needs contiguous fp32 gradients")
                st["step"] += 1
                plane = 1 if id(p) in early_ids else 0
                host[plane, j, 0] = g.data_ptr()
                host[plane, j, 1] = st["step"]
                any_early |= plane == 1
                any_late |= plane == 0
            if not (any_late or any_early):
                continue
            t["dyn_dev"].copy_(t["ring"][slot], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
            t["events"][slot] = ev
            self._pending.append((t, group, any_late))

    def _launch(self, t, group, plane: int, r0: int, r1: int, stream) -> None:
        b1, b2 = group["betas"]
        _lib.call("aptai_adam_multi", t["table"][r0].data_ptr(), t["dyn_dev"][plane, r0].data_ptr(), r1 - r0, max(t["numels"][r0:r1]),
                  float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), stream.cuda_stream)

    @torch.no_grad()
    def launch_early(self, params, stream=None) -> None:
        """Update `params` (named in prepare(early=...)) now, on `stream`: one launch per parameter group over the row range that
        spans them (rows in between that were not named `early` carry no gradient pointer in this plane and are skipped)."""
        stream = stream or torch.cuda.current_stream()
        ids = {id(p) for p in params}
        for t, group, _ in self._pending:
            rows = [j for j, p in enumerate(t["params"]) if id(p) in ids]
            if rows:
                self._launch(t, group, 1, min(rows), max(rows) + 1, stream)

    @torch.no_grad()
    def finish(self, stream=None) -> None:
        """Update every parameter that has a gradient and was not named `early`."""
        stream = stream or torch.cuda.current_stream()
        for t, group, any_late in self._pending:
            if any_late:
                self._launch(t, group, 0, 0, len(t["params"]), stream)
        self._pending = []
        for plan in self._plans:                    # the copies of every parameter that had a gradient are fresh now;
            plan.optimizer_synced = True            # parameters without one did not move

    @property
    def publishes_copies(self) -> bool:
        return bool(self._plans)

    # ------------------------------------------------------------------ checkpoints interchange with torch.optim.Adam
    def state_dict(self):
        sd = super().state_dict()
        for st in sd["state"].values():
            if "step" in st and not torch.is_tensor(st["step"]):
                st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if torch.is_tensor(st.get("step")):
                st["step"] = int(st["step"].item())
        self._tables.clear()                        # the moment buffers were replaced: rebuild the job tables
