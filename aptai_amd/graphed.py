"""hipGraph-captured APTAI train step (HIP streams + graphs instead of a tracing compiler).

The eager drop-in loop (``model(epoch, **batch)``; ``loss.backward()``; ``optimizer.step()``) issues ~340 kernel
launches per step from Python and autograd callbacks; that, not the GPU, sets its step time (13.7 ms against 10.4 ms of kernels).  ``GraphedAPTAIStep`` captures
the same kernels, through the same C ABI and the same fwd/bwd implementations the autograd path uses, into SEGMENT
graphs:

    prep (bf16 weight copies) | conv stack + encoder front | layer i forward ... | heads fwd+bwd | layer i backward ... | front bwd

and replays them (~30 ``hipGraphLaunch`` per step).  Host-side randomness keeps the reference's semantics:
 * LayerDrop (HF:701-703 / 774-776): a dropped layer's two graphs are simply not replayed (its activations / gradients are
   forwarded by one device copy) and its parameters get ``grad = None`` that step, exactly like eager;
 * SpecAugment (HF:101-217): the span mask is sampled by a kernel at the head of the front segment (ops.spec_augment_mask:
   HF's span-count rule, fresh spans on every replay through the salted seed) - no host copy, no synchronisation;
 * dropout: kernels XOR a per-step device salt into their counter-RNG seeds (``aptai_set_seed_salt``), so every replay
   draws fresh masks while the forward and backward of one step regenerate identical ones.
The optimiser step is issued eagerly after the last segment (aptai_amd.optim.Adam: one launch per parameter group; or any
torch optimiser).  Nothing in a step synchronises host and device.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib, hostlogic, ops
from .aptai import APTAI, heads_bwd, heads_fwd
from .w2v2_pr import Wav2Vec2_PR, pr_head_bwd, pr_head_fwd
from .wav2vec2 import _FinalLNImpl, _FrontImpl, _LayerImpl, _seed


# thread-local capture: under data parallelism the process group's watchdog thread polls its events while this thread captures,
# and in the default "global" mode a call from ANY thread can invalidate the capture
_CAPTURE_MODE = "thread_local"

# stream handle -> the device tensor whose two words that stream's kernels XOR into their dropout seeds (kept alive here for
# as long as the library holds the pointer)
_SALTS: Dict[int, torch.Tensor] = {}


def _bind_salt(stream_handle: int, salt: torch.Tensor) -> None:
    _SALTS[stream_handle] = salt
    _lib.call("aptai_set_seed_salt", stream_handle, salt.data_ptr())


def _unbind_salt(stream_handle: int) -> None:
    _lib.call("aptai_set_seed_salt", stream_handle, None)
    _lib.call("aptai_set_frame_bounds", stream_handle, None)
    _SALTS.pop(stream_handle, None)
    _BOUNDS.pop(stream_handle, None)


# stream handle -> the device tensor with that stream's frame bounds {conv0 GroupNorm frame count, FIR frame bound}
_BOUNDS: Dict[int, torch.Tensor] = {}


def _bind_bounds(stream_handle: int, bounds: torch.Tensor) -> None:
    _BOUNDS[stream_handle] = bounds
    _lib.call("aptai_set_frame_bounds", stream_handle, bounds.data_ptr())


class GraphedAPTAIStep:
    def __init__(self, model, optimizer: torch.optim.Optimizer, batch: Dict[str, torch.Tensor], reducer=None):
        assert model.training, "call model.train() first"
        self.model, self.opt, self.reducer = model, optimizer, reducer
        # data parallel: gradients are averaged per graph segment (dp.GradGroupReducer), overlapped with the next segment
        self.group_reducer = None
        if reducer is not None and getattr(reducer, "world", 1) > 1:
            from .dp import GradGroupReducer
            self.group_reducer = GradGroupReducer(comm_dtype=reducer.comm_dtype, process_group=reducer.group)
        w = model.wav2vec2
        self.w, self.cfg = w, w.config
        cfg = self.cfg
        dev = next(model.parameters()).device
        self.dev = dev
        # two model kinds share the segments: APTAI (regression + frame classification heads, models/aptai.py) and Wav2Vec2_PR
        # (CTC head, models/w2v2_pr.py).  A trainable feature encoder (the recogniser's fine-tuning,
        # train/train_phoneme_recognizer.py) adds its backward to the front-backward segment.
        self.kind = "pr" if isinstance(model, Wav2Vec2_PR) else "aptai"
        self.train_conv = any(p.requires_grad for p in w.feature_extractor.parameters())
        # ---- static inputs
        audio_key = "input_values" if self.kind == "pr" else "audio_inputs"
        self.audio = batch[audio_key].float().contiguous().clone()
        B, S = self.audio.shape
        self.g = w._geometry(B, S)
        g = self.g
        self.lens_i32 = torch.zeros(B, device=dev, dtype=torch.int32)
        self.spec = torch.zeros((B, g.T), device=dev, dtype=torch.uint8)
        if self.kind == "pr":
            self.labels = torch.full(tuple(batch["phoneme_labels"].shape), -100, device=dev, dtype=torch.int32)
            self.state_lens = torch.zeros(B, device=dev, dtype=torch.int32)
            self.target_lens = torch.zeros(B, device=dev, dtype=torch.int32)
        else:
            self.tv_tgt = torch.zeros((B, g.T, model.n_tv), device=dev, dtype=torch.float32)
            self.phn_tgt = torch.zeros((B, g.T), device=dev, dtype=torch.int64)
        self.salt = torch.zeros(2, device=dev, dtype=torch.int32)
        self._salt_ring = [torch.zeros(2, dtype=torch.int32).pin_memory() for _ in range(4)]
        self._salt_events = [None] * 4
        self._salt_turn = 0
        self._salt_gen = np.random.RandomState(0xC0FFEE + w.base_seed)
        # the salt is bound to THIS runner's capture stream (include/aptai_hip.h: no process-global state); the module-level
        # registry keeps the two words alive until the binding is cleared, whatever happens to the runner object
        self._cap_stream = torch.cuda.Stream(device=dev)
        _bind_salt(self._cap_stream.cuda_stream, self.salt)
        # frame bounds of the batch inside the captured shape (BucketedGraphedStep feeds shorter batches): by default the shape itself
        self.bounds = torch.tensor([g.Tl[0], g.T], device=dev, dtype=torch.int32)
        self._bounds_ring = [torch.zeros(2, dtype=torch.int32).pin_memory() for _ in range(4)]
        self._bounds_events = [None] * 4
        self._bounds_turn = 0
        self._bounds_host = (g.Tl[0], g.T)
        _bind_bounds(self._cap_stream.cuda_stream, self.bounds)
        self.set_batch(batch)
        # one eager step first: allocates every persistent scratch buffer, loads the code objects and sets the kernel
        # attributes outside of stream capture
        if getattr(w, "_cache_mode", None) == "frozen":          # another runner of this model froze the cache: the eager step rebuilds
            w._cache_mode = None
        model.zero_grad(set_to_none=True)
        (model(**batch) if self.kind == "pr" else model(0, **batch))["loss"].backward()
        model.zero_grad(set_to_none=True)
        try:
            self._capture()
        finally:
            ops.ln_defer_end()                 # whatever happened during capture, eager LayerNorm backwards reduce their own partials again

    # ------------------------------------------------------------------ inputs
    def set_batch(self, batch: Dict[str, torch.Tensor]) -> None:
        """Copies a batch (the collate_fn's dict, host or device tensors) into the static input buffers of the graphs.
        Host tensors go through a two-slot ring of pinned staging buffers and asynchronous copies on the step's stream:
        10.7 ms/step against 10.3 with the batch resident (pageable copies, which block the host until the GPU has drained:
        34 ms/step)."""
        cfg, g = self.cfg, self.g
        if self.kind == "pr":
            return self._set_batch_pr(batch)
        lens_cpu = batch["audio_lengths"].detach().cpu().long()
        fl = hostlogic.feat_extract_output_lengths(lens_cpu, cfg.conv_kernel, cfg.conv_stride).clamp(min=1, max=g.T)
        self.frame_lens_cpu = fl
        tracks = [batch[k] for k in batch if k not in ("audio_inputs", "audio_lengths", "phn_frames_49hz", "phoneme_labels")]
        if batch["audio_inputs"].device.type == "cpu":
            # staging copies through numpy: single-threaded memcpy / cast.  torch's copy_ goes parallel above 32 K elements
            # and the OpenMP workers then spin on the cores the launching thread needs (host time per step 8 -> 19 ms).
            st = self._host_stage()
            np.copyto(st["audio_np"], batch["audio_inputs"].detach().numpy(), casting="same_kind")
            np.copyto(st["lens_np"], fl.numpy(), casting="same_kind")
            for j, t in enumerate(tracks):                       # f64 (B, T) tracks -> one (B, T, n_tv) f32 block
                st["tv_np"][:, :, j] = t.detach().numpy()
            np.copyto(st["phn_np"], batch["phn_frames_49hz"].numpy(), casting="same_kind")
            for dst, key in ((self.audio, "audio"), (self.lens_i32, "lens"), (self.tv_tgt, "tv"), (self.phn_tgt, "phn")):
                dst.copy_(st[key], non_blocking=True)
            st["event"] = torch.cuda.Event()
            st["event"].record()
            return
        self.audio.copy_(batch["audio_inputs"].float())
        self.lens_i32.copy_(fl.to(torch.int32))
        self.tv_tgt.copy_(torch.stack(tracks, dim=-1).float())
        self.phn_tgt.copy_(batch["phn_frames_49hz"])

    def set_bounds(self, conv0_frames: int, frames: int) -> None:
        """Frame bounds of the NEXT step's batch inside the captured shape (see aptai_set_frame_bounds): how many first-conv-layer
        frames count for the GroupNorm statistics and where the low-pass filter's zero padding begins.  No synchronisation (the
        two words travel through a ring of pinned buffers)."""
        if (conv0_frames, frames) == self._bounds_host:
            return
        slot = self._bounds_turn
        self._bounds_turn = (slot + 1) % len(self._bounds_ring)
        if self._bounds_events[slot] is not None:
            self._bounds_events[slot].synchronize()
        self._bounds_ring[slot][0], self._bounds_ring[slot][1] = int(conv0_frames), int(frames)
        self.bounds.copy_(self._bounds_ring[slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._bounds_events[slot] = ev
        self._bounds_host = (conv0_frames, frames)

    def _set_batch_pr(self, batch: Dict[str, torch.Tensor]) -> None:
        """Wav2Vec2_PR batch (input_values, input_lengths, phoneme_labels with -100 padding, models/w2v2_pr.py:40-70).  The label
        block may be narrower than the one the graphs were captured with (padded with -100), never wider."""
        cfg, g = self.cfg, self.g
        lens_cpu = batch["input_lengths"].detach().cpu().long().reshape(-1)
        sl = hostlogic.feat_extract_output_lengths(lens_cpu, cfg.conv_kernel, cfg.conv_stride)
        self.frame_lens_cpu = sl.clamp(min=1, max=g.T)
        lab = batch["phoneme_labels"]
        if lab.shape[0] != self.labels.shape[0] or lab.shape[1] > self.labels.shape[1]:
            raise ValueError(f"label block {tuple(lab.shape)} does not fit the captured {tuple(self.labels.shape)}")
        dev = self.dev
        self.audio.copy_(batch["input_values"].float(), non_blocking=True)
        self.lens_i32.copy_(self.frame_lens_cpu.to(torch.int32), non_blocking=True)
        self.state_lens.copy_(sl.to(torch.int32), non_blocking=True)
        self.target_lens.copy_(hostlogic.ctc_target_lengths(lab).to(torch.int32), non_blocking=True)
        if lab.shape[1] < self.labels.shape[1]:
            self.labels.fill_(-100)
        self.labels[:, :lab.shape[1]].copy_(lab.to(torch.int32), non_blocking=True)

    def _host_stage(self):
        ring = getattr(self, "_stage_ring", None)
        if ring is None:
            mk = lambda ref: torch.empty(ref.shape, dtype=ref.dtype).pin_memory()
            ring = [dict(audio=mk(self.audio), lens=mk(self.lens_i32), tv=mk(self.tv_tgt), phn=mk(self.phn_tgt), event=None)
                    for _ in range(2)]
            for st in ring:
                for key in ("audio", "lens", "tv", "phn"):
                    st[key + "_np"] = st[key].numpy()            # views of the pinned buffers
            self._stage_ring, self._stage_turn = ring, 0
        st = ring[self._stage_turn]
        self._stage_turn ^= 1
        if st["event"] is not None:
            st["event"].synchronize()                            # the copies that last read this slot have finished
        return st

    def _host_randomness(self):
        cfg, g = self.cfg, self.g
        # the per-step salt travels through a small ring of pinned buffers (a pageable H2D copy would block the host until
        # the GPU has drained); the SpecAugment mask is sampled by a kernel inside the front segment from the same salt
        slot = self._salt_turn
        self._salt_turn = (slot + 1) % len(self._salt_ring)
        if self._salt_events[slot] is not None:
            self._salt_events[slot].synchronize()
        self._salt_ring[slot].copy_(torch.from_numpy(self._salt_gen.randint(-2 ** 31, 2 ** 31 - 1, size=2).astype(np.int32)))
        self.salt.copy_(self._salt_ring[slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._salt_events[slot] = ev
        keep = [True] * cfg.num_hidden_layers
        if cfg.layerdrop > 0:
            keep = [float(torch.rand([], generator=self.w._layerdrop_gen)) >= cfg.layerdrop for _ in keep]
        return keep

    # ------------------------------------------------------------------ capture
    def _capture(self):
        # fault injection for the multi-rank fallback rehearsal (bench.py, tests/test_gpu_bench.py): "1" = every rank,
        # "rank<k>" = that rank only.  Placed here, after the constructor's eager step, where a real capture failure would occur.
        inject = os.environ.get("APTAI_GRAPH_FAIL_CAPTURE")
        if inject and inject in ("1", f"rank{os.environ.get('RANK', '0')}"):
            raise RuntimeError("APTAI_GRAPH_FAIL_CAPTURE is set")
        model, w, cfg, g = self.model, self.w, self.cfg, self.g
        L = cfg.num_hidden_layers
        pool = torch.cuda.graph_pool_handle()
        seed = _seed(w.base_seed, 0xABCD)
        mk = torch.cuda.CUDAGraph
        torch.cuda.synchronize()

        # -- prep: every bf16 compute copy, rebuilt from the fp32 parameters on each replay
        w._cache.clear()
        w._cache_mode = "build"
        w._layer_plan()                                      # persistent copies + job table exist before capture
        w._cache_mode = None
        w._refresh_layer_copies(force=True)                  # copies valid before the first replay whoever refreshes them later
        w._cache_mode = "build"
        torch.cuda.synchronize()
        if not self.train_conv:
            w._conv_weights()                                # frozen feature encoder: its bf16 weight copies are built once, here
            torch.cuda.synchronize()
        self.g_prep = mk()
        with torch.cuda.graph(self.g_prep, pool=pool, stream=self._cap_stream, capture_error_mode=_CAPTURE_MODE):
            if self.train_conv:
                w._conv_weights()
            if not getattr(self.opt, "publishes_copies", False):     # else the optimiser kernel refreshes the copies itself
                w._refresh_layer_copies(force=True)
            self.lw = [w._layer_weights(i, g.M) for i in range(L)]
            fp, pc = w.feature_projection, w.encoder.pos_conv_embed.conv
            w._cached(("proj",), [fp.projection.weight], lambda: ops.cast_bf16(fp.projection.weight))
            w._cached(("posconv",), [pc.parametrizations.weight.original0, pc.parametrizations.weight.original1],
                      lambda: ops.posconv_weight(pc.parametrizations.weight.original1,
                                                 pc.parametrizations.weight.original0.reshape(-1),
                                                 cfg.num_conv_pos_embedding_groups))
        w._cache_mode = "frozen"

        # -- front: conv stack (frozen) + projection + masking + positional conv (+ LN)
        embed = getattr(w, "masked_spec_embed", None)
        self.front = _FrontImpl(cfg, g, self.lens_i32, self.spec if embed is not None else None, True, _seed(seed, 1), w)
        fp, pc = w.feature_projection, w.encoder.pos_conv_embed.conv
        self.fparams = [fp.layer_norm.weight, fp.layer_norm.bias, fp.projection.weight, fp.projection.bias, embed,
                        pc.parametrizations.weight.original0, pc.parametrizations.weight.original1, pc.bias,
                        w.encoder.layer_norm.weight, w.encoder.layer_norm.bias]
        self.g_front = mk()
        with torch.cuda.graph(self.g_front, pool=pool, stream=self._cap_stream, capture_error_mode=_CAPTURE_MODE):
            if embed is not None and cfg.apply_spec_augment and cfg.mask_time_prob > 0:
                ops.spec_augment_mask(self.lens_i32, g.B, g.T, cfg.mask_time_prob, cfg.mask_time_length, cfg.mask_time_min_masks,
                                      _seed(seed, 77), out=self.spec)        # fresh spans on every replay (salted seed)
            feats, self.sv_conv = w._conv_forward(self.audio, g, save=self.train_conv)
            (h,), self.s_front = self.front.fwd(feats, self.fparams, True)
        self.X = [h]

        # -- layers forward
        self.impl, self.s_layer, self.g_fwd, self.lparams = [], [], [], []
        for i, layer in enumerate(w.encoder.layers):
            wt, lin = self.lw[i]
            impl = _LayerImpl(cfg, g, self.lens_i32, wt, True, _seed(seed, 100 + i))
            params = [layer.layer_norm.weight, layer.layer_norm.bias, layer.final_layer_norm.weight, layer.final_layer_norm.bias] + lin
            gr = mk()
            with torch.cuda.graph(gr, pool=pool, stream=self._cap_stream, capture_error_mode=_CAPTURE_MODE):
                (y,), s = impl.fwd(self.X[i], params, True)
            self.impl.append(impl); self.s_layer.append(s); self.g_fwd.append(gr); self.lparams.append(params)
            self.X.append(y)

        # -- tail: [final LN] + heads forward + loss + heads backward + [final LN backward]
        self.fin = _FinalLNImpl(cfg, g) if cfg.do_stable_layer_norm else None
        self.loss_norm = None
        if self.kind == "pr":
            self.hparams = [model.pr_head.weight, model.pr_head.bias]
            self.st_heads = SimpleNamespace(g=g, p_final=model.dropout.p, seed=_seed(seed, 777), targets=self.labels,
                                            state_lens=self.state_lens, target_lens=self.target_lens, blank=getattr(cfg, "blank", 0),
                                            reduction=cfg.ctc_loss_reduction, zero_infinity=cfg.ctc_zero_infinity)
        else:
            self.hparams = [model.tv_head[2].weight, model.tv_head[2].bias, model.phn_head[2].weight, model.phn_head[2].bias]
            self.st_heads = SimpleNamespace(g=g, p_tv=model.tv_head[0].p, p_ph=model.phn_head[0].p, seed=_seed(seed, 999),
                                            taps=model.tv_lowpass.taps(), tv_tgt=self.tv_tgt, phn_tgt=self.phn_tgt, w_mse=0.5, w_ce=0.5)
            # data parallel: the loss backward inside the tail graph reads the global valid counts / world from this static buffer
            self.loss_norm = getattr(model, "dp_loss_norm", None)
            if self.loss_norm is not None:
                self.st_heads.norm_scalars = self.loss_norm.buffer(self.dev)
        # single GPU: the ~24 LayerNorm parameter-gradient reductions of the backward pass (one 4.8 us launch + a kernel boundary each)
        # are collected here and replayed as ONE launch behind the front-end backward; nothing reads them before the optimiser.
        # Data parallel keeps them per call: a layer's gradients leave for the all-reduce right after its segment.
        self._ln_jobs = ops.ln_defer_begin() if (self.group_reducer is None and os.environ.get("APTAI_LN_DEFER", "1") != "0") else None
        self.g_tail = mk()
        with torch.cuda.graph(self.g_tail, pool=pool, stream=self._cap_stream, capture_error_mode=_CAPTURE_MODE):
            hl = self.X[L]
            if self.fin is not None:
                (hl,), s_fin = self.fin.fwd(hl, [w.encoder.layer_norm.weight, w.encoder.layer_norm.bias], True)
            if self.kind == "pr":
                self.outs, s_heads = pr_head_fwd(hl, *self.hparams, self.st_heads)
                dh, *hgrads = pr_head_bwd(s_heads, self.st_heads, None)
            else:
                self.outs, s_heads = heads_fwd(hl, *self.hparams, self.st_heads)
                dh, *hgrads = heads_bwd(s_heads, self.st_heads, None)
            fin_grads = []
            if self.fin is not None:
                dh, fin_grads = self.fin.bwd(s_fin, (dh,), True)
        self.dX: List[Optional[torch.Tensor]] = [None] * (L + 1)
        self.dX[L] = dh
        self.grads: Dict[torch.nn.Parameter, torch.Tensor] = {}
        for p, gt in zip(self.hparams, hgrads):
            self.grads[p] = gt
        if self.fin is not None:
            self.grads[w.encoder.layer_norm.weight], self.grads[w.encoder.layer_norm.bias] = fin_grads

        # -- layers backward (reverse capture order = replay order).  APTAI_WGRAD_OVERLAP=1 (experiment, OFF by default): the layer's
        # grouped weight-gradient launch as its own graph, replayed on a side stream beside the NEXT layer's dgrad chain (nothing
        # downstream reads it before the optimiser), hoping its 486 tiles fill the launch gaps and one-round tails of the ~12
        # dependent kernels of that chain.  Measured on one box, interleaved: 9.57 / 9.57 ms inline vs 9.93 / 9.97 ms overlapped -
        # two chip-filling launches at once evict each other's operand panels from L2 and lose more than the gaps they fill.
        # (The tensors the side graph reads stay referenced in `self._wpending`, so the shared graph pool never hands their memory
        # to a later segment.)
        self.g_bwd = [None] * L
        self.g_bww = [None] * L
        self._wpending = [None] * L
        self.layer_grads = [None] * L
        self.overlap_wgrad = (os.environ.get("APTAI_WGRAD_OVERLAP", "0") != "0") and self.group_reducer is None
        self._w_stream = torch.cuda.Stream(device=self.dev) if self.overlap_wgrad else None
        for i in range(L - 1, -1, -1):
            gr = mk()
            if self.overlap_wgrad:
                with torch.cuda.graph(gr, pool=pool, stream=self._cap_stream, capture_error_mode=_CAPTURE_MODE):
                    res = self.impl[i].bwd(self.s_layer[i], (self.dX[i + 1],), True, defer_wgrad=True)
                if len(res) == 3:
                    dx, ln_grads, self._wpending[i] = res
                    gw = mk()
                    with torch.cuda.graph(gw, pool=pool, stream=self._cap_stream, capture_error_mode=_CAPTURE_MODE):
                        wg = self.impl[i].bwd_wgrad(self._wpending[i])
                    self.g_bww[i] = gw
                    pg = tuple(ln_grads) + tuple(wg)
                else:                                   # APTAI_GROUPED_WGRAD=0: nothing to defer
                    dx, pg = res
            else:
                with torch.cuda.graph(gr, pool=pool, stream=self._cap_stream, capture_error_mode=_CAPTURE_MODE):
                    dx, pg = self.impl[i].bwd(self.s_layer[i], (self.dX[i + 1],), True)
            self.g_bwd[i] = gr
            self.dX[i] = dx
            self.layer_grads[i] = list(zip(self.lparams[i], pg))

        # -- front backward
        self.g_front_bwd = mk()
        with torch.cuda.graph(self.g_front_bwd, pool=pool, stream=self._cap_stream, capture_error_mode=_CAPTURE_MODE):
            dfeats, fg = self.front.bwd(self.s_front, (self.dX[0],), self.train_conv)
            conv_grads = w._conv_backward(self.sv_conv, g, dfeats.contiguous()) if self.train_conv else []
        for p, gt in zip(self.fparams, fg):
            if p is not None and gt is not None:
                self.grads[p] = gt
        for p, gt in zip(w._conv_params(), conv_grads):
            self.grads[p] = gt
        self.g_ln = None
        if self._ln_jobs is not None:
            ops.ln_defer_end()
            if self._ln_jobs:
                torch.cuda.synchronize()
                self._ln_table, n_jobs, max_cols = ops.ln_defer_table(self._ln_jobs)        # host-to-device copy: outside capture
                torch.cuda.synchronize()
                self.g_ln = mk()
                with torch.cuda.graph(self.g_ln, pool=pool, stream=self._cap_stream, capture_error_mode=_CAPTURE_MODE):
                    ops.ln_finalize_multi(self._ln_table, n_jobs, max_cols)
        # optimiser under the backward pass (step()): the parameters of layer i whose gradients are final when its backward segment ends
        self.adam_overlap = (self.group_reducer is None and not self.overlap_wgrad and hasattr(self.opt, "launch_early")
                             and os.environ.get("APTAI_ADAM_OVERLAP", "0") != "0")
        self._o_stream = torch.cuda.Stream(device=self.dev) if self.adam_overlap else None
        self._early_params = [[p for k, (p, _) in enumerate(self.layer_grads[i]) if p.requires_grad and (k >= 4 or self.g_ln is None)]
                              for i in range(L)]
        # the graphs hold raw addresses of the compute copies in the model's cache: keep them alive for this runner's lifetime (a
        # second runner of the same model - another shape bucket - clears and rebuilds the cache for its own capture)
        self._keep_cache = dict(w._cache)
        torch.cuda.synchronize()

    def _assign_grads(self, keep) -> None:
        """Every trainable parameter's `.grad` = its static gradient buffer (None for a layer LayerDrop skipped this step)."""
        for p, gt in self.grads.items():
            if p.requires_grad:
                p.grad = gt
        for i in range(self.cfg.num_hidden_layers):
            for p, gt in self.layer_grads[i]:
                if p.requires_grad:
                    p.grad = gt if keep[i] else None
        if not getattr(self, "_checked", False):
            names = {id(p): n for n, p in self.model.named_parameters()}
            for p in self.model.parameters():
                if p.grad is not None and (p.grad.dtype != p.dtype or p.grad.shape != p.shape or not p.grad.is_contiguous()
                                           or p.grad.device != p.device):
                    raise RuntimeError(f"static gradient of {names[id(p)]} has dtype {p.grad.dtype} shape {tuple(p.grad.shape)} "
                                       f"contiguous={p.grad.is_contiguous()} (parameter: {p.dtype} {tuple(p.shape)})")
            self._checked = True

    # ------------------------------------------------------------------ one optimiser step
    def step(self, batch: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        if batch is not None:
            self.set_batch(batch)
        keep = self._host_randomness()
        L = self.cfg.num_hidden_layers
        self.w._train_marker = getattr(self.w, "_train_marker", 0) + 1     # eval-mode weight copies built before this step are stale
        if self.loss_norm is not None:
            self.loss_norm.begin(self.tv_tgt, self.phn_tgt)      # travels under the encoder forward
        # Optimiser under the backward pass (APTAI_ADAM_OVERLAP=1; experiment, OFF by default; single GPU, aptai_amd.optim.Adam): a
        # layer's six weight / bias pairs are updated on a side stream as soon as its backward segment has finished - an HBM-bound
        # launch beside the next layers' matrix-bound ones.  Adam is element-wise per parameter, so WHEN a parameter is updated
        # changes no bit (tests/test_gpu_graphed.py).  Measured, interleaved on one box: 8.97 / 8.97 / 8.97 ms without, 9.15 / 9.15 /
        # 9.21 ms with - the 0.34 ms of Adam cost 0.52 ms beside the GEMMs (2.8 GB streamed through the L2s their operand panels
        # live in).  (The layer's LayerNorm gradients are final only after the deferred finalize launch: they stay with finish().)
        overlap_opt = self.adam_overlap
        if overlap_opt:
            self._assign_grads(keep)
            early = [self._early_params[i] if keep[i] else [] for i in range(L)]
            self.opt.prepare(early=[p for lst in early for p in lst])
        self.g_prep.replay()
        self.g_front.replay()
        for i in range(L):
            if keep[i]:
                self.g_fwd[i].replay()
            else:
                self.X[i + 1].copy_(self.X[i])
        red = self.group_reducer
        if self.loss_norm is not None:
            self.loss_norm.scalars()                 # counts launched at the top of the step have arrived
        self.g_tail.replay()
        if red is not None:
            red.launch("heads", [gt for p, gt in self.grads.items() if p.requires_grad and any(p is q for q in self.hparams)])
        cur = torch.cuda.current_stream(self.dev) if (self.overlap_wgrad or overlap_opt) else None
        for i in range(L - 1, -1, -1):
            if keep[i]:
                self.g_bwd[i].replay()
                if self.g_bww[i] is not None:  # the layer's weight gradients on the side stream, beside layer i-1's dgrad chain
                    self._w_stream.wait_stream(cur)
                    with torch.cuda.stream(self._w_stream):
                        self.g_bww[i].replay()
                if overlap_opt and early[i]:
                    self._o_stream.wait_stream(cur)
                    self.opt.launch_early(early[i], self._o_stream)
                if red is not None:            # layer i's gradients travel while layer i-1's backward runs
                    red.launch(("layer", i), self._layer_grad_tensors(i))
            else:
                self.dX[i].copy_(self.dX[i + 1])
        self.g_front_bwd.replay()
        if self.g_ln is not None:
            self.g_ln.replay()                 # every LayerNorm dgamma / dbeta of the step in one launch
        if self.overlap_wgrad:
            cur.wait_stream(self._w_stream)    # every weight gradient is in place before the optimiser reads it
        if red is not None:
            red.launch("front", [gt for p, gt in self.grads.items() if p.requires_grad and not any(p is q for q in self.hparams)])
            red.finish()
        if not overlap_opt:
            self._assign_grads(keep)
        if overlap_opt:
            cur.wait_stream(self._o_stream)    # (the next step's prep segment reads what these launches wrote)
            self.opt.finish()
        else:
            self.opt.step()
        if self.kind == "pr":
            loss, logits, log_probs, hd = self.outs
            g, V = self.g, self.hparams[0].shape[0]
            return {"loss": loss, "phoneme_logits": logits.view(g.B, g.Tp, -1)[:, :g.T, :V], "log_probs": log_probs,
                    "hidden_states": hd.view(g.B, g.Tp, -1)[:, :g.T]}
        loss, mse, ce, tvs, pred, _ = self.outs
        return {"loss": loss, "mse_loss": mse, "ce_loss": ce, "tvs_pred": tvs, "phn_fc_pred": pred}

    def plan_groups(self):
        """[(name, gradient elements)] of the per-segment gradient groups in the order step() hands them to dp.GradGroupReducer
        (heads, layers L-1 .. 0, front): input of dp.collective_plan."""
        heads = [gt for p, gt in self.grads.items() if p.requires_grad and any(p is q for q in self.hparams)]
        front = [gt for p, gt in self.grads.items() if p.requires_grad and not any(p is q for q in self.hparams)]
        out = [("heads", sum(t.numel() for t in heads))]
        for i in range(self.cfg.num_hidden_layers - 1, -1, -1):
            out.append((f"layer{i}", sum(t.numel() for t in self._layer_grad_tensors(i))))
        out.append(("front", sum(t.numel() for t in front)))
        return out

    def _layer_grad_tensors(self, i: int) -> List[torch.Tensor]:
        """Distinct gradient buffers of layer i (the q/k/v weight and bias gradients are row slices of one buffer each)."""
        out, seen = [], set()
        for p, gt in self.layer_grads[i]:
            if not p.requires_grad:
                continue
            base = gt._base if gt._base is not None else gt
            if base.data_ptr() not in seen:
                seen.add(base.data_ptr())
                out.append(base)
        return out

    def close(self):
        """Back to the eager loop: drop the salt binding of the capture stream and the frozen weight cache."""
        if getattr(self, "_cap_stream", None) is not None:
            _unbind_salt(self._cap_stream.cuda_stream)
            self._cap_stream = None
            self.w._cache_mode = None
            self.w._cache.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        # a runner dropped without close() must not leave the library a pointer into a freed tensor; the model's weight
        # cache is left alone here (a newer runner of the same model may own it by now)
        try:
            if getattr(self, "_cap_stream", None) is not None:
                _unbind_salt(self._cap_stream.cuda_stream)
                self._cap_stream = None
        except Exception:               # noqa: BLE001 - interpreter shutdown
            pass


class BucketedGraphedStep:
    """hipGraph replay for REAL batches.  The reference's collate pads every batch to ITS OWN longest utterance
    (train/train_aptai.py:268-285, train/train_phoneme_recognizer.py:224-239), so the tensor shapes change from step to step,
    while a captured graph has one shape.  This wrapper keeps a small cache of GraphedAPTAIStep runners, one per (batch size, BUCKET
    length): a batch is zero-padded up to the next bucket, fed to that bucket's graphs (captured on first use, sharing the model's
    parameters, persistent weight copies and the optimiser), and the outputs are cut back to the batch's own frame count.

    Results equal the eager loop on the un-bucketed batch: every length-dependent quantity is dynamic inside the graphs -
    per-utterance frame counts travel as device tensors (attention key mask, padded-frame zeroing, SpecAugment span counts, loss
    masks, CTC lengths), and the two places where the reference sees the batch's PADDED length (GroupNorm statistics of the first
    conv layer over all frames of the collated batch; the 'same' zero padding of LowPassFilterLayer) read per-step frame bounds
    (aptai_set_frame_bounds).  Extra frames of the bucket are padded frames like the reference's own: zeroed, masked as keys, outside
    every loss, zero gradient.  288 GB of HBM hold a dozen buckets of saved activations (~3.5 GB each at 16 x 10 s)."""

    DEFAULT_SECONDS = (1, 2, 3, 4, 5, 6, 8, 10, 12, 15, 20, 25, 30)

    def __init__(self, model, optimizer, bucket_samples=None, reducer=None, label_bucket: int = 32, max_runners: int = 8, log=None):
        self.model, self.opt, self.reducer = model, optimizer, reducer
        # Every runner holds a capture stream with a salt and a bounds slot (64 of each in the library, csrc/runtime.hip) and several GB of
        # saved activations and graph pools; a real corpus spans 13 length buckets x several label widths (+ a short last batch), so the
        # cache is bounded: beyond `max_runners` the least recently used runner is closed (its graphs are re-captured if its shape returns).
        self.max_runners, self.log, self.evictions = max(1, int(max_runners)), log, 0
        self.kind = "pr" if isinstance(model, Wav2Vec2_PR) else "aptai"
        self.buckets = sorted(bucket_samples) if bucket_samples else [16000 * s for s in self.DEFAULT_SECONDS]
        self.label_bucket = label_bucket
        self.runners: Dict[tuple, GraphedAPTAIStep] = {}
        self.w = model.wav2vec2

    def bucket_of(self, S: int) -> int:
        for b in self.buckets:
            if b >= S:
                return b
        return -(-S // 16000) * 16000                      # longer than every bucket: whole seconds

    def _padded(self, batch, Sb: int, Tb: int):
        akey = "input_values" if self.kind == "pr" else "audio_inputs"
        out = {}
        for k, v in batch.items():
            if k == akey:
                if v.shape[1] < Sb:
                    v = torch.nn.functional.pad(v, (0, Sb - v.shape[1]))
            elif k == "phoneme_labels" and self.kind == "pr":
                wl = -(-v.shape[1] // self.label_bucket) * self.label_bucket
                if v.shape[1] < wl:
                    v = torch.nn.functional.pad(v, (0, wl - v.shape[1]), value=-100)
            elif k in ("audio_lengths", "input_lengths", "phoneme_labels"):
                pass
            elif v.dim() == 2 and self.kind == "aptai":            # frame-rate targets (B, T): -100.0 / 0 = the collate's own padding values
                if v.shape[1] < Tb:
                    v = torch.nn.functional.pad(v, (0, Tb - v.shape[1]), value=0 if k == "phn_frames_49hz" else -100.0)
                elif v.shape[1] > Tb:
                    v = v[:, :Tb]
            out[k] = v
        return out

    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        akey = "input_values" if self.kind == "pr" else "audio_inputs"
        B, S = batch[akey].shape
        Sb = self.bucket_of(S)
        g_s, g_b = self.w._geometry(B, S), self.w._geometry(B, Sb)
        padded = self._padded(batch, Sb, g_b.T)
        key = (B, Sb) + ((padded["phoneme_labels"].shape[1],) if self.kind == "pr" else ())
        r = self.runners.pop(key, None)
        if r is None:
            while len(self.runners) >= self.max_runners:
                old_key = next(iter(self.runners))              # dicts keep insertion order: the first key is the least recently used
                self.runners.pop(old_key).close()
                self.evictions += 1
                if self.log is not None:
                    self.log(f"[graphed] closed the runner of shape {old_key} (cache of {self.max_runners} is full)")
            dev = next(self.model.parameters()).device
            r = GraphedAPTAIStep(self.model, self.opt, {k: v.to(dev) for k, v in padded.items()}, reducer=self.reducer)
            if self.log is not None:
                self.log(f"[graphed] captured shape {key}: {len(self.runners) + 1} runner(s), "
                         f"{torch.cuda.memory_allocated(dev) / 2 ** 30:.1f} GiB allocated")
        self.runners[key] = r                                   # (re-)inserted last = most recently used
        r.set_bounds(g_s.Tl[0], g_s.T)
        out = r.step(padded)
        T = g_s.T
        if self.kind == "pr":
            return {"loss": out["loss"], "phoneme_logits": out["phoneme_logits"][:, :T], "log_probs": out["log_probs"][:T],
                    "hidden_states": out["hidden_states"][:, :T]}
        return {"loss": out["loss"], "mse_loss": out["mse_loss"], "ce_loss": out["ce_loss"], "tvs_pred": out["tvs_pred"][:, :T],
                "phn_fc_pred": out["phn_fc_pred"][:, :T]}

    def suspend(self) -> None:
        """Before an EAGER phase on the same model (validation between epochs): un-freeze the model's compute-copy cache so that
        eager forwards rebuild their copies from the current parameters.  The captured graphs keep their own copies (refreshed by
        their prep segment on every replay), so the next step() needs nothing."""
        self.w._cache_mode = None
        self.w._cache.clear()

    def close(self):
        for r in self.runners.values():
            r.close()
        self.runners = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class GraphedForceStep:
    """hipGraph runner for the Force_APTAI train step (models/force_aptai.py:80-178, train/train_force_aptai.py:392-531): TWO graphs,

        encoder (frozen recogniser in inference mode + best-path decode, on a side stream)  |  heads forward + loss + heads backward

    replayed one batch apart: while the heads of batch n replay on the caller's stream, the encoder of batch n+1 (handed to
    step() as `next_batch`, or the same batch again) replays on the side stream.  Same kernels and the same two Python functions
    (force_heads_fwd / force_heads_bwd) as the eager autograd path; dropout draws fresh masks per replay through the per-stream
    salt; nothing synchronises host and device (`lists()` makes the Python lists of the reference's return value on demand).
    The optimiser step is issued eagerly after the heads graph."""

    def __init__(self, model, optimizer, batch: Dict[str, torch.Tensor]):
        from .force_aptai import Force_APTAI, force_heads_bwd, force_heads_fwd
        assert isinstance(model, Force_APTAI) and model.training, "a Force_APTAI model in train() mode"
        if getattr(model, "dp_loss_norm", None) is not None:
            raise NotImplementedError("GraphedForceStep is single-process (the eager step carries the DP loss normalisation)")
        self.model, self.opt = model, optimizer
        dev = next(model.parameters()).device
        self.dev = dev
        self.audio = batch["audio_inputs"].float().contiguous().clone()
        self.lengths = batch["audio_lengths"].clone()
        self._tracks = [k for k in batch if k not in ("audio_inputs", "audio_lengths", "phn_frames_49hz", "phoneme_labels")]
        w = model.w2v2_pr.wav2vec2
        B, S = self.audio.shape
        self.g = w._geometry(B, S)
        g = self.g
        n_tv = model.rnn.linear[3].weight.shape[0]
        self.tv_tgt = torch.zeros((B, g.T, n_tv), device=dev, dtype=torch.float32)
        self.salt = torch.zeros(2, device=dev, dtype=torch.int32)
        self._salt_ring = [torch.zeros(2, dtype=torch.int32).pin_memory() for _ in range(4)]
        self._salt_events = [None] * 4
        self._salt_turn = 0
        self._salt_gen = np.random.RandomState(0xF0CE + w.base_seed)
        self._cap_stream = torch.cuda.Stream(device=dev)
        self._enc_stream = torch.cuda.Stream(device=dev)
        _bind_salt(self._cap_stream.cuda_stream, self.salt)
        # one eager step: scratch buffers, code objects, weight copies
        model.zero_grad(set_to_none=True)
        model(0, **batch)["loss"].backward()
        model.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        # ---- encoder graph (side stream): static inputs -> enc.{ac, ids, nlen, frame_lens}
        self.g_enc = torch.cuda.CUDAGraph()
        enc_tile = int(os.environ.get("APTAI_FORCE_ENC_TILE", "0"))
        # ^ ops.auto_tile around the side-stream pass.  Rounds 3-4a forced 128-row tiles here: whole-CU workgroups kept the heads' BiLSTM
        # clusters waiting (12 % of the step).  Since the cluster kernels hold their compute units alone (csrc/lstm.hip, APTAI_LSTM_LDS_KB)
        # the dispatcher's own rule is the faster one again: 6.36 ms with 128, 6.09 ms with 0, 6.63 ms with 192 (interleaved, one box).
        with torch.cuda.graph(self.g_enc, pool=pool, stream=self._enc_stream, capture_error_mode=_CAPTURE_MODE), ops.auto_tile(enc_tile):
            self.enc = model._encode(self.audio, self.lengths)
        # ---- heads graph: its own copies of the encoder outputs (the next encoder replay overwrites enc.*)
        e = self.enc
        self.h_ac, self.h_ids, self.h_nlen, self.h_fl = (torch.empty_like(t) for t in (e.ac, e.ids, e.nlen, e.frame_lens))
        self.st, self.P = model._heads_state(g, self.h_ids, self.h_nlen, self.h_fl, self.tv_tgt, 0xF0)
        self.g_heads = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_heads, pool=pool, stream=self._cap_stream, capture_error_mode=_CAPTURE_MODE):
            self.st.vocab_sizes = self.h_nlen + 1             # derived from a static input: recomputed on every replay
            self.outs, saved = force_heads_fwd(self.h_ac, self.st, self.P)
            self.pgrads = force_heads_bwd(saved, self.st, self.P, self.h_ac, None)
        torch.cuda.synchronize()
        self._have_enc = False
        self._enc_done = None
        self._last = batch
        self.set_batch(batch)

    # ------------------------------------------------------------------ inputs of the HEADS (targets) and of the encoder
    def set_batch(self, batch: Dict[str, torch.Tensor]) -> None:
        """Targets of the batch whose heads run in the next step()."""
        self.tv_tgt.copy_(torch.stack([batch[k] for k in self._tracks], dim=-1).float(), non_blocking=True)

    def _launch_encoder(self, batch: Dict[str, torch.Tensor]) -> None:
        cur = torch.cuda.current_stream(self.dev)
        es = self._enc_stream
        es.wait_stream(cur)                                   # the batch tensors, and the previous copies out of enc.*
        with torch.cuda.stream(es):
            self.audio.copy_(batch["audio_inputs"].float(), non_blocking=True)
            self.lengths.copy_(batch["audio_lengths"], non_blocking=True)
            self.g_enc.replay()
            self._enc_done = torch.cuda.Event()
            self._enc_done.record(es)
        self.model.w2v2_pr.wav2vec2._step += 1
        self._have_enc = True

    def _salt_step(self):
        slot = self._salt_turn
        self._salt_turn = (slot + 1) % len(self._salt_ring)
        if self._salt_events[slot] is not None:
            self._salt_events[slot].synchronize()
        self._salt_ring[slot].copy_(torch.from_numpy(self._salt_gen.randint(-2 ** 31, 2 ** 31 - 1, size=2).astype(np.int32)))
        self.salt.copy_(self._salt_ring[slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._salt_events[slot] = ev

    def _assign_grads(self, keep) -> None:
        """Every trainable parameter's `.grad` = its static gradient buffer (None for a layer LayerDrop skipped this step)."""
        for p, gt in self.grads.items():
            if p.requires_grad:
                p.grad = gt
        for i in range(self.cfg.num_hidden_layers):
            for p, gt in self.layer_grads[i]:
                if p.requires_grad:
                    p.grad = gt if keep[i] else None
        if not getattr(self, "_checked", False):
            names = {id(p): n for n, p in self.model.named_parameters()}
            for p in self.model.parameters():
                if p.grad is not None and (p.grad.dtype != p.dtype or p.grad.shape != p.shape or not p.grad.is_contiguous()
                                           or p.grad.device != p.device):
                    raise RuntimeError(f"static gradient of {names[id(p)]} has dtype {p.grad.dtype} shape {tuple(p.grad.shape)} "
                                       f"contiguous={p.grad.is_contiguous()} (parameter: {p.dtype} {tuple(p.shape)})")
            self._checked = True

    # ------------------------------------------------------------------ one optimiser step
    def step(self, batch: Optional[Dict[str, torch.Tensor]] = None, next_batch: Optional[Dict[str, torch.Tensor]] = None):
        """Heads of `batch` (default: the batch given last) + optimiser step; the encoder pass of `next_batch` (default: `batch`
        again) goes out on the side stream first.  The very first call encodes `batch` inline."""
        if batch is not None:
            self.set_batch(batch)
            self._last = batch
        batch = batch if batch is not None else self._last
        cur = torch.cuda.current_stream(self.dev)
        if not self._have_enc:
            self._launch_encoder(batch)
        cur.wait_event(self._enc_done)
        e = self.enc
        self.h_ac.copy_(e.ac, non_blocking=True); self.h_ids.copy_(e.ids, non_blocking=True)
        self.h_nlen.copy_(e.nlen, non_blocking=True); self.h_fl.copy_(e.frame_lens, non_blocking=True)
        self._launch_encoder(next_batch if next_batch is not None else batch)      # waits for the four copies above
        self._salt_step()
        self.g_heads.replay()
        for p, gt in zip(self.P, self.pgrads):
            if p.requires_grad:
                p.grad = gt                                    # (bias_ih / bias_hh of a direction share one gradient buffer)
        self.opt.step()
        loss, tv_loss, align_loss, tvs, frame_phns = self.outs[:5]
        return {"loss": loss, "tv_loss": tv_loss, "align_loss": align_loss, "tvs_pred": tvs, "frame_phns": frame_phns,
                "ids": self.h_ids, "n_ids": self.h_nlen, "frame_lens": self.h_fl}

    def lists(self, out) -> Dict[str, list]:
        """The Python lists of Force_APTAI.forward (pred_frame_phns, pred_ctc_phn_seq) for a step() result: the only transfers."""
        given, fl, n, table = self.model._lists((out["ids"], out["n_ids"], out["frame_lens"], None))
        fp = out["frame_phns"].cpu().numpy()
        return {"pred_frame_phns": [fp[b, :fl[b]].tolist() for b in range(self.g.B)], "pred_ctc_phn_seq": given}

    def close(self):
        if getattr(self, "_cap_stream", None) is not None:
            torch.cuda.synchronize(self.dev)
            _unbind_salt(self._cap_stream.cuda_stream)
            self._cap_stream = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            if getattr(self, "_cap_stream", None) is not None:
                _unbind_salt(self._cap_stream.cuda_stream)
                self._cap_stream = None
        except Exception:               # noqa: BLE001 - interpreter shutdown
            pass
