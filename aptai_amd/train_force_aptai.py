"""The Force_APTAI training loop of the reference (train/train_force_aptai.py) on the MI355X build: same function surface
(`load_model_optimizer`, `train`, `validate`, `test`), per-batch protocol (`model(epoch, **batch_x)` -> `loss.backward()` ->
`optimizer.step()`; only the heads train, the `Wav2Vec2_PR` encoder is frozen and runs in inference mode), schedule, validation /
test keys and best-checkpoint files.  Differences from train_aptai.py follow the reference's own diff: `--pr_model_path`,
`phoneme_labels` in the batch (:271-275), tv/align losses in the log, CTC-based PER (:578-586), `pred_frame_phns` as the
frame prediction.  The corpus reader, LOSO bookkeeping and wandb are out of scope: `SyntheticHPRC` items
carry a `phoneme_label` sequence as well.

    python -m aptai_amd.train_force_aptai --pr_model_path <dir with best-model-ckpt/> --num_epochs 2
"""
from __future__ import annotations

import argparse
import pickle
from pathlib import Path
from types import SimpleNamespace
from typing import Dict

import numpy as np
import torch

from . import hostlogic, metrics
from .force_aptai import Force_APTAI
from .train_aptai import SyntheticHPRC, _eval_frames, _stack_gt, _tv_test_summary


class SyntheticHPRCWithLabels(SyntheticHPRC):
    """SyntheticHPRC items + the `phoneme_label` id sequence of data/dataset_hprc.py (20..55 ids)."""

    def __init__(self, *a, vocab_size: int = 40, **kw):
        super().__init__(*a, **kw)
        self.vocab_size = vocab_size

    def __getitem__(self, i):
        item = super().__getitem__(i)
        g = np.random.RandomState(self.seed * 7919 + i + 17)
        item["phn_frames_49hz"] = (item["phn_frames_49hz"] % (self.vocab_size - 1) + 1).astype(np.int64)
        item["phoneme_label"] = g.randint(1, self.vocab_size, size=int(g.randint(20, 56))).astype(np.int32)
        return item


def collate(batch):
    return hostlogic.collate_aptai(batch, with_phoneme_labels=True)


def load_model_optimizer(args_cfg):
    """train/train_force_aptai.py:328-368: Force_APTAI over a trained recogniser checkpoint, Adam over the parameters that
    require gradients (the heads), LambdaLR with the 10x warm-up schedule."""
    model = Force_APTAI(args_cfg.pr_model_path, args_cfg.device, args_cfg.vocab).to(args_cfg.device)
    from .optim import Adam
    optimizer = Adam([p for p in model.parameters() if p.requires_grad], lr=args_cfg.learning_rate,
                     betas=(args_cfg.adam_beta1, args_cfg.adam_beta2), eps=args_cfg.adam_epsilon, weight_decay=args_cfg.adam_weight_decay)
    lr_scheduler = torch.optim.lr_scheduler.LambdaLR(
        optimizer=optimizer, lr_lambda=hostlogic.get_lr_schedule(args_cfg.num_warmup_epochs, args_cfg.num_static_epochs, args_cfg.lr_decay))
    return model, optimizer, lr_scheduler


def train(cfg, model, optimizer, lr_scheduler, train_dataloader, valid_dataloader, test_spk, best_ckpt_path, log=print):
    """train/train_force_aptai.py:392-531.  Returns the per-epoch log dicts."""
    eval_target = None
    history = []
    best_ckpt_path = Path(best_ckpt_path)
    best_ckpt_path.mkdir(parents=True, exist_ok=True)
    for epoch in range(cfg.num_epochs):
        sum_train_loss, steps = 0.0, 0
        model.train()
        # one batch of lookahead: the frozen recogniser's pass for batch i+1 is started on a side stream before the heads of
        # batch i are launched (Force_APTAI.prefetch; results do not depend on it).  cfg.pipeline_encoder = False turns it off.
        pipelined = bool(getattr(cfg, "pipeline_encoder", True)) and str(cfg.device).startswith("cuda")
        it = iter(train_dataloader)
        nxt = next(it, None)
        if nxt is not None:
            nxt = {k: v.to(cfg.device) for k, v in nxt.items()}
        batch_idx = -1
        while nxt is not None:
            batch_idx += 1
            batch_x, nxt = nxt, next(it, None)
            if nxt is not None:
                nxt = {k: v.to(cfg.device) for k, v in nxt.items()}
            optimizer.zero_grad()
            ahead = (nxt["audio_inputs"], nxt["audio_lengths"]) if (pipelined and nxt is not None) else None
            outputs = model(epoch, **batch_x, _prefetch_next=ahead)
            outputs["loss"].backward()
            optimizer.step()
            sum_train_loss += float(outputs["loss"].detach())
            steps += 1
            log(f"\tepoch {epoch + 1} ~ batch {batch_idx + 1}/{len(train_dataloader)}, train_loss: {float(outputs['loss'].detach()):.4f}, "
                f"train_tv_loss: {float(outputs['tv_loss'].detach()):.4f}, train_align_loss: {float(outputs['align_loss'].detach()):.4f}, "
                f"lr: {optimizer.param_groups[0]['lr']:.6f}")
        lr_scheduler.step()
        model.eval()
        val_logs = validate(model, cfg.device, cfg.vocab, epoch, getattr(cfg, "exp_dir", None), test_spk, valid_dataloader)
        better = (eval_target is None
                  or (cfg.target_metric_bigger_better and eval_target <= val_logs[cfg.target_metric])
                  or (not cfg.target_metric_bigger_better and eval_target >= val_logs[cfg.target_metric]))
        if better:
            eval_target = val_logs[cfg.target_metric]
            torch.save(model.state_dict(), best_ckpt_path / "pytorch_model.bin")
            pickle.dump(model.get_config(), open(best_ckpt_path / "model_cfg.pkl", "wb"))
        epoch_log = dict(val_logs, epoch=epoch, mean_train_loss=sum_train_loss / max(steps, 1), lr=optimizer.param_groups[0]["lr"],
                         saved=bool(better))
        history.append(epoch_log)
        log(f"Epoch {epoch + 1}/{cfg.num_epochs} -> " + " | ".join(f"{k}: {v:.4f}" for k, v in epoch_log.items() if isinstance(v, float)))
    return history


def _one_file(model, device, epoch, batch_x):
    """One batch-1 evaluation pass shared by validate() and test(): TV arrays, CTC-based edit distance (:578-586), frame scores
    with `pred_frame_phns` as the prediction (:588-600)."""
    with torch.no_grad():
        tvs_gt = _stack_gt(batch_x)
        batch_x = {k: v.to(device) for k, v in batch_x.items()}
        outputs = model(epoch, **batch_x)
    tvs_gt = torch.squeeze(tvs_gt, dim=0).cpu().numpy()
    tvs_pred = torch.squeeze(outputs["tvs_pred"], dim=0).float().cpu().numpy()
    gt_phn = batch_x["phoneme_labels"].cpu().numpy()[0]
    pred_phn = np.asarray(outputs["pred_ctc_phn_seq"][0]).tolist()
    ed, n = metrics.edit_distance(gt_phn, pred_phn), len(gt_phn)
    pred_frames = torch.tensor(outputs["pred_frame_phns"], device=device)
    return outputs, tvs_gt, tvs_pred, ed, n, _eval_frames(batch_x["phn_frames_49hz"], pred_frames)


def validate(model, device, vocab, epoch, exp_dir, test_spk, val_dl, log_step=100) -> Dict[str, float]:
    """train/train_force_aptai.py:533-652, batch size 1 (incl. the TTCD-twice ground-truth stack)."""
    val_losses, val_rmses, val_pccs, val_overlaps = [], [], [], []
    val_ps, val_rs, val_f1s, val_rvals, edit_d, n_phn = [], [], [], [], [], []
    total_frames = corr_frames = 0
    for batch_x in val_dl:
        outputs, tvs_gt, tvs_pred, ed, n, (frames, corr, overlap, (p, r, f1, rval), _, _) = _one_file(model, device, epoch, batch_x)
        val_losses.append(outputs["loss"].item())
        val_rmses.append(np.mean(list(metrics.tvs_metric_rmse(tvs_gt, tvs_pred).values())))
        val_pccs.append(np.mean([v[0] for v in metrics.tvs_metric_ppc(tvs_gt, tvs_pred).values()]))
        edit_d.append(ed); n_phn.append(n)
        total_frames += frames
        corr_frames += corr
        val_overlaps.append(overlap)
        val_ps.append(p); val_rs.append(r); val_f1s.append(f1); val_rvals.append(rval)
    return {
        "val_mean_loss": float(np.mean(val_losses)), "val_mean_rmse": float(np.mean(val_rmses)),
        "val_mean_pcc": float(np.mean(val_pccs)), "val_mean_FER": 1 - (corr_frames / total_frames),
        "val_mean_PER": float(np.sum(edit_d) / np.sum(n_phn)), "val_mean_F1": float(np.mean(val_f1s)),
        "val_mean_p": float(np.mean(val_ps)), "val_mean_r": float(np.mean(val_rs)), "val_mean_Rval": float(np.mean(val_rvals)),
        "val_mean_overlap": float(np.mean(val_overlaps)),
    }


def test(model, device, vocab, exp_dir, test_spk, test_dl, rate, log_step=100, num_epochs=0) -> Dict[str, float]:
    """train/train_force_aptai.py:655-838: as train_aptai.test plus the std entries and the CTC-based PER."""
    assert rate in ["F", "N"]
    names = hostlogic.TV_NAMES
    rmse_tvs, pcc_tvs = {n: [] for n in names}, {n: [] for n in names}
    overlaps, ps, rs, f1s, rvals, edit_d, n_phn, pers = [], [], [], [], [], [], [], []
    total_frames = corr_frames = 0
    model.eval()
    for batch_x in test_dl:
        _, tvs_gt, tvs_pred, ed, n, (frames, corr, overlap, (p, r, f1, rval), _, _) = _one_file(model, device, num_epochs, batch_x)
        edit_d.append(ed); n_phn.append(n); pers.append(ed / n)
        total_frames += frames
        corr_frames += corr
        overlaps.append(overlap)
        ps.append(p); rs.append(r); f1s.append(f1); rvals.append(rval)
        rm, pc = metrics.tvs_metric_rmse(tvs_gt, tvs_pred), metrics.tvs_metric_ppc(tvs_gt, tvs_pred)
        for nme in names:
            rmse_tvs[nme].append(rm[nme])
            pcc_tvs[nme].append(pc[nme][0])
    out = _tv_test_summary(rate, rmse_tvs, pcc_tvs, with_std=True)
    out.update({f"test_{rate}_mean_FER": 1 - (corr_frames / total_frames),
                f"test_{rate}_mean_PER": float(np.sum(edit_d) / np.sum(n_phn)), f"test_{rate}_std_PER": float(np.std(pers)),
                f"test_{rate}_mean_overlap": float(np.mean(overlaps)), f"test_{rate}_std_overlap": float(np.std(overlaps)),
                f"test_{rate}_mean_F1": float(np.mean(f1s)), f"test_{rate}_mean_p": float(np.mean(ps)),
                f"test_{rate}_mean_r": float(np.mean(rs)), f"test_{rate}_mean_Rval": float(np.mean(rvals))})
    return out


def default_cfg(**kw):
    """Hyper-parameters at the reference's argparse defaults (train/train_force_aptai.py:45-140; start_train_force_aptai.sh)."""
    vocab = {"(blank)": 0, "(...)": 1}
    vocab.update({f"p{i}": i for i in range(2, 40)})
    cfg = SimpleNamespace(device="cuda", num_epochs=2, batch_size=5, learning_rate=1e-5, adam_beta1=0.9, adam_beta2=0.999,
                          adam_epsilon=1e-8, adam_weight_decay=0.0, num_warmup_epochs=10, num_static_epochs=30, lr_decay=0.96,
                          target_metric="val_mean_rmse", target_metric_bigger_better=False, exp_dir=None, vocab=vocab,
                          pr_model_path=None)
    cfg.__dict__.update(kw)
    return cfg


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--pr_model_path", required=True, help="directory holding best-model-ckpt/{pytorch_model.bin, model_cfg.pkl}")
    ap.add_argument("--num_epochs", type=int, default=2)
    ap.add_argument("--steps_per_epoch", type=int, default=8)
    ap.add_argument("--val_items", type=int, default=4)
    ap.add_argument("--batch_size", type=int, default=5)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--learning_rate", type=float, default=1e-5)
    ap.add_argument("--out", default="force_aptai_ckpt")
    a = ap.parse_args(argv)
    cfg = default_cfg(num_epochs=a.num_epochs, batch_size=a.batch_size, learning_rate=a.learning_rate, pr_model_path=a.pr_model_path)
    model, optimizer, lr_scheduler = load_model_optimizer(cfg)
    w2v = model.w2v2_pr.wav2vec2.config
    train_ds = SyntheticHPRCWithLabels(a.steps_per_epoch * a.batch_size, a.seconds, seed=1, cfg=w2v, vocab_size=len(cfg.vocab))
    val_ds = SyntheticHPRCWithLabels(a.val_items, a.seconds, seed=2, cfg=w2v, vocab_size=len(cfg.vocab))
    train_dl = torch.utils.data.DataLoader(train_ds, batch_size=a.batch_size, shuffle=True, drop_last=True, collate_fn=collate)
    val_dl = torch.utils.data.DataLoader(val_ds, batch_size=1, shuffle=False, collate_fn=collate)
    return train(cfg, model, optimizer, lr_scheduler, train_dl, val_dl, "synthetic", a.out)


if __name__ == "__main__":
    main()
