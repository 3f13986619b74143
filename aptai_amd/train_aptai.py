"""The APTAI training loop of the reference (train/train_aptai.py) on the MI355X build: same function surface
(`load_model_optimizer`, `train`, `validate`), same per-batch protocol (`model(epoch, **batch_x)` -> `loss.backward()` ->
`optimizer.step()`), same schedule, same validation metrics and return keys, same best-checkpoint files
(`pytorch_model.bin` + `model_cfg.pkl`).  What is NOT here: the HPRC corpus reader, leave-one-speaker-out bookkeeping, wandb
(SURVEY.md §2 rows marked out of scope) — `SyntheticHPRC` yields items with the fields `_collate_fn` consumes instead.

    python -m aptai_amd.train_aptai --model_dir <local wav2vec2 dir> --num_epochs 2 --steps_per_epoch 20 --batch_size 16
    python -m aptai_amd.train_aptai --random_init base --graphed        # hipGraph segments, one captured runner per length bucket
"""
from __future__ import annotations

import argparse
import pickle
import tempfile
from pathlib import Path
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch

from . import hostlogic, metrics
from .aptai import APTAI
from .config import W2V2Config
from .wav2vec2 import Wav2Vec2Model

VOCAB_SIZE = 46        # models/aptai.py:54 hard-codes Linear(1024, 46)


class SyntheticHPRC(torch.utils.data.Dataset):
    """Items shaped like data/dataset_hprc.py's (audio, audio_len, phn_frames_49hz, tvs_norm_49hz[9 tracks]) with the
    synthetic content SURVEY.md §8d prescribes: N(0,1) audio, uniform frame labels, N(0,1) trajectories."""

    def __init__(self, n_items: int, seconds: float = 10.0, vary_length: bool = True, seed: int = 0, cfg: Optional[W2V2Config] = None):
        self.n, self.S, self.vary, self.seed = n_items, int(16000 * seconds), vary_length, seed
        self.cfg = cfg or W2V2Config.base()

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = np.random.RandomState(self.seed * 100003 + i)
        n = self.S if (not self.vary or i % 2 == 0) else int(g.randint(int(0.8 * self.S), self.S + 1))
        T = int(hostlogic.feat_extract_output_lengths(n, self.cfg.conv_kernel, self.cfg.conv_stride))
        labels = np.repeat(g.randint(1, VOCAB_SIZE, size=T // 4 + 1), 4)[:T]        # phone-like runs of 4 frames
        return {"audio": g.randn(n).astype(np.float32), "audio_len": n, "phn_frames_49hz": labels.astype(np.int64),
                "tvs_norm_49hz": {k: g.randn(T) for k in hostlogic.TV_NAMES}}


def load_model_optimizer(args_cfg):
    """train/train_aptai.py:334-372: APTAI on a LOCAL wav2vec2 directory, Adam, LambdaLR with the 10x warm-up schedule."""
    pretrain_cfg = args_cfg.pretrain_cfg
    model = APTAI(device=args_cfg.device, vocab=args_cfg.vocab, huggingface_model_id=args_cfg.huggingface_model_id,
                  pretrain_cfg=pretrain_cfg, cache_dir=getattr(args_cfg, "cache_dir", None)).to(args_cfg.device)
    # torch.optim.Adam's update rule as one multi-tensor HIP kernel (aptai_amd.optim.Adam; same constructor, param_groups and
    # state keys).  `torch.optim.Adam(model.parameters(), ...)` works unchanged on the same parameters.
    from .optim import Adam
    optimizer = Adam(model.parameters(), lr=args_cfg.learning_rate, betas=(args_cfg.adam_beta1, args_cfg.adam_beta2),
                     eps=args_cfg.adam_epsilon, weight_decay=args_cfg.adam_weight_decay).publish_to(model)
    lr_scheduler = torch.optim.lr_scheduler.LambdaLR(
        optimizer=optimizer, lr_lambda=hostlogic.get_lr_schedule(args_cfg.num_warmup_epochs, args_cfg.num_static_epochs, args_cfg.lr_decay))
    return model, optimizer, lr_scheduler


def train(cfg, model, optimizer, lr_scheduler, train_dataloader, valid_dataloader, test_spk, best_ckpt_path, log=print):
    """train/train_aptai.py:392-531.  Returns the per-epoch log dicts (the reference only prints them)."""
    eval_target = None
    history = []
    runner = None
    best_ckpt_path = Path(best_ckpt_path)
    best_ckpt_path.mkdir(parents=True, exist_ok=True)
    for epoch in range(cfg.num_epochs):
        sum_train_loss, steps = 0.0, 0
        model.train()
        for batch_idx, batch_x in enumerate(train_dataloader):
            if getattr(cfg, "graphed", False):
                # same step as below, replayed as hipGraph segments.  The collate pads every batch to its own longest utterance
                # (train/train_aptai.py:268-285), so shapes vary: BucketedGraphedStep keeps one captured runner per (batch size,
                # bucket length) and feeds each batch to the next larger bucket (results equal the eager step on the batch's own
                # shape).  The runners take the collate_fn's HOST batch: pinned staging ring + asynchronous copies (set_batch)
                if runner is None:
                    from .graphed import BucketedGraphedStep
                    runner = BucketedGraphedStep(model, optimizer)
                outputs = runner.step(batch_x)
            else:
                batch_x = {k: v.to(cfg.device) for k, v in batch_x.items()}
                optimizer.zero_grad()
                outputs = model(epoch, **batch_x)
                outputs["loss"].backward()
                optimizer.step()
            sum_train_loss += float(outputs["loss"].detach())
            steps += 1
            log(f"\tepoch {epoch + 1} ~ batch {batch_idx + 1}/{len(train_dataloader)}, train_loss: {float(outputs['loss'].detach()):.4f}, "
                f"train_mse_loss: {float(outputs['mse_loss'].detach()):.4f}, train_ce_loss: {float(outputs['ce_loss'].detach()):.4f}, "
                f"lr: {optimizer.param_groups[0]['lr']:.6f}")
        lr_scheduler.step()
        if runner is not None:
            runner.suspend()             # the eager validation below rebuilds its weight copies; the captured buckets stay
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
    if runner is not None:
        runner.close()
    return history


def validate(model, device, vocab, epoch, exp_dir, test_spk, val_dl, log_step=100) -> Dict[str, float]:
    """train/train_aptai.py:533-652, batch size 1.  Reproduces the reference as written, including its two quirks: the ground
    truth stack lists TTCD in the TMCD slot (:557-560) and `get_stats` receives frame label sequences, not boundary times."""
    val_losses, val_rmses, val_pccs, val_overlaps = [], [], [], []
    val_ps, val_rs, val_f1s, val_rvals, edit_d, n_phn = [], [], [], [], [], []
    total_frames = corr_frames = 0
    for batch_x in val_dl:
        with torch.no_grad():
            tvs_gt = torch.stack([batch_x["LA"], batch_x["LP"], batch_x["JA"], batch_x["TTCL"], batch_x["TTCD"], batch_x["TMCL"],
                                  batch_x["TTCD"], batch_x["TBCL"], batch_x["TBCD"]], dim=-1).float()
            batch_x = {k: v.to(device) for k, v in batch_x.items()}
            outputs = model(epoch, **batch_x)
        val_losses.append(outputs["loss"].item())
        tvs_gt = torch.squeeze(tvs_gt, dim=0).cpu().numpy()
        tvs_pred = torch.squeeze(outputs["tvs_pred"], dim=0).float().cpu().numpy()
        val_rmses.append(np.mean(list(metrics.tvs_metric_rmse(tvs_gt, tvs_pred).values())))
        val_pccs.append(np.mean([v[0] for v in metrics.tvs_metric_ppc(tvs_gt, tvs_pred).values()]))
        gt_frames, pred_frames = batch_x["phn_frames_49hz"], outputs["phn_fc_pred"]
        total_frames += gt_frames.size(1)
        corr_frames += int(torch.sum(torch.eq(gt_frames, pred_frames)).item())
        gt_f, p_f = gt_frames.cpu().numpy(), pred_frames.cpu().numpy()
        val_overlaps.append(metrics.evaluate_overlap(gt_f, p_f))
        y, yhat = gt_f.squeeze(), p_f.squeeze()
        p, r, f1, rval = metrics.get_stats(y, yhat, tolerance=0.02)
        val_ps.append(p); val_rs.append(r); val_f1s.append(f1); val_rvals.append(rval)
        y_grp, yhat_grp = metrics.phn_frame_id2phn(y.tolist()), metrics.phn_frame_id2phn(yhat.tolist())
        edit_d.append(metrics.compute_PER(y_grp, yhat_grp) / 100.0 * len(y_grp))
        n_phn.append(len(y_grp))
    return {
        "val_mean_loss": float(np.mean(val_losses)), "val_mean_rmse": float(np.mean(val_rmses)),
        "val_mean_pcc": float(np.mean(val_pccs)), "val_mean_FER": 1 - (corr_frames / total_frames),
        "val_mean_PER": float(np.sum(edit_d) / np.sum(n_phn)), "val_mean_F1": float(np.mean(val_f1s)),
        "val_mean_p": float(np.mean(val_ps)), "val_mean_r": float(np.mean(val_rs)), "val_mean_Rval": float(np.mean(val_rvals)),
        "val_mean_overlap": float(np.mean(val_overlaps)),
    }


def _eval_frames(gt_frames, pred_frames):
    """Frame-level scores shared by validate()/test() of both TV models (train/train_aptai.py:583-607, 717-749): number of
    frames, number correct, overlap, boundary precision / recall / F1 / R-value as the reference calls `get_stats`."""
    gt_f, p_f = gt_frames.cpu().numpy(), pred_frames.cpu().numpy()
    y, yhat = gt_f.squeeze(), p_f.squeeze()
    return (gt_frames.size(1), int(torch.sum(torch.eq(gt_frames, pred_frames.to(gt_frames.device))).item()),
            metrics.evaluate_overlap(gt_f, p_f), metrics.get_stats(y, yhat, tolerance=0.02), y, yhat)


def _stack_gt(batch_x):
    """Ground-truth stack of validate()/test() AS WRITTEN in the reference: TTCD sits in the TMCD slot (:557-560, :702-705)."""
    return torch.stack([batch_x["LA"], batch_x["LP"], batch_x["JA"], batch_x["TTCL"], batch_x["TTCD"], batch_x["TMCL"],
                        batch_x["TTCD"], batch_x["TBCL"], batch_x["TBCD"]], dim=-1).float()


def _tv_test_summary(rate, rmse_tvs, pcc_tvs, with_std=False):
    names = hostlogic.TV_NAMES
    m_rmse = {n: float(np.mean(rmse_tvs[n])) for n in names}
    m_pcc = {n: float(np.mean(pcc_tvs[n])) for n in names}
    out = {f"test_{rate}_mean_rmse": float(np.mean(list(m_rmse.values()))), f"test_{rate}_mean_pcc": float(np.mean(list(m_pcc.values())))}
    if with_std:
        out[f"test_{rate}_std_rmse"] = float(np.std(list(m_rmse.values())))
        out[f"test_{rate}_std_pcc"] = float(np.std(list(m_pcc.values())))
    for n in names:
        out[f"test_{rate}_mean_{n}_pcc"] = m_pcc[n]
    for n in names:
        out[f"test_{rate}_mean_{n}_rmse"] = m_rmse[n]
    return out


def test(model, device, vocab, exp_dir, test_spk, test_dl, rate, log_step=100, num_epochs=0) -> Dict[str, float]:
    """train/train_aptai.py:655-850, batch size 1: per-track RMSE / PCC means, FER, frame-grouped PER, overlap, boundary scores,
    keyed `test_{rate}_...` with rate in {'F', 'N'} (fast / normal speaking rate splits of the corpus).  `num_epochs` stands for
    the module-global `cfg.num_epochs` the reference passes as the epoch argument (:709)."""
    assert rate in ["F", "N"]
    names = hostlogic.TV_NAMES
    rmse_tvs, pcc_tvs = {n: [] for n in names}, {n: [] for n in names}
    overlaps, ps, rs, f1s, rvals, edit_d, n_phn = [], [], [], [], [], [], []
    total_frames = corr_frames = 0
    model.eval()
    for batch_x in test_dl:
        with torch.no_grad():
            tvs_gt = _stack_gt(batch_x)
            batch_x = {k: v.to(device) for k, v in batch_x.items()}
            outputs = model(num_epochs, **batch_x)
        tvs_gt = torch.squeeze(tvs_gt, dim=0).cpu().numpy()
        tvs_pred = torch.squeeze(outputs["tvs_pred"], dim=0).float().cpu().numpy()
        frames, corr, overlap, (p, r, f1, rval), y, yhat = _eval_frames(batch_x["phn_frames_49hz"], outputs["phn_fc_pred"])
        total_frames += frames
        corr_frames += corr
        overlaps.append(overlap)
        ps.append(p); rs.append(r); f1s.append(f1); rvals.append(rval)
        y_grp, yhat_grp = metrics.phn_frame_id2phn(y.tolist()), metrics.phn_frame_id2phn(yhat.tolist())
        edit_d.append(metrics.edit_distance(y_grp, yhat_grp))
        n_phn.append(len(y_grp))
        rm, pc = metrics.tvs_metric_rmse(tvs_gt, tvs_pred), metrics.tvs_metric_ppc(tvs_gt, tvs_pred)
        for n in names:
            rmse_tvs[n].append(rm[n])
            pcc_tvs[n].append(pc[n][0])
    out = _tv_test_summary(rate, rmse_tvs, pcc_tvs)
    out.update({f"test_{rate}_mean_FER": 1 - (corr_frames / total_frames),
                f"test_{rate}_mean_PER": float(np.sum(edit_d) / np.sum(n_phn)),
                f"test_{rate}_mean_overlap": float(np.mean(overlaps)), f"test_{rate}_mean_F1": float(np.mean(f1s)),
                f"test_{rate}_mean_p": float(np.mean(ps)), f"test_{rate}_mean_r": float(np.mean(rs)),
                f"test_{rate}_mean_Rval": float(np.mean(rvals))})
    return out


def default_cfg(**kw):
    """Hyper-parameters at the reference's argparse defaults (train/train_aptai.py:45-140)."""
    cfg = SimpleNamespace(device="cuda", num_epochs=2, batch_size=16, learning_rate=1e-5, adam_beta1=0.9, adam_beta2=0.999,
                          adam_epsilon=1e-8, adam_weight_decay=0.0, num_warmup_epochs=10, num_static_epochs=30, lr_decay=0.96,
                          target_metric="val_mean_rmse", target_metric_bigger_better=False, graphed=False, exp_dir=None,
                          vocab={f"p{i}": i for i in range(VOCAB_SIZE)}, cache_dir=None)
    cfg.__dict__.update(kw)
    return cfg


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--model_dir", default=None, help="local wav2vec2 checkpoint directory (config.json + weights)")
    ap.add_argument("--random_init", default="base", choices=["base", "large"], help="without --model_dir: random-init backbone")
    ap.add_argument("--num_epochs", type=int, default=2)
    ap.add_argument("--steps_per_epoch", type=int, default=8)
    ap.add_argument("--val_items", type=int, default=4)
    ap.add_argument("--batch_size", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--learning_rate", type=float, default=1e-5)
    ap.add_argument("--graphed", action="store_true")
    ap.add_argument("--out", default="aptai_ckpt")
    a = ap.parse_args(argv)
    w2v = W2V2Config.base(vocab_size=VOCAB_SIZE) if a.random_init == "base" else W2V2Config.large(vocab_size=VOCAB_SIZE)
    with tempfile.TemporaryDirectory() as tmp:
        model_dir = a.model_dir
        if model_dir is None:
            torch.manual_seed(0)
            Wav2Vec2Model(w2v).save_pretrained(tmp)
            model_dir = tmp
        cfg = default_cfg(num_epochs=a.num_epochs, batch_size=a.batch_size, learning_rate=a.learning_rate, graphed=a.graphed,
                          huggingface_model_id=model_dir, pretrain_cfg=w2v)
        model, optimizer, lr_scheduler = load_model_optimizer(cfg)
    train_ds = SyntheticHPRC(a.steps_per_epoch * a.batch_size, a.seconds, vary_length=True, seed=1, cfg=w2v)
    val_ds = SyntheticHPRC(a.val_items, a.seconds, vary_length=True, seed=2, cfg=w2v)
    train_dl = torch.utils.data.DataLoader(train_ds, batch_size=a.batch_size, shuffle=True, drop_last=True, collate_fn=hostlogic.collate_aptai)
    val_dl = torch.utils.data.DataLoader(val_ds, batch_size=1, shuffle=False, collate_fn=hostlogic.collate_aptai)
    return train(cfg, model, optimizer, lr_scheduler, train_dl, val_dl, "synthetic", a.out)


if __name__ == "__main__":
    main()
