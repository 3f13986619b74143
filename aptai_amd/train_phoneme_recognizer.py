"""The CTC phoneme-recogniser fine-tuning loop of the reference (train/train_phoneme_recognizer.py) on the MI355X build: same
function surface (`load_model_optimizer`, `train`, `validate`, `test`), per-batch protocol (`model(**batch_x)` ->
`loss.backward()` -> `optimizer.step()`, everything trainable unless `freeze_feature_extractor`), the reference's RANDOM SUBSET of
batches per epoch (:406,413), LambdaLR schedule, PER metric, and the three checkpoint families it writes (:472-486):
`best-model-ckpt/`, `model-ckpts/e%04d.bin` (with `save_all_epochs`) and `last-model-ckpt/` incl. optimizer / scheduler state.
Decoding for the PER is the best path (the torchaudio beam decoder of utility.py:448-471 is absent: parity unpinned).
The CommonPhone reader, wandb and resume-from-hub are out of scope; `SyntheticCommonPhone` yields items
with the fields `_collator` consumes.  The shipped script's stale imports / constructor arity are not
reproduced: the model is `aptai_amd.w2v2_pr.Wav2Vec2_PR(pretrain_cfg, cache_dir, huggingface_model_id, vocab)`.

    python -m aptai_amd.train_phoneme_recognizer --random_init base --num_epochs 2 --samples_per_epoch 64 --batch_size 16
"""
from __future__ import annotations

import argparse
import pickle
import random
import tempfile
from pathlib import Path
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch

from . import hostlogic, metrics
from .config import W2V2Config
from .w2v2_pr import Wav2Vec2_PR
from .wav2vec2 import Wav2Vec2Model


def default_vocab(n: int = 40) -> dict:
    vocab = {"(blank)": 0, "(...)": 1}
    vocab.update({f"p{i}": i for i in range(2, n)})
    return vocab


class SyntheticCommonPhone(torch.utils.data.Dataset):
    """Items shaped like data/dataset_commonphone.py's (audio, audio_len, phoneme_label): N(0,1) 16 kHz audio (optionally the
    reference's 1-second crop), 20..55 label ids in [1, V-1] (scaled down for clips that hold fewer frames)."""

    def __init__(self, n_items: int, seconds: float = 10.0, vocab_size: int = 40, vary_length: bool = True, seed: int = 0):
        self.n, self.S, self.V, self.vary, self.seed = n_items, int(16000 * seconds), vocab_size, vary_length, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = np.random.RandomState(self.seed * 100003 + i)
        n = self.S if (not self.vary or i % 2 == 0) else int(g.randint(int(0.8 * self.S), self.S + 1))
        frames = max(n // 320 - 1, 2)
        hi = max(2, min(55, frames // 3))
        lo = max(1, min(20, hi - 1))
        return {"audio": g.randn(n).astype(np.float32), "audio_len": n,
                "phoneme_label": g.randint(1, self.V, size=int(g.randint(lo, hi + 1))).astype(np.int32)}


def load_model_optimizer(args_cfg, vocab):
    """train/train_phoneme_recognizer.py:322-379: config edits (:339-342), model on a LOCAL wav2vec2 directory, Adam over ALL
    parameters, LambdaLR."""
    pretrain_cfg = W2V2Config.from_any(args_cfg.pretrain_cfg)
    pretrain_cfg.vocab_size = len(vocab)
    pretrain_cfg.final_dropout = args_cfg.final_dropout
    if getattr(args_cfg, "num_hidden_layers", None):
        pretrain_cfg.num_hidden_layers = args_cfg.num_hidden_layers
    pretrain_cfg.ctc_loss_reduction = "mean"
    pretrain_cfg.ctc_zero_infinity = True
    pretrain_cfg.blank = 0
    model = Wav2Vec2_PR(pretrain_cfg, getattr(args_cfg, "cache_dir", None), args_cfg.huggingface_model_id, vocab).to(args_cfg.device)
    if getattr(args_cfg, "freeze_feature_extractor", False):
        model.freeze_feature_encoder()
    from .optim import Adam
    optimizer = Adam(model.parameters(), lr=args_cfg.learning_rate, betas=(args_cfg.adam_beta1, args_cfg.adam_beta2),
                     eps=args_cfg.adam_epsilon, weight_decay=args_cfg.adam_weight_decay).publish_to(model)
    lr_scheduler = torch.optim.lr_scheduler.LambdaLR(
        optimizer=optimizer, lr_lambda=hostlogic.get_lr_schedule(args_cfg.num_warmup_epochs, args_cfg.num_static_epochs, args_cfg.lr_decay))
    return model, optimizer, lr_scheduler


def train(cfg, model, optimizer, lr_scheduler, vocab, train_dataloader, valid_dataloader, best_ckpt_path, last_ckpt_path,
          all_ckpt_path, log=print):
    """train/train_phoneme_recognizer.py:384-505.  Returns the per-epoch log dicts."""
    eval_target = None
    history = []
    runner = None
    best_ckpt_path, last_ckpt_path, all_ckpt_path = Path(best_ckpt_path), Path(last_ckpt_path), Path(all_ckpt_path)
    best_ckpt_path.mkdir(parents=True, exist_ok=True)
    last_ckpt_path.mkdir(parents=True, exist_ok=True)
    if cfg.save_all_epochs:
        all_ckpt_path.mkdir(parents=True, exist_ok=True)
    for epoch in range(cfg.num_epochs):
        epoch_train_steps = int(cfg.samples_per_epoch / cfg.batch_size)
        # a random subset of this epoch's batches is trained on, the others are skipped (:406,413); `random` is the module the
        # reference draws from, so `random.seed` reproduces an epoch's subset
        subset_random = set(random.sample(range(len(train_dataloader)), epoch_train_steps))
        subset_random_idx, sum_train_loss = 0, 0.0
        model.train()
        for batch_idx, batch_x in enumerate(train_dataloader):
            if batch_idx not in subset_random:
                continue
            if getattr(cfg, "graphed", False):
                # hipGraph replay of the same step; the collate pads each batch to its own longest utterance and label list
                # (train/train_phoneme_recognizer.py:224-239), so one captured runner per (batch size, length bucket, label width)
                if runner is None:
                    from .graphed import BucketedGraphedStep
                    runner = BucketedGraphedStep(model, optimizer)
                outputs = runner.step({k: v.to(cfg.device) for k, v in batch_x.items()})
            else:
                batch_x = {k: v.to(cfg.device) for k, v in batch_x.items()}
                optimizer.zero_grad()
                outputs = model(**batch_x)
                outputs["loss"].backward()
                optimizer.step()
            sum_train_loss += float(outputs["loss"].detach())
            log(f"\tepoch {epoch + 1} ~ batch {subset_random_idx + 1}/{epoch_train_steps}, train_loss: {float(outputs['loss'].detach()):.4f}")
            subset_random_idx += 1
        lr_scheduler.step()
        if runner is not None:
            runner.suspend()             # the eager validation below rebuilds its weight copies; the captured buckets stay
        model.eval()
        val_logs = validate(model, cfg.device, vocab, epoch, valid_dataloader)
        better = (eval_target is None
                  or (cfg.target_metric_bigger_better and eval_target <= val_logs[cfg.target_metric])
                  or (not cfg.target_metric_bigger_better and eval_target >= val_logs[cfg.target_metric]))
        if better:
            eval_target = val_logs[cfg.target_metric]
            torch.save(model.state_dict(), best_ckpt_path / "pytorch_model.bin")
            pickle.dump(model.get_config(), open(best_ckpt_path / "model_cfg.pkl", "wb"))
        if cfg.save_all_epochs:
            torch.save(model.state_dict(), all_ckpt_path / f"e{epoch:04d}.bin")
            if not (all_ckpt_path / "model_cfg.pkl").exists():
                pickle.dump(model.get_config(), open(all_ckpt_path / "model_cfg.pkl", "wb"))
        torch.save(optimizer.state_dict(), last_ckpt_path / "optimizer.pt")
        torch.save({"last_epoch": cfg.num_epochs}, last_ckpt_path / "scheduler.pt")          # as written (:484)
        torch.save(model.state_dict(), last_ckpt_path / "pytorch_model.bin")
        pickle.dump(model.get_config(), open(last_ckpt_path / "model_cfg.pkl", "wb"))
        epoch_log = dict(val_logs, epoch=epoch, mean_train_loss=sum_train_loss / max(epoch_train_steps, 1),
                         lr=optimizer.param_groups[0]["lr"], saved=bool(better), trained_batches=subset_random_idx)
        history.append(epoch_log)
        log(f"Epoch {epoch + 1}/{cfg.num_epochs} -> lr: {epoch_log['lr']}| mean_train_loss: {epoch_log['mean_train_loss']}| "
            f"mean_val_loss: {val_logs['mean_val_loss']}| val_per: {val_logs['mean_val_per']}")
    if runner is not None:
        runner.close()
    return history


def _decode(model, outputs) -> list:
    """Stand-in for `_ctc_decode(vocab, phoneme_logits)` (utility.py:448-471): best path over all frames of the batch-1 logits
    (the device decode kernel reads the fp32 logits the forward just produced)."""
    lg = outputs["phoneme_logits"].float().contiguous()
    B, T, V = lg.shape
    from . import ops
    ids, n = ops.ctc_greedy_decode(lg, V, T, B, T, V, model._blank(), T)
    return [int(i) for i in ids[0, :int(n[0])].cpu().numpy()]


def validate(model, device, vocab, epoch, validate_dataloader, log_step=100) -> Dict[str, float]:
    """train/train_phoneme_recognizer.py:509-561, batch size 1."""
    val_losses, edit_d, n_phn = [], [], []
    for batch_x in validate_dataloader:
        with torch.no_grad():
            phoneme_label = batch_x["phoneme_labels"].numpy()[0]
            batch_x = {k: v.to(device) for k, v in batch_x.items()}
            outputs = model(**batch_x)
        val_losses.append(outputs["loss"].item())
        edit_d.append(metrics.edit_distance(phoneme_label, _decode(model, outputs)))
        n_phn.append(len(phoneme_label))
    return {"mean_val_per": float(np.sum(edit_d) / np.sum(n_phn)), "mean_val_loss": float(np.mean(val_losses))}


def test(model, device, vocab, test_dl, dataset_name, log_step=100, laptop=False) -> Dict[str, float]:
    """train/train_phoneme_recognizer.py:566-617."""
    edit_d, n_phn = [], []
    model.eval()
    for batch_idx, batch_x in enumerate(test_dl):
        if laptop and batch_idx >= 1:
            break
        with torch.no_grad():
            phoneme_label = batch_x["phoneme_labels"].numpy()[0]
            batch_x = {k: v.to(device) for k, v in batch_x.items()}
            outputs = model(**batch_x)
        edit_d.append(metrics.edit_distance(phoneme_label, _decode(model, outputs)))
        n_phn.append(len(phoneme_label))
    return {"mean_test_per": float(np.sum(edit_d) / np.sum(n_phn))}


def default_cfg(**kw):
    """Hyper-parameters at the reference's argparse defaults / start_train_phoneme_recognizer.sh (bs 2, lr 5e-6)."""
    cfg = SimpleNamespace(device="cuda", num_epochs=2, batch_size=2, samples_per_epoch=8, learning_rate=5e-6, adam_beta1=0.9,
                          adam_beta2=0.999, adam_epsilon=1e-8, adam_weight_decay=0.0, num_warmup_epochs=10, num_static_epochs=30,
                          lr_decay=0.96, target_metric="mean_val_per", target_metric_bigger_better=False, final_dropout=0.1,
                          num_hidden_layers=None, freeze_feature_extractor=False, save_all_epochs=False, cache_dir=None)
    cfg.__dict__.update(kw)
    return cfg


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--model_dir", default=None, help="local wav2vec2 checkpoint directory (config.json + weights)")
    ap.add_argument("--random_init", default="base", choices=["base", "large"])
    ap.add_argument("--num_epochs", type=int, default=2)
    ap.add_argument("--samples_per_epoch", type=int, default=64)
    ap.add_argument("--train_items", type=int, default=128)
    ap.add_argument("--val_items", type=int, default=4)
    ap.add_argument("--batch_size", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--learning_rate", type=float, default=5e-6)
    ap.add_argument("--save_all_epochs", action="store_true")
    ap.add_argument("--out", default="pr_exp")
    a = ap.parse_args(argv)
    vocab = default_vocab()
    w2v = W2V2Config.base() if a.random_init == "base" else W2V2Config.large()
    with tempfile.TemporaryDirectory() as tmp:
        model_dir = a.model_dir
        if model_dir is None:
            torch.manual_seed(0)
            Wav2Vec2Model(w2v).save_pretrained(tmp)
            model_dir = tmp
        cfg = default_cfg(num_epochs=a.num_epochs, batch_size=a.batch_size, samples_per_epoch=a.samples_per_epoch,
                          learning_rate=a.learning_rate, save_all_epochs=a.save_all_epochs, huggingface_model_id=model_dir,
                          pretrain_cfg=w2v)
        model, optimizer, lr_scheduler = load_model_optimizer(cfg, vocab)
    tr = torch.utils.data.DataLoader(SyntheticCommonPhone(a.train_items, a.seconds, len(vocab), seed=1), batch_size=a.batch_size,
                                     shuffle=True, drop_last=True, collate_fn=hostlogic.collate_pr)
    va = torch.utils.data.DataLoader(SyntheticCommonPhone(a.val_items, a.seconds, len(vocab), seed=2), batch_size=1,
                                     collate_fn=hostlogic.collate_pr)
    out = Path(a.out)
    return train(cfg, model, optimizer, lr_scheduler, vocab, tr, va, out / "best-model-ckpt", out / "last-model-ckpt", out / "model-ckpts")


if __name__ == "__main__":
    main()
