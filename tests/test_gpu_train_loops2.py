"""GPU: the other two training loops of the reference on the build (protocol, not accuracy; synthetic corpora):
 * train/train_force_aptai.py:392-531 + validate + test: only the heads receive gradients, CTC-based PER, the reference's
   validation / test keys, best checkpoint that a fresh Force_APTAI loads;
 * train/train_phoneme_recognizer.py:384-486 + validate + test: random subset of batches per epoch, best / all / last checkpoint
   families incl. optimizer and scheduler state, mean_val_per / mean_test_per;
 * train/train_aptai.py:655-850 test(): the `test_{rate}_*` key set;
 * the device best-path decode against the host definition."""
import json
import os
import pickle
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VAL_KEYS = {"val_mean_loss", "val_mean_rmse", "val_mean_pcc", "val_mean_FER", "val_mean_PER", "val_mean_F1", "val_mean_p",
            "val_mean_r", "val_mean_Rval", "val_mean_overlap"}
TVN = ("LA", "LP", "JA", "TTCL", "TTCD", "TMCL", "TMCD", "TBCL", "TBCD")


def _pr_checkpoint(tmp_path, cfg, vocab, blank_bias=2.5):
    """A random-init recogniser written the way train_phoneme_recognizer.py writes its best checkpoint."""
    from aptai_amd.w2v2_pr import Wav2Vec2_PR
    from aptai_amd.wav2vec2 import Wav2Vec2Model
    torch.manual_seed(0)
    mdir = tmp_path / "w2v2"
    Wav2Vec2Model(cfg).save_pretrained(str(mdir))
    pr = Wav2Vec2_PR(cfg, None, str(mdir), vocab)
    with torch.no_grad():
        pr.pr_head.bias[0] += blank_bias                         # a trained recogniser's regime: mostly blank frames
    ck = tmp_path / "pr" / "best-model-ckpt"
    ck.mkdir(parents=True)
    torch.save(pr.state_dict(), ck / "pytorch_model.bin")
    pickle.dump({"pretrain_cfg": cfg.to_dict(), "cache_dir": None, "huggingface_model_id": str(mdir)}, open(ck / "model_cfg.pkl", "wb"))
    return str(tmp_path / "pr")


def test_force_aptai_loop_validate_test_and_checkpoint(tmp_path):
    from aptai_amd import train_force_aptai as T
    from aptai_amd.config import W2V2Config
    from aptai_amd.force_aptai import Force_APTAI
    cfg0 = T.default_cfg()
    w2v = W2V2Config.base(vocab_size=len(cfg0.vocab), num_hidden_layers=2, ctc_loss_reduction="mean", ctc_zero_infinity=True)
    pr_path = _pr_checkpoint(tmp_path, w2v, cfg0.vocab, blank_bias=3.0)
    cfg = T.default_cfg(num_epochs=2, batch_size=2, learning_rate=1e-4, pr_model_path=pr_path, num_warmup_epochs=2)
    model, opt, sched = T.load_model_optimizer(cfg)
    mk = lambda n, seed: torch.utils.data.DataLoader(T.SyntheticHPRCWithLabels(n, 1.0, seed=seed, cfg=w2v, vocab_size=40),
                                                     batch_size=2 if n > 2 else 1, drop_last=True, collate_fn=T.collate)
    tr, va = mk(6, 1), mk(2, 2)
    lines = []
    hist = T.train(cfg, model, opt, sched, tr, va, "synthetic", tmp_path / "best", log=lines.append)
    assert len(hist) == 2 and VAL_KEYS <= set(hist[0]) and hist[0]["saved"]
    assert all(np.isfinite(v) for h in hist for v in h.values() if isinstance(v, float)), hist
    assert sum("train_tv_loss" in l and "train_align_loss" in l for l in lines) == 6
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-4 * 10.0)
    # only the heads train: the frozen recogniser holds no gradient and no optimiser state
    assert all(p.grad is None for n, p in model.named_parameters() if n.startswith("w2v2_pr."))
    assert all(not n.startswith("w2v2_pr.") for n, p in model.named_parameters() if p.requires_grad)
    sd = torch.load(tmp_path / "best" / "pytorch_model.bin", weights_only=True)
    assert set(sd) == set(model.state_dict())
    fresh = Force_APTAI(pr_path, "cuda", cfg.vocab).cuda()
    fresh.load_state_dict(sd)
    res = T.test(model, "cuda", cfg.vocab, None, "synthetic", va, "N")
    want = {"test_N_mean_rmse", "test_N_std_rmse", "test_N_mean_pcc", "test_N_std_pcc", "test_N_mean_FER", "test_N_mean_PER",
            "test_N_std_PER", "test_N_mean_overlap", "test_N_std_overlap", "test_N_mean_F1", "test_N_mean_p", "test_N_mean_r",
            "test_N_mean_Rval"} | {f"test_N_mean_{n}_{m}" for n in TVN for m in ("pcc", "rmse")}
    assert set(res) == want and all(np.isfinite(v) for v in res.values())
    assert 0.0 <= res["test_N_mean_FER"] <= 1.0


def test_phoneme_recognizer_loop_subset_checkpoints_and_per(tmp_path):
    from aptai_amd import hostlogic, train_phoneme_recognizer as T
    from aptai_amd.config import W2V2Config
    from aptai_amd.wav2vec2 import Wav2Vec2Model
    vocab = T.default_vocab()
    w2v = W2V2Config.base(num_hidden_layers=2, layerdrop=0.0)
    torch.manual_seed(0)
    d = tmp_path / "w2v"
    Wav2Vec2Model(w2v).save_pretrained(str(d))
    cfg = T.default_cfg(num_epochs=2, batch_size=2, samples_per_epoch=4, learning_rate=2e-5, save_all_epochs=True,
                        huggingface_model_id=str(d), pretrain_cfg=w2v, num_warmup_epochs=2)
    model, opt, sched = T.load_model_optimizer(cfg, vocab)
    assert model.wav2vec2.config.ctc_loss_reduction == "mean" and model.wav2vec2.config.ctc_zero_infinity        # :339-342
    assert any(p.requires_grad for p in model.wav2vec2.feature_extractor.parameters())                          # conv stack trains
    tr = torch.utils.data.DataLoader(T.SyntheticCommonPhone(10, 1.0, len(vocab), seed=1), batch_size=2, drop_last=True,
                                     collate_fn=hostlogic.collate_pr)
    va = torch.utils.data.DataLoader(T.SyntheticCommonPhone(2, 1.0, len(vocab), seed=2), batch_size=1, collate_fn=hostlogic.collate_pr)
    random.seed(7)
    lines = []
    hist = T.train(cfg, model, opt, sched, vocab, tr, va, tmp_path / "best-model-ckpt", tmp_path / "last-model-ckpt",
                   tmp_path / "model-ckpts", log=lines.append)
    assert len(hist) == 2 and {"mean_val_per", "mean_val_loss"} <= set(hist[0])
    # 5 batches per epoch, samples_per_epoch / batch_size = 2 of them trained on (:406,413)
    assert all(h["trained_batches"] == 2 for h in hist) and sum(l.startswith("\tepoch") for l in lines) == 4
    for f in ("best-model-ckpt/pytorch_model.bin", "best-model-ckpt/model_cfg.pkl", "model-ckpts/e0000.bin", "model-ckpts/e0001.bin",
              "model-ckpts/model_cfg.pkl", "last-model-ckpt/optimizer.pt", "last-model-ckpt/scheduler.pt",
              "last-model-ckpt/pytorch_model.bin", "last-model-ckpt/model_cfg.pkl"):
        assert (tmp_path / f).exists(), f
    assert torch.load(tmp_path / "last-model-ckpt" / "scheduler.pt", weights_only=True) == {"last_epoch": 2}
    osd = torch.load(tmp_path / "last-model-ckpt" / "optimizer.pt", weights_only=True)
    assert len(osd["state"]) > 0 and "exp_avg" in next(iter(osd["state"].values()))
    sd = torch.load(tmp_path / "last-model-ckpt" / "pytorch_model.bin", weights_only=True)
    assert set(sd) == set(model.state_dict())
    res = T.test(model, "cuda", vocab, va, "synthetic")
    assert set(res) == {"mean_test_per"} and np.isfinite(res["mean_test_per"])
    # the checkpoint directory is what Force_APTAI consumes (models/force_aptai.py:60-75)
    from aptai_amd.config import load_model_cfg
    back = load_model_cfg(str(tmp_path / "best-model-ckpt" / "model_cfg.pkl"))
    assert set(back) == {"huggingface_model_id", "cache_dir", "pretrain_cfg"}


def test_aptai_test_function_keys(tmp_path):
    from aptai_amd import hostlogic, train_aptai as T
    from aptai_amd.config import W2V2Config
    from aptai_amd.wav2vec2 import Wav2Vec2Model
    w2v = W2V2Config.base(vocab_size=T.VOCAB_SIZE, num_hidden_layers=2)
    torch.manual_seed(0)
    d = tmp_path / "w2v"
    Wav2Vec2Model(w2v).save_pretrained(str(d))
    cfg = T.default_cfg(huggingface_model_id=str(d), pretrain_cfg=w2v)
    model, _, _ = T.load_model_optimizer(cfg)
    dl = torch.utils.data.DataLoader(T.SyntheticHPRC(3, 1.0, seed=5, cfg=w2v), batch_size=1, collate_fn=hostlogic.collate_aptai)
    res = T.test(model, "cuda", cfg.vocab, None, "synthetic", dl, "F", num_epochs=cfg.num_epochs)
    want = {"test_F_mean_rmse", "test_F_mean_pcc", "test_F_mean_FER", "test_F_mean_PER", "test_F_mean_overlap", "test_F_mean_F1",
            "test_F_mean_p", "test_F_mean_r", "test_F_mean_Rval"} | {f"test_F_mean_{n}_{m}" for n in TVN for m in ("pcc", "rmse")}
    assert set(res) == want and all(np.isfinite(v) for v in res.values())
    assert not model.training
    with pytest.raises(AssertionError):
        T.test(model, "cuda", cfg.vocab, None, "synthetic", dl, "X")


def test_device_best_path_decode_matches_the_host_definition():
    """aptai_ctc_greedy_decode vs argmax -> collapse repeats -> drop blank on the host (oracle.heads_ref.ctc_best_path), incl. rows
    longer than one 64-frame round, runs that straddle a round boundary, ties (first maximum) and an over-long result."""
    from aptai_amd import ops
    from oracle import heads_ref
    g = torch.Generator().manual_seed(2)
    B, T, V, Np, Tp = 5, 499, 40, 64, 512
    logits = torch.randn(B, Tp, Np, generator=g)
    logits[0, :, 0] += 3.0                                     # mostly blank
    logits[1, 60:70, :] = 0.0
    logits[1, 60:70, 7] = 5.0                                  # one run across the round boundary at frame 64
    logits[2, :, 3] = logits[2, :, 5] = 9.0                    # ties: the first maximum (3) wins everywhere -> one label
    logits[3] = torch.randn(Tp, Np, generator=g) * 5           # ~T distinct labels: longer than max_n
    ids, n = ops.ctc_greedy_decode(logits.cuda().contiguous(), Np, Tp, B, T, V, 0, 60)
    ids, n = ids.cpu().numpy(), n.cpu().numpy()
    for b in range(B):
        ref = heads_ref.ctc_best_path(logits[b, :T, :V].numpy(), blank=0)
        assert n[b] == len(ref), (b, n[b], len(ref))
        m = min(len(ref), 60)
        assert list(ids[b, :m]) == list(ref[:m]) and (ids[b, m:] == 0).all()
    assert list(ids[2, :n[2]]) == [3] and n[3] > 60


def test_phoneme_recognizer_loop_replays_graphs_on_the_reference_collate(tmp_path):
    """train_phoneme_recognizer.train with cfg.graphed: the reference's collate (per-batch padding of waveforms and label lists,
    train/train_phoneme_recognizer.py:224-239) feeds BucketedGraphedStep; the trainable conv stack, CTC head and optimiser run as
    replayed segments.  With the regularisers off the epoch's losses equal the eager loop's from the same initial state.

    Round 3 this test was RED (step-8 losses 6.7262 vs 6.763).  Cause, established in round 4 by A/B builds (tools/pr_loop_ab.py,
    profiles/r04_pr_loop_cause.txt): the bucketed graphs summed the first conv layer's GroupNorm moments over the BUCKET's frames, and frame
    T(S) of the zero-padded waveform still covers 5..9 real samples - one frame the reference does not have, ~1/T0 relative in the
    statistics, amplified by Adam over 8 steps.  (The CTC gradient's float atomics were NOT it; they are order-fixed now all the same.)
    With the sums clamped to the collated frame count the step-1 gradients of every parameter are bit-identical to the eager step's
    (tools/pr_graph_vs_eager.py, profiles/r04_pr_graph_vs_eager_step1.txt) and the two 8-step traces agree in every printed digit, so
    the bound is the print resolution, not a trajectory tolerance."""
    from aptai_amd import hostlogic, train_phoneme_recognizer as T
    from aptai_amd.config import W2V2Config
    from aptai_amd.wav2vec2 import Wav2Vec2Model
    vocab = T.default_vocab()
    w2v = W2V2Config.base(num_hidden_layers=2, layerdrop=0.0, hidden_dropout=0., activation_dropout=0., attention_dropout=0.,
                          feat_proj_dropout=0., apply_spec_augment=False)
    torch.manual_seed(0)
    d = tmp_path / "w2v"
    Wav2Vec2Model(w2v).save_pretrained(str(d))
    logs = {}
    for graphed in (False, True):
        cfg = T.default_cfg(num_epochs=2, batch_size=2, samples_per_epoch=8, learning_rate=2e-5, save_all_epochs=False, final_dropout=0.0,
                            huggingface_model_id=str(d), pretrain_cfg=w2v, num_warmup_epochs=2, graphed=graphed)
        torch.manual_seed(3)                                            # pr_head is freshly initialised (not in the wav2vec2 checkpoint)
        model, opt, sched = T.load_model_optimizer(cfg, vocab)
        tr = torch.utils.data.DataLoader(T.SyntheticCommonPhone(8, 1.2, len(vocab), seed=1), batch_size=2, drop_last=True,
                                         collate_fn=hostlogic.collate_pr)
        va = torch.utils.data.DataLoader(T.SyntheticCommonPhone(2, 1.0, len(vocab), seed=2), batch_size=1, collate_fn=hostlogic.collate_pr)
        random.seed(7)
        lines = []
        sub = tmp_path / ("g" if graphed else "e")
        hist = T.train(cfg, model, opt, sched, vocab, tr, va, sub / "best", sub / "last", sub / "all", log=lines.append)
        logs[graphed] = ([float(l.split("train_loss:")[1]) for l in lines if l.startswith("\tepoch")], hist)
    le, lg = logs[False][0], logs[True][0]
    assert len(le) == len(lg) == 8
    print(f"[bands] PR loop graph vs eager: largest relative loss deviation over 8 steps {max(abs(a - b) / abs(a) for a, b in zip(le, lg)):.2e}")
    for a, b in zip(le, lg):
        assert abs(a - b) <= 1.5e-4 + 1e-5 * abs(a), (le, lg)              # losses are logged with 4 decimals
    for he, hg in zip(logs[False][1], logs[True][1]):
        assert abs(he["mean_val_loss"] - hg["mean_val_loss"]) <= 1e-4 * abs(he["mean_val_loss"])
