"""GPU: the reference's train()/validate() loop surface (train/train_aptai.py:392-652) on the build, synthetic corpus.
Checks the protocol, not accuracy: finite decreasing-ish losses, the reference's validation keys, LambdaLR schedule applied
per epoch, best-checkpoint files that load back into a fresh model with identical state-dict keys, eager == graphed driver."""
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VAL_KEYS = {"val_mean_loss", "val_mean_rmse", "val_mean_pcc", "val_mean_FER", "val_mean_PER", "val_mean_F1", "val_mean_p",
            "val_mean_r", "val_mean_Rval", "val_mean_overlap"}


def _setup(tmp_path, graphed):
    from aptai_amd import hostlogic, train_aptai as T
    from aptai_amd.config import W2V2Config
    from aptai_amd.wav2vec2 import Wav2Vec2Model
    w2v = W2V2Config.base(vocab_size=T.VOCAB_SIZE, num_hidden_layers=2, layerdrop=0.0)
    torch.manual_seed(0)
    d = tmp_path / "w2v"
    Wav2Vec2Model(w2v).save_pretrained(str(d))
    cfg = T.default_cfg(num_epochs=2, batch_size=2, learning_rate=2e-5, graphed=graphed, huggingface_model_id=str(d), pretrain_cfg=w2v,
                        num_warmup_epochs=2)
    model, opt, sched = T.load_model_optimizer(cfg)
    tr = torch.utils.data.DataLoader(T.SyntheticHPRC(6, 1.0, vary_length=not graphed, seed=1, cfg=w2v), batch_size=2, drop_last=True,
                                     collate_fn=hostlogic.collate_aptai)
    va = torch.utils.data.DataLoader(T.SyntheticHPRC(2, 1.0, seed=2, cfg=w2v), batch_size=1, collate_fn=hostlogic.collate_aptai)
    return T, cfg, model, opt, sched, tr, va


@pytest.mark.parametrize("graphed", [False, True])
def test_train_validate_checkpoint_round_trip(tmp_path, graphed):
    T, cfg, model, opt, sched, tr, va = _setup(tmp_path, graphed)
    lines = []
    hist = T.train(cfg, model, opt, sched, tr, va, "synthetic", tmp_path / "best", log=lines.append)
    assert len(hist) == 2 and VAL_KEYS <= set(hist[0])
    assert all(np.isfinite(v) for h in hist for v in h.values() if isinstance(v, float)), hist
    assert hist[0]["saved"]                                           # first epoch always becomes the best checkpoint
    # LambdaLR: 10 * (epoch + 1) / warmup  ->  after epoch 0 the multiplier is 10 * 2 / 2
    assert opt.param_groups[0]["lr"] == pytest.approx(2e-5 * 10.0)
    assert sum(l.startswith("\tepoch") for l in lines) == 6
    sd = torch.load(tmp_path / "best" / "pytorch_model.bin", weights_only=True)
    assert set(sd) == set(model.state_dict())
    assert "tv_lowpass.lowpass.weight" in sd and sd["tv_lowpass.lowpass.weight"].dtype == torch.float64
    cfg_back = pickle.load(open(tmp_path / "best" / "model_cfg.pkl", "rb"))      # our own file
    assert set(cfg_back) == {"device", "vocab", "huggingface_model_id", "pretrain_cfg"}
    fresh, _, _ = T.load_model_optimizer(cfg)
    fresh.load_state_dict(sd)
