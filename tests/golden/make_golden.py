#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in the build container.

Runs only where /root/reference exists (never on the GPU box, never from pytest).  The reference
ships no fixtures of its own, so these outputs are what pins the oracle.

How the reference is imported: flat imports with sys.path = [reference/models,
reference]; the corpus/logging packages it imports at module scope but never touches on the hot
path (editdistance, wandb, textgrids, librosa, torchaudio) are absent from the image and are
represented by empty stand-in modules.  The one absent piece that IS called on the Force_APTAI path
— torchaudio's lexicon-free CTC beam decoder (models/w2v2_pr.py:144-155) — is replaced by a greedy
best-path stand-in; its output is stored as an INPUT fixture and that step is "parity unpinned".

Weights: pretrained checkpoints are unreachable offline; every case loads oracle.synth tensors
(pure functions of name/shape/seed) into the unmodified reference modules.

Usage:  python tests/golden/make_golden.py [case ...]
"""
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import transformers  # noqa: E402  (must precede the stand-ins: its availability probes reject spec-less stubs)
from transformers import Wav2Vec2Config, Wav2Vec2Model  # noqa: E402
from transformers.models.wav2vec2 import modeling_wav2vec2 as hf_w2v2  # noqa: E402

from oracle import synth  # noqa: E402
from oracle.heads_ref import ctc_best_path  # noqa: E402


def _install_standins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    mod("editdistance", eval=None)
    mod("wandb")
    mod("textgrids")
    lib = mod("librosa")
    lib.filters = mod("librosa.filters", mel=None)
    lib.sequence = mod("librosa.sequence", dtw=None)

    class _Hyp:
        def __init__(self, tokens):
            self.tokens = torch.as_tensor(tokens)
            self.timesteps = torch.zeros(len(tokens), dtype=torch.int32)

    def ctc_decoder(lexicon=None, tokens=None, lm=None, nbest=1, beam_size=10, beam_size_token=None,
                    beam_threshold=50, blank_token='(blank)', sil_token='(...)', **kw):
        blank = list(tokens).index(blank_token)

        def run(emissions):
            return [[_Hyp(ctc_best_path(e.numpy(), blank))] for e in emissions]
        return run

    ta = mod("torchaudio")
    ta.models = mod("torchaudio.models")
    ta.models.decoder = mod("torchaudio.models.decoder", ctc_decoder=ctc_decoder)


_install_standins()
sys.path[:0] = [os.path.join(REF, "models"), REF]
import aptai as ref_aptai          # noqa: E402
import modules as ref_modules      # noqa: E402
import w2v2_pr as ref_w2v2_pr      # noqa: E402
import force_aptai as ref_force    # noqa: E402

VERSIONS = f"torch {torch.__version__}; transformers {transformers.__version__}; numpy {np.__version__}"
TV = ("LA", "LP", "JA", "TTCL", "TTCD", "TMCL", "TMCD", "TBCL", "TBCD")


def hf_config(cfg_kw, **extra):
    kw = dict(cfg_kw)
    kw.update(extra)
    return Wav2Vec2Config(**kw)


def local_model_dir(cfg, tmp):
    """Random-init HF checkpoint in a temp dir, so the reference's from_pretrained(<path>) works offline."""
    d = os.path.join(tmp, "w2v2")
    torch.manual_seed(0)
    Wav2Vec2Model(cfg).save_pretrained(d)
    return d


def grads_summary(named_params, picks):
    out = {}
    for n, p in named_params:
        if p.grad is None:
            continue
        g = p.grad.detach().float()
        out[f"gnorm/{n}"] = np.float64(g.double().norm().item())
        if n in picks:
            flat = g.flatten()
            step = max(1, flat.numel() // 512)
            out[f"gslice/{n}"] = flat[::step][:512].numpy().copy()
    return out


NOREG = dict(hidden_dropout=0.0, activation_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0,
             final_dropout=0.0, layerdrop=0.0, apply_spec_augment=False)

BASE = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)
LARGE = dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
             feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)


def save(name, meta, arrays):
    arrays = {k: (np.asarray(v)) for k, v in arrays.items()}
    arrays["__meta__"] = np.array(repr(dict(meta, versions=VERSIONS, generator="tests/golden/make_golden.py")))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.0f} KiB)")


# ============================================================================= cases
def case_pr_base(name="pr_base_2x4s", layers=12, seconds=4.0, seed=0):
    """BASELINE config 1: Wav2Vec2_PR, wav2vec2-base, batch 2 x 4 s, vocab 40, CTC mean/zero_infinity."""
    cfg_kw = dict(BASE, num_hidden_layers=layers, vocab_size=40, ctc_loss_reduction="mean",
                  ctc_zero_infinity=True, blank=0, **NOREG)
    cfg = hf_config(cfg_kw)
    S = int(16000 * seconds)
    batch = synth.synth_pr_batch(cfg, 2, S, seed=1234, lo=8 if seconds < 2 else 20, hi=12 if seconds < 2 else 55)
    vocab = {f"p{i}": i for i in range(40)}
    with tempfile.TemporaryDirectory() as tmp:
        model = ref_w2v2_pr.Wav2Vec2_PR(cfg, None, local_model_dir(cfg, tmp), vocab)
    sd = synth.make_state_dict(synth.pr_param_shapes(cfg), seed)
    model.load_state_dict(sd)
    arrays = {}
    model.train()
    out = model(**batch)
    out["loss"].backward()
    arrays["train/loss"] = out["loss"].detach().numpy()
    arrays["train/phoneme_logits"] = out["phoneme_logits"].detach().numpy()
    arrays["train/log_probs"] = out["log_probs"].detach().numpy()
    arrays["train/hidden_states_sub"] = out["hidden_states"].detach().numpy()[:, ::7, ::16]
    picks = {"pr_head.weight", "pr_head.bias", "wav2vec2.encoder.layers.0.attention.q_proj.weight",
             f"wav2vec2.encoder.layers.{layers-1}.feed_forward.output_dense.weight",
             "wav2vec2.encoder.layer_norm.weight", "wav2vec2.feature_projection.projection.weight",
             "wav2vec2.feature_extractor.conv_layers.0.conv.weight",
             "wav2vec2.feature_extractor.conv_layers.0.layer_norm.weight",
             "wav2vec2.feature_extractor.conv_layers.3.conv.weight",
             "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0",
             "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1",
             "wav2vec2.encoder.pos_conv_embed.conv.bias"}
    arrays.update(grads_summary(model.named_parameters(), picks))
    model.eval()
    with torch.no_grad():
        out = model(**batch)
    arrays["eval/loss"] = out["loss"].numpy()
    arrays["eval/phoneme_logits"] = out["phoneme_logits"].numpy()
    # SpecAugment semantics: train mode, np.random seeded, everything else still 0
    cfg2 = hf_config(cfg_kw, apply_spec_augment=True, mask_time_prob=0.05)
    model.wav2vec2.config.apply_spec_augment = True
    model.train()
    np.random.seed(4321)
    with torch.no_grad():
        out = model(**batch)
    arrays["specaug/np_seed"] = np.int64(4321)
    arrays["specaug/loss"] = out["loss"].numpy()
    arrays["specaug/phoneme_logits_sub"] = out["phoneme_logits"].numpy()[:, ::3]
    for k, v in batch.items():
        arrays["in/" + k] = v.numpy()
    save(name, dict(case=name, cfg=cfg_kw, seed=seed, S=S, batch_seed=1234, model="Wav2Vec2_PR",
                    note="train/* and g*/: train mode with all stochastic regularisers at 0"), arrays)


def case_pr_base_mini():
    case_pr_base(name="pr_base_mini_2x1s", layers=2, seconds=1.0)


def case_aptai_large(name="aptai_large_2x1s", seconds=1.0, seed=0):
    """APTAI as shipped: wav2vec2-large shape (models/aptai.py hard-codes 1024 / hidden_states[24])."""
    cfg_kw = dict(LARGE, vocab_size=46, **NOREG)
    cfg = hf_config(cfg_kw)
    S = int(16000 * seconds)
    batch = synth.synth_aptai_batch(cfg, 2, S, seed=1234)
    vocab = {f"p{i}": i for i in range(46)}
    with tempfile.TemporaryDirectory() as tmp:
        model = ref_aptai.APTAI("cpu", vocab, local_model_dir(cfg, tmp), cfg, None)
    sd = synth.make_state_dict(synth.aptai_param_shapes(cfg), seed)
    model.load_state_dict(sd)
    # the reference module uses Dropout(0.1) in its heads: zero them for the deterministic fixture
    model.tv_head[0].p = 0.0
    model.phn_head[0].p = 0.0
    arrays = {}
    model.train()
    hs = {}
    hook = model.wav2vec2.register_forward_hook(lambda m, i, o: hs.update(h=o.hidden_states))
    out = model(0, **batch)
    hook.remove()
    out["loss"].backward()
    for k in ("loss", "mse_loss", "ce_loss", "tvs_pred", "phn_fc_pred"):
        arrays["train/" + k] = out[k].detach().numpy()
    for i in (0, 1, 12, 24):
        arrays[f"train/hidden_{i}_sub"] = hs["h"][i].detach().numpy()[:, ::4, ::16]
    picks = {"tv_head.2.weight", "tv_head.2.bias", "phn_head.2.weight", "phn_head.2.bias",
             "wav2vec2.encoder.layers.0.attention.q_proj.weight", "wav2vec2.encoder.layers.0.layer_norm.weight",
             "wav2vec2.encoder.layers.23.feed_forward.output_dense.weight", "wav2vec2.encoder.layer_norm.weight",
             "wav2vec2.feature_projection.projection.weight", "wav2vec2.feature_projection.layer_norm.weight",
             "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0",
             "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1"}
    arrays.update(grads_summary(model.named_parameters(), picks))
    arrays["frozen_has_grad"] = np.array(
        [p.grad is not None for n, p in model.named_parameters() if "feature_extractor" in n])
    model.eval()
    with torch.no_grad():
        out = model(0, **batch)
    for k in ("loss", "tvs_pred", "phn_fc_pred"):
        arrays["eval/" + k] = out[k].numpy()
    for k, v in batch.items():
        arrays["in/" + k] = v.numpy()
    save(name, dict(case=name, cfg=cfg_kw, seed=seed, S=S, batch_seed=1234, model="APTAI"), arrays)


def case_force(name="force_aptai_1x2s", seed=0):
    """Force_APTAI.forward at B=1 (the only batch size the shipped RNN.forward can run) + the B=2
    sub-module vectors that pin CrossAttention / ForwardSumLoss / LSTM / MLP / LowPass."""
    pr_kw = dict(LARGE, num_hidden_layers=2, vocab_size=40, ctc_loss_reduction="mean", ctc_zero_infinity=True,
                 blank=0, **NOREG)
    pr_cfg = hf_config(pr_kw)
    vocab = {"(blank)": 0, "(...)": 1}
    vocab.update({f"p{i}": i for i in range(2, 40)})
    S = 32000
    arrays = {}
    with tempfile.TemporaryDirectory() as tmp:
        mdir = local_model_dir(pr_cfg, tmp)
        pr = ref_w2v2_pr.Wav2Vec2_PR(pr_cfg, None, mdir, vocab)
        shapes = synth.force_aptai_param_shapes(pr_cfg, len(vocab))
        sd = synth.make_state_dict(shapes, seed)
        sd["w2v2_pr.pr_head.bias"][0] += 2.5        # favour blank so the greedy decode stays < 60 phonemes
        pr.load_state_dict({k[len("w2v2_pr."):]: v for k, v in sd.items() if k.startswith("w2v2_pr.")})
        ck = os.path.join(tmp, "pr", "best-model-ckpt")
        os.makedirs(ck)
        torch.save(pr.state_dict(), os.path.join(ck, "pytorch_model.bin"))
        pickle.dump({"pretrain_cfg": pr_cfg, "cache_dir": None, "huggingface_model_id": mdir},
                    open(os.path.join(ck, "model_cfg.pkl"), "wb"))
        model = ref_force.Force_APTAI(os.path.join(tmp, "pr"), "cpu", vocab)
    model.load_state_dict(sd)
    model.frame_drop.p = 0.0
    model.pe_phn.dropout.p = 0.0
    model.rnn.linear[1].p = 0.0
    batch = synth.synth_aptai_batch(pr_cfg, 1, S, seed=99, n_phn=40)
    batch["phoneme_labels"] = synth.synth_ctc_labels(1, 40, 99)
    model.train()
    out = model(0, **batch)
    out["loss"].backward()
    for k in ("loss", "tv_loss", "align_loss", "tvs_pred"):
        arrays["b1/" + k] = out[k].detach().numpy()
    arrays["b1/pred_frame_phns"] = np.array(out["pred_frame_phns"][0], dtype=np.int64)
    arrays["b1/pred_ctc_phn_seq"] = np.asarray(out["pred_ctc_phn_seq"][0], dtype=np.int64)
    picks = {n for n, p in model.named_parameters() if p.requires_grad}
    arrays.update({"b1/" + k: v for k, v in grads_summary(model.named_parameters(), picks).items()})
    for k, v in batch.items():
        arrays["b1/in/" + k] = v.numpy()
    # ---- B=2 sub-module vectors
    g = torch.Generator().manual_seed(5)
    T, N = 99, 60
    frame = torch.randn(2, T, 128, generator=g)
    phn_ids = torch.zeros(2, N, dtype=torch.int32)
    phn_ids[0, :37] = torch.randint(1, 40, (37,), generator=g, dtype=torch.int32)
    phn_ids[1, :21] = torch.randint(1, 40, (21,), generator=g, dtype=torch.int32)
    mask = (phn_ids != 0).to(torch.int)
    with torch.no_grad():
        emb = model.pe_phn(model.phn_emb_layer(phn_ids).permute(1, 0, 2)).permute(1, 0, 2)
        att_out, energy = model.xatt(frame, emb, mask)
        att = torch.log_softmax(energy + ((1 - mask) * -1000.0).unsqueeze(1).repeat(1, T, 1), dim=-1)
        fs = model.align_loss(att.unsqueeze(1), [37, 21], [T, 80])
        from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
        packed = pack_padded_sequence(att_out, [T, 80], batch_first=True, enforce_sorted=False)
        po, _ = model.rnn.lstm(packed)
        lstm_out, _ = pad_packed_sequence(po, batch_first=True)
        rnn_out = model.rnn.linear(lstm_out)
        tvs = model.tv_lowpass(rnn_out)
    arrays.update({"b2/frame": frame.numpy(), "b2/phn_ids": phn_ids.numpy(), "b2/phn_embs": emb.numpy(),
                   "b2/att_out": att_out.numpy(), "b2/energy": energy.numpy(), "b2/att": att.numpy(),
                   "b2/align_idx": torch.max(att, axis=2)[1].numpy(), "b2/fs_loss": fs.numpy(),
                   "b2/lstm_out": lstm_out.numpy(), "b2/rnn_out": rnn_out.numpy(), "b2/tvs": tvs.numpy(),
                   "b2/text_lens": np.array([37, 21]), "b2/mel_lens": np.array([T, 80])})
    save(name, dict(case=name, pr_cfg=pr_kw, seed=seed, vocab_len=len(vocab), model="Force_APTAI", blank_bias=2.5,
                    note="decoder = greedy best-path stand-in (parity unpinned); b1/pred_ctc_phn_seq is an INPUT"),
         arrays)


def case_ops(name="ops_small"):
    """Op-level vectors straight from the libraries the reference calls: F.ctc_loss (edge cases),
    HF _compute_mask_indices, HF _get_feat_extract_output_lengths, LowPassFilterLayer."""
    arrays = {}
    g = torch.Generator().manual_seed(11)
    # CTC edge cases: normal, repeated labels, target longer than input (inf -> 0), empty target, T=1
    T, B, V = 30, 6, 12
    logits = torch.randn(T, B, V, generator=g).requires_grad_(True)
    lp = torch.log_softmax(logits, dim=-1)
    tg = torch.full((B, 20), -100, dtype=torch.int32)
    tl = [7, 9, 20, 0, 1, 5]
    il = [30, 25, 18, 12, 1, 30]
    for b in range(B):
        tg[b, :tl[b]] = torch.randint(1, V, (tl[b],), generator=g, dtype=torch.int32)
    tg[1, :9] = torch.tensor([3, 3, 3, 4, 4, 5, 5, 5, 5], dtype=torch.int32)
    lp_r = lp
    for red in ("mean", "sum", "none"):
        for zi in (True, False):
            loss = torch.nn.functional.ctc_loss(lp_r, tg, torch.tensor(il), torch.tensor(tl), blank=0,
                                                reduction=red, zero_infinity=zi)
            arrays[f"ctc/loss_{red}_zi{int(zi)}"] = loss.detach().numpy()
    loss = torch.nn.functional.ctc_loss(lp_r, tg, torch.tensor(il), torch.tensor(tl), blank=0,
                                        reduction="mean", zero_infinity=True)
    loss.backward()
    arrays.update({"ctc/logits": logits.detach().numpy(), "ctc/targets": tg.numpy(),
                   "ctc/input_lengths": np.array(il), "ctc/target_lengths": np.array(tl),
                   "ctc/grad_logits_mean_zi1": logits.grad.numpy()})
    # SpecAugment sampler
    for i, (shape, lens, seed) in enumerate([((4, 499), [499, 450, 499, 420], 1), ((2, 199), [199, 150], 2),
                                             ((3, 49), [49, 30, 12], 3), ((2, 1499), None, 4)]):
        np.random.seed(seed)
        am = None
        if lens is not None:
            am = (torch.arange(shape[1])[None] < torch.tensor(lens)[:, None])
        m = hf_w2v2._compute_mask_indices(shape, 0.05, 10, attention_mask=am, min_masks=2)
        arrays[f"mask/{i}/out"] = m
        arrays[f"mask/{i}/shape"] = np.array(shape)
        arrays[f"mask/{i}/lens"] = np.array(lens if lens is not None else [-1])
        arrays[f"mask/{i}/seed"] = np.int64(seed)
    # frame-count arithmetic
    cfg = Wav2Vec2Config()
    m = Wav2Vec2Model.__new__(Wav2Vec2Model)
    m.config = cfg
    n = torch.tensor([400, 401, 16000, 64000, 160000, 159999, 480000, 12345, 799, 800])
    arrays["lens/in"] = n.numpy()
    arrays["lens/out"] = hf_w2v2.Wav2Vec2PreTrainedModel._get_feat_extract_output_lengths(m, n).numpy()
    # low-pass layer
    lpf = ref_modules.LowPassFilterLayer("cpu", 10, 49, 9)
    y = torch.randn(3, 70, 9, generator=g)
    arrays["lowpass/taps"] = lpf.lowpass.weight.detach().numpy()
    arrays["lowpass/in"] = y.numpy()
    arrays["lowpass/out"] = lpf(y).detach().numpy()
    y2 = torch.randn(1, 20, 9, generator=g)          # shorter than the 51 taps
    arrays["lowpass/in_short"] = y2.numpy()
    arrays["lowpass/out_short"] = lpf(y2).detach().numpy()
    save(name, dict(case=name), arrays)


def case_metrics(name="metrics_small"):
    """utility.py metrics as validate()/test() call them (train/train_aptai.py:577-609): per-track RMSE / Pearson,
    boundary P/R/F1/R-value, frame overlap, run-length helpers.  compute_PER needs the absent editdistance package."""
    import utility as ref_util
    arrays = {}
    rng = np.random.RandomState(5)
    gt = rng.randn(700, 9)
    pred = gt * 0.8 + 0.3 * rng.randn(700, 9) + 0.05
    arrays["tv/gt"], arrays["tv/pred"] = gt, pred
    rm = ref_util.tvs_metric_rmse(gt, pred)
    pc = ref_util.tvs_metric_ppc(gt, pred)
    arrays["tv/rmse"] = np.array([rm[k] for k in TV])
    arrays["tv/pcc_r"] = np.array([pc[k][0] for k in TV])
    arrays["tv/pcc_p"] = np.array([pc[k][1] for k in TV])
    for i, (ny, nh) in enumerate([(40, 37), (5, 9), (1, 1)]):
        y = np.sort(rng.uniform(0, 10, ny))
        yh = np.sort(np.concatenate([y[:min(ny, nh) // 2] + rng.uniform(-0.03, 0.03, min(ny, nh) // 2),
                                     rng.uniform(0, 10, nh - min(ny, nh) // 2)]))
        arrays[f"seg/{i}/y"], arrays[f"seg/{i}/yhat"] = y, yh
        arrays[f"seg/{i}/prf"] = np.array([float(v) for v in ref_util.get_stats(y, yh, tolerance=0.02)])
    arrays["seg/metrics_in"] = np.array([13.0, 11.0, 20.0, 17.0])
    arrays["seg/metrics_out"] = np.array([float(v) for v in ref_util.get_metrics(13, 11, 20, 17)])
    frames = [rng.randint(1, 6, n).repeat(rng.randint(1, 5, n)) for n in (30, 12, 1)]
    preds = [np.where(rng.rand(len(f)) < 0.8, f, rng.randint(1, 6, len(f))) for f in frames]
    arrays["ovl/value"] = np.float64(ref_util.evaluate_overlap([f.tolist() for f in frames], [p.tolist() for p in preds]))
    for i, (f, p) in enumerate(zip(frames, preds)):
        arrays[f"ovl/{i}/gt"], arrays[f"ovl/{i}/pred"] = f, p
        arrays[f"rle/{i}/phn"] = np.array(ref_util.phn_frame_id2phn(f.tolist()))
        d = ref_util.phn_frames2dur(f.tolist())
        arrays[f"rle/{i}/dur"] = np.array([[a, b, c] for a, b, c in d], dtype=np.float64)
    save(name, dict(case=name), arrays)


def case_dataprep(name="dataprep_small"):
    """Target preparation the corpus scripts apply before the batch reaches the model:
    `interpolate_signal` (data/dataset_hprc.py:2307-2313: 100 Hz trajectories -> 49 Hz frames) and
    `match_phonemes_to_frames` (utility.py:317-342: phoneme boundaries -> 20 ms frame labels)."""
    import importlib
    for missing in ("soundfile", "textgrid", "praatio", "tgt", "seaborn"):
        if missing not in sys.modules:
            try:
                importlib.import_module(missing)
            except Exception:
                sys.modules[missing] = types.ModuleType(missing)
    sys.path.insert(0, os.path.join(REF, "data"))
    import dataset_hprc as ref_data
    import utility as ref_util
    arrays = {}
    rng = np.random.RandomState(9)
    for i, (n, c, tar) in enumerate([(1000, 9, 490), (337, 1, 165), (50, 3, 50), (2, 2, 7)]):
        sig = rng.randn(n, c) if c > 1 else rng.randn(n)
        arrays[f"interp/{i}/in"] = sig
        arrays[f"interp/{i}/tar_len"] = np.int64(tar)
        arrays[f"interp/{i}/out"] = np.asarray(ref_data.interpolate_signal(sig, tar))
    for i, nph in enumerate([12, 3, 1]):
        bounds = np.round(np.cumsum(rng.uniform(0.03, 0.4, nph)), 2)
        labels = rng.randint(1, 40, nph)
        got = ref_util.match_phonemes_to_frames(list(bounds), list(labels), frame_duration=0.02)
        arrays[f"match/{i}/bounds"] = bounds
        arrays[f"match/{i}/labels"] = labels
        arrays[f"match/{i}/out"] = np.array([-1 if v is None else int(v) for v in got], dtype=np.int64)
    save(name, dict(case=name), arrays)


def case_pr_embgrad(name="pr_embgrad_2x1s", layers=3, seconds=1.0, seed=0):
    """Wav2Vec2_PR.get_embeddings_grad (models/w2v2_pr.py:91-122) of the unmodified reference: the seven returned tensors in
    eval mode (regularisers off) and the gradient norms of sum(phoneme_logits_inter**2) + sum(phoneme_logits_last**2)."""
    cfg_kw = dict(BASE, num_hidden_layers=layers, vocab_size=40, ctc_loss_reduction="mean", ctc_zero_infinity=True, blank=0, **NOREG)
    cfg = hf_config(cfg_kw)
    S = int(16000 * seconds)
    batch = synth.synth_pr_batch(cfg, 2, S, seed=77, lo=8, hi=12)
    vocab = {f"p{i}": i for i in range(40)}
    with tempfile.TemporaryDirectory() as tmp:
        model = ref_w2v2_pr.Wav2Vec2_PR(cfg, None, local_model_dir(cfg, tmp), vocab)
    model.load_state_dict(synth.make_state_dict(synth.pr_param_shapes(cfg), seed))
    model.eval()
    out = model.get_embeddings_grad(batch["input_values"], batch["input_lengths"], vocab, 1, 2)
    (out["phoneme_logits_inter"].pow(2).sum() + out["phoneme_logits_last"].pow(2).sum()).backward()
    arrays = {"out/" + k: v.detach().numpy() for k, v in out.items() if k.startswith("phoneme_logits")}
    for k in ("last_transf_hidden", "intermediate_hidden", "latter_hidden"):        # (batch, feat, time): every 4th feature
        arrays["out/" + k + "_sub"] = out[k].detach().numpy()[:, ::4]
    arrays["out/features_hidden_sub"] = out["features_hidden"].detach().numpy()[:, ::8]
    picks = {"pr_head.weight", "wav2vec2.encoder.layers.0.attention.q_proj.weight", "wav2vec2.encoder.layers.2.feed_forward.output_dense.weight"}
    arrays.update(grads_summary(model.named_parameters(), picks))
    for k, v in batch.items():
        arrays["in/" + k] = v.numpy()
    save(name, dict(case=name, cfg=cfg_kw, seed=seed, S=S, batch_seed=77, intermediate_hidden=1, latter_hidden=2,
                    model="Wav2Vec2_PR.get_embeddings_grad"), arrays)


CASES = {"ops": case_ops, "metrics": case_metrics, "dataprep": case_dataprep, "pr_mini": case_pr_base_mini, "pr_base": case_pr_base, "aptai_large": case_aptai_large,
         "force": case_force, "pr_embgrad": case_pr_embgrad}

if __name__ == "__main__":
    torch.set_num_threads(8)
    for c in (sys.argv[1:] or list(CASES)):
        CASES[c]()
