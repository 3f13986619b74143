"""CPU: host-side logic (integer frame arithmetic, SpecAugment sampler, LR schedule, collate), module surface
(state-dict keys/shapes identical to the reference's), the C-ABI library (loads, exports every declared symbol;
no compute without a GPU) and the no-fallback rule."""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from aptai_amd import hostlogic
from aptai_amd.config import W2V2Config


def test_frame_lengths_match_hf_golden_and_known_values():
    z, _ = load_golden("ops_small")
    cfg = W2V2Config()
    got = hostlogic.feat_extract_output_lengths(torch.from_numpy(z["lens/in"]), cfg.conv_kernel, cfg.conv_stride)
    assert got.dtype == torch.int64 and (got.numpy() == z["lens/out"]).all()
    # SURVEY §2.4 probe values
    for s, t in ((64000, 199), (160000, 499), (480000, 1499)):
        assert hostlogic.feat_extract_output_lengths(s, cfg.conv_kernel, cfg.conv_stride) == t
    assert hostlogic.conv_layer_lengths(160000, cfg.conv_kernel, cfg.conv_stride) == [31999, 15999, 7999, 3999, 1999, 999, 499]
    assert (hostlogic.feat_extract_output_lengths(np.array([160000, 400]), cfg.conv_kernel, cfg.conv_stride) == [499, 1]).all()


def test_frame_mask():
    m = hostlogic.frame_attention_mask(5, torch.tensor([5, 2, 0]))
    assert m.tolist() == [[True] * 5, [True, True, False, False, False], [False] * 5]


def test_lr_schedule_matches_reference_formula():
    f = hostlogic.get_lr_schedule(2, 8, 0.96)          # train/start_train_aptai.sh values
    assert [f(e) for e in range(3)] == [5.0, 10.0, 10.0]
    assert f(9) == 10.0 and abs(f(10) - 10.0) < 1e-12 and abs(f(12) - 10 * 0.96 ** 2) < 1e-12


def test_collate_pads_like_the_reference():
    rng = np.random.RandomState(0)
    items = []
    for n, t in ((1000, 3), (700, 2)):
        items.append({"audio": torch.randn(n), "audio_len": n, "phn_frames_49hz": list(range(1, t + 1)),
                      "phoneme_label": [3, 4][:t - 1], "tvs_norm_49hz": {k: rng.randn(t) for k in hostlogic.TV_NAMES}})
    b = hostlogic.collate_aptai(items, with_phoneme_labels=True)
    assert b["audio_inputs"].shape == (2, 1000) and float(b["audio_inputs"][1, 700:].abs().sum()) == 0.0
    assert b["audio_lengths"].tolist() == [1000, 700] and b["audio_lengths"].dtype == torch.int64
    assert b["phn_frames_49hz"].tolist() == [[1, 2, 3], [1, 2, 0]]
    assert b["LA"].dtype == torch.float64 and b["LA"][1, 2].item() == -100.0
    assert b["phoneme_labels"].dtype == torch.int32 and b["phoneme_labels"].tolist() == [[3, 4], [3, -100]]
    p = hostlogic.collate_pr(items)
    assert set(p) == {"input_values", "input_lengths", "phoneme_labels"}
    assert hostlogic.ctc_target_lengths(p["phoneme_labels"]).tolist() == [2, 1]


def test_lowpass_taps_probe_values():
    h = hostlogic.lowpass_taps(10, 49)
    assert len(h) == 51 and h[0] == 0.0 and abs(h[1] - (-3.127e-5)) < 1e-7 and abs(h[25] - 0.408130) < 1e-6
    assert abs(h.sum() - 1.0) < 1e-12 and np.allclose(h, h[::-1])
    z, _ = load_golden("ops_small")
    assert np.array_equal(h, z["lowpass/taps"].reshape(-1))


def test_best_path_decode():
    logits = np.full((8, 4), -5.0)
    for t, k in enumerate([0, 2, 2, 0, 2, 3, 3, 0]):
        logits[t, k] = 1.0
    assert hostlogic.ctc_best_path(logits, 8).tolist() == [2, 2, 3]
    assert hostlogic.ctc_best_path(logits, 3).tolist() == [2]


@pytest.mark.parametrize("kind", ["base", "large"])
def test_state_dict_keys_match_reference_layout(kind):
    """Keys and shapes equal the HF/reference layout (oracle.synth tables were checked against the reference's own
    load_state_dict when the golden vectors were generated)."""
    from oracle import synth
    from aptai_amd.wav2vec2 import Wav2Vec2Model
    cfg = (W2V2Config.base if kind == "base" else W2V2Config.large)(num_hidden_layers=2)
    m = Wav2Vec2Model(cfg)
    want = synth.w2v2_param_shapes(cfg, "")
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == dict(want)
    m.freeze_feature_encoder()
    assert all(not p.requires_grad for n, p in m.named_parameters() if n.startswith("feature_extractor"))
    assert all(p.requires_grad for n, p in m.named_parameters() if not n.startswith("feature_extractor"))


def test_from_pretrained_roundtrip_and_model_surfaces():
    from oracle import synth
    from aptai_amd.aptai import APTAI
    from aptai_amd.w2v2_pr import Wav2Vec2_PR
    from aptai_amd.wav2vec2 import Wav2Vec2Model
    cfg = W2V2Config.large(num_hidden_layers=1, vocab_size=40)
    with tempfile.TemporaryDirectory() as tmp:
        src = Wav2Vec2Model(cfg)
        src.save_pretrained(tmp)
        assert json.load(open(os.path.join(tmp, "config.json")))["hidden_size"] == 1024
        again = Wav2Vec2Model.from_pretrained(tmp)
        for (k, a), (_, b) in zip(src.state_dict().items(), again.state_dict().items()):
            assert torch.equal(a, b), k
        vocab = {f"p{i}": i for i in range(46)}
        a = APTAI("cpu", vocab, tmp, cfg, None)
        assert {k: tuple(v.shape) for k, v in a.state_dict().items()} == dict(synth.aptai_param_shapes(cfg))
        assert a.state_dict()["tv_lowpass.lowpass.weight"].dtype == torch.float64
        assert set(a.get_config()) == {"device", "vocab", "huggingface_model_id", "pretrain_cfg"}
        assert not any(p.requires_grad for n, p in a.named_parameters() if "feature_extractor" in n)      # default frozen
        pr = Wav2Vec2_PR(cfg, None, tmp, vocab)
        assert {k: tuple(v.shape) for k, v in pr.state_dict().items()} == dict(synth.pr_param_shapes(cfg))
    with pytest.raises(FileNotFoundError):
        Wav2Vec2Model.from_pretrained("facebook/wav2vec2-large-robust")        # hub ids cannot be resolved offline


def test_library_loads_and_exports_every_declared_symbol():
    from aptai_amd import _lib
    L = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert L.aptai_version() >= 100
    undeclared = [n for n in _lib.ARGTYPES if n not in names]
    assert not undeclared, undeclared


def test_product_never_imports_the_oracle_and_has_no_cpu_fallback():
    import aptai_amd
    root = os.path.dirname(aptai_amd.__file__)
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
    from aptai_amd import ops
    from aptai_amd._lib import AptaiHipError
    x = torch.zeros(128, 64, dtype=torch.bfloat16)
    with pytest.raises(AptaiHipError):
        ops.gemm(x, x, 128, 128, 64)
    with pytest.raises(AptaiHipError):
        ops.layernorm_fwd(torch.zeros(4, 256, dtype=torch.bfloat16), torch.ones(256), torch.zeros(256), 1e-5)


def test_target_preparation_matches_reference_vectors(golden):
    """interpolate_signal / match_phonemes_to_frames against vectors generated by importing the reference
    (tests/golden/make_golden.py::case_dataprep)."""
    import numpy as np
    from aptai_amd import hostlogic
    z, _ = golden("dataprep_small")
    for i in range(4):
        got = hostlogic.interpolate_signal(z[f"interp/{i}/in"], int(z[f"interp/{i}/tar_len"]))
        np.testing.assert_allclose(got, z[f"interp/{i}/out"], rtol=1e-12, atol=1e-12)
    for i in range(3):
        got = hostlogic.match_phonemes_to_frames(z[f"match/{i}/bounds"].tolist(), z[f"match/{i}/labels"].tolist(), frame_duration=0.02)
        assert [-1 if v is None else int(v) for v in got] == z[f"match/{i}/out"].tolist()



def test_model_cfg_pkl_loader_is_restricted(tmp_path):
    """model_cfg.pkl (models/force_aptai.py:61-64) goes through a restricted unpickler: plain containers, torch.device and
    configuration OBJECTS (rebuilt as attribute bags) load; any other global is refused before it can run."""
    import pickle
    from aptai_amd.config import W2V2Config, load_model_cfg
    cfg = W2V2Config.base(vocab_size=40)
    p = tmp_path / "a.pkl"
    pickle.dump({"pretrain_cfg": cfg.to_dict(), "cache_dir": None, "huggingface_model_id": "/x", "device": torch.device("cpu")},
                open(p, "wb"))
    back = load_model_cfg(str(p))
    assert W2V2Config.from_any(back["pretrain_cfg"]) == cfg and back["device"] == torch.device("cpu")
    pickle.dump({"pretrain_cfg": cfg}, open(p, "wb"))                     # the build's own config object
    assert W2V2Config.from_any(load_model_cfg(str(p))["pretrain_cfg"]) == cfg
    try:                                                                   # what the reference pickles: an HF config object
        from transformers import Wav2Vec2Config
        hf = Wav2Vec2Config(vocab_size=40, hidden_size=768)
        pickle.dump({"pretrain_cfg": hf, "cache_dir": None, "huggingface_model_id": "/x"}, open(p, "wb"))
        got = W2V2Config.from_any(load_model_cfg(str(p))["pretrain_cfg"])
        assert got.vocab_size == 40 and got.conv_kernel == tuple(hf.conv_kernel) and got.layerdrop == hf.layerdrop
    except ImportError:
        pass

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > /dev/null",))
    pickle.dump({"pretrain_cfg": Evil()}, open(p, "wb"))
    with pytest.raises(pickle.UnpicklingError):
        load_model_cfg(str(p))


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """bench.py --gpus N under a launcher that started a different number of ranks must fail, not measure the wrong job."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_resample_and_one_second_crop():
    """Input pipeline (data/dataset_commonphone.py:17-86): sinc resampling to 16 kHz (torchaudio absent: parity unpinned, properties
    pinned) and the 1-second crop with its phoneme labels."""
    import math
    import random
    sr = 44100
    tt = torch.arange(sr, dtype=torch.float64) / sr
    tone = torch.sin(2 * math.pi * 440.0 * tt).float()
    y = hostlogic.resample(tone, sr, 16000)
    assert y.shape == (16000,) and y.dtype == torch.float32
    want = torch.sin(2 * math.pi * 440.0 * torch.arange(16000, dtype=torch.float64) / 16000).float()
    assert (y[200:-200] - want[200:-200]).abs().max().item() < 2e-3            # a tone keeps frequency, phase and amplitude
    dc = hostlogic.resample(torch.ones(48000), 48000, 16000)
    assert dc.shape == (16000,) and (dc[100:-100] - 1).abs().max().item() < 1e-3
    assert torch.equal(hostlogic.resample(tone, 16000, 16000), tone)
    assert hostlogic.resample(torch.randn(2, 22050), 22050).shape == (2, 16000)
    hi = torch.sin(2 * math.pi * 12000.0 * tt).float()                           # above the new Nyquist: removed
    assert hostlogic.resample(hi, sr, 16000)[200:-200].abs().max().item() < 2e-2
    # crop
    vocab = {"a": 1, "b": 2, "c": 3, "d": 4}
    audio = torch.arange(40000, dtype=torch.float32)
    ts = "[(0.0, 0.5), (0.5, 1.2), (1.2, 1.9), (1.9, 2.5)]"
    random.seed(3)
    item = hostlogic.crop_one_second(audio, ts, "a b c d", vocab)
    s0 = int(item["audio"][0])
    assert item["audio_len"] == 16000 and torch.equal(item["audio"], audio[s0:s0 + 16000])
    t0, t1 = s0 / 16000, s0 / 16000 + 1.0
    spans = hostlogic.convert_ts_float(ts)
    first = [i for i, (a, b) in enumerate(spans) if a <= t0 < b][0]
    last = [i for i, (a, b) in enumerate(spans) if a < t1 <= b][0]
    assert item["phoneme_label"] == [1, 2, 3, 4][first:last + 1]


def test_composed_frame_length_formula_equals_the_step_by_step_one():
    """hostlogic.feat_extract_output_lengths folds the seven floor((n - k) / s) + 1 steps (HF:997-1016) into one floor((n + c) / d)
    for device tensors: identical for every integer, negative intermediate lengths included."""
    from aptai_amd import hostlogic as h
    n = np.arange(-3000, 500000, dtype=np.int64)
    for ck, cs in (((10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)), ((7, 5, 3), (3, 2, 2)), ((4, 4), (4, 3)), ((10,), (5,))):
        c, d = h._composed_length_constants(ck, cs)
        assert np.array_equal(h.feat_extract_output_lengths(n, ck, cs), np.floor_divide(n + c, d)), (ck, cs)
    assert h._composed_length_constants((10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)) == (-80, 320)
    t = torch.tensor([159999, 160000, 400, 399, 0], dtype=torch.int64)
    assert h.feat_extract_output_lengths(t, (10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)).tolist() == [499, 499, 1, 0, -1]


def test_auto_tile_context_sets_and_restores_the_default_tile():
    """ops.auto_tile: what `tile = 0` means to the GEMM descriptor builder inside the context (Force_APTAI's side-stream encoder pass)."""
    from aptai_amd import ops
    assert ops._AUTO_TILE == 0
    with ops.auto_tile(128):
        assert ops._AUTO_TILE == 128
        with ops.auto_tile(0):
            assert ops._AUTO_TILE == 0
        assert ops._AUTO_TILE == 128
    assert ops._AUTO_TILE == 0


def _history(path, blank, sil):
    prev, prev_blank, hist = sil, False, []
    for n in path:
        if n != blank and (n != prev or prev_blank):
            hist.append(n)
        prev_blank = n == blank
        prev = n
    return tuple(hist)


def test_restated_beam_decoder_merges_like_an_exhaustive_search():
    """hostlogic.ctc_beam_search restates flashlight's lexicon-free CTC decoder (the reference's torchaudio ctc_decoder call,
    models/w2v2_pr.py:144-155; the package is absent: PARITY UNPINNED).  With a beam that prunes nothing, its hypotheses must be
    exactly the distinct token histories of all N^T alignments, each scored with the best alignment (log_add=False, the reference's
    setting) or the log-sum of its alignments (log_add=True) - pinned against brute-force enumeration."""
    import itertools
    rng = np.random.default_rng(1)
    for _ in range(12):
        T, N = int(rng.integers(1, 7)), 3
        blank = int(rng.integers(0, 3))
        sil = (blank + 1) % 3
        em = rng.normal(size=(T, N))
        tot, best = {}, {}
        for path in itertools.product(range(N), repeat=T):
            k, sc = _history(path, blank, sil), sum(em[t, p] for t, p in enumerate(path))
            tot[k] = np.logaddexp(tot.get(k, -np.inf), sc)
            best[k] = max(best.get(k, -np.inf), sc)
        for log_add, ref in ((False, best), (True, tot)):
            res = hostlogic.ctc_beam_search(em, blank, sil, beam_size=10 ** 6, beam_threshold=1e9, log_add=log_add, nbest=10 ** 6)
            assert len(res) == len(ref)
            np.testing.assert_allclose(sorted(r[2] for r in res), sorted(ref.values()), rtol=0, atol=1e-12)
            assert all(res[i][2] >= res[i + 1][2] for i in range(len(res) - 1))


def test_reference_decoder_settings_return_the_framed_best_path():
    """With the reference's settings (no LM, max-merge, beam_size 10, beam_threshold 50) the first hypothesis of the restated beam search
    is the frame-wise best path FRAMED by the decoder's opening / closing silence tokens, and its timesteps are row positions of the
    T + 2 long token row (frame t -> t + 1): hostlogic.ctc_bracketed_best_path, the closed form the opt-in
    Wav2Vec2_PR.decoder = "flashlight" uses.  Random emissions incl. silence at the first / last frame (merges with the framing token),
    narrow beams and a zero threshold."""
    rng = np.random.default_rng(0)
    for trial in range(120):
        T, N = int(rng.integers(1, 25)), int(rng.integers(3, 8))
        blank = int(rng.integers(0, N))
        sil = int((blank + 1 + rng.integers(0, N - 1)) % N)
        em = rng.normal(size=(T, N)) * 3
        if trial % 3 == 0:
            em[0, sil] += 10
        if trial % 4 == 0:
            em[-1, sil] += 10
        want_tok, want_ts = hostlogic.ctc_bracketed_best_path(em, T, blank, sil)
        assert want_tok[0] == sil and want_tok[-1] == sil and want_ts[0] == 0
        for beam, thr in ((1, 50.0), (3, 50.0), (10, 50.0), (10, 0.0)):
            tok, ts, score = hostlogic.ctc_beam_search(em, blank, sil, beam_size=beam, beam_threshold=thr)[0]
            assert np.array_equal(tok, want_tok) and np.array_equal(ts, want_ts)
            assert abs(score - em.max(axis=1).sum()) < 1e-9
    # without the framing the closed form is the plain best path and its timesteps are frame numbers
    em = np.array([[0., 5, 0], [0, 5, 0], [9, 0, 0], [0, 0, 7], [0, 0, 7]])
    tok, ts = hostlogic.ctc_bracketed_best_path(em, 5, blank=0, sil=None)
    assert tok.tolist() == [1, 2] and ts.tolist() == [0, 3]
    assert np.array_equal(tok, hostlogic.ctc_best_path(em, 5, blank=0))
    tok, ts = hostlogic.ctc_bracketed_best_path(em, 5, blank=0, sil=1)
    assert tok.tolist() == [1, 2, 1] and ts.tolist() == [0, 4, 6]          # the emitted 1 at frame 0 merged with the opening silence
