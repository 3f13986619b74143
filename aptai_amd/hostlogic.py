"""Host-side (integer / control) logic of the hot path.  No device work happens here.

Every function cites the reference call site it mirrors.  "HF:" = the un-vendored
``transformers/models/wav2vec2/modeling_wav2vec2.py`` (5.15.0 in the survey container).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

TV_NAMES = ("LA", "LP", "JA", "TTCL", "TTCD", "TMCL", "TMCD", "TBCL", "TBCD")   # models/aptai.py:67-70


# ----------------------------------------------------------------------------- frame arithmetic
def conv_out_length(n, kernel: int, stride: int):
    """``floor((n - k) / s) + 1`` — HF:1004-1007 (torch.div(..., rounding_mode='floor') + 1)."""
    if isinstance(n, torch.Tensor):
        return torch.div(n - kernel, stride, rounding_mode="floor") + 1
    if isinstance(n, np.ndarray):
        return np.floor_divide(n - kernel, stride) + 1
    return (int(n) - kernel) // stride + 1


def feat_extract_output_lengths(n, conv_kernel: Sequence[int], conv_stride: Sequence[int]):
    """HF:997-1016 ``_get_feat_extract_output_lengths`` (no adapter). Exact integer arithmetic.

    Device tensors take the composed form: ``floor((floor(a / p) + q) / r) == floor((a + p q) / (p r))`` for integers with
    p, r > 0, so the seven ``floor((n - k) / s) + 1`` steps fold into ONE ``floor((n + c) / d)`` (c, d from the loop below:
    c = -80, d = 320 for wav2vec2) - two tiny launches instead of 21 on the step's dependency chain, bit-identical for every
    integer n (tests/test_cpu_host.py sweeps it against the step-by-step form)."""
    if isinstance(n, torch.Tensor) and n.is_cuda and not n.is_floating_point():
        c, d = 0, 1
        for k, s in zip(conv_kernel, conv_stride):
            c += d * (s - k)
            d *= s
        return torch.div(n + c, d, rounding_mode="floor")
    for k, s in zip(conv_kernel, conv_stride):
        n = conv_out_length(n, k, s)
    return n


def _composed_length_constants(conv_kernel: Sequence[int], conv_stride: Sequence[int]):
    c, d = 0, 1
    for k, s in zip(conv_kernel, conv_stride):
        c += d * (s - k)
        d *= s
    return c, d


def conv_layer_lengths(n_samples: int, conv_kernel: Sequence[int], conv_stride: Sequence[int]) -> List[int]:
    """Frame count after every conv layer (31999, 15999, ... 499 for 160000 samples)."""
    out = []
    n = int(n_samples)
    for k, s in zip(conv_kernel, conv_stride):
        n = (n - k) // s + 1
        out.append(n)
    return out


def frame_attention_mask(n_frames: int, frame_lengths: torch.Tensor) -> torch.Tensor:
    """(B, T) bool mask of valid frames — same result as HF:1018-1036's flip/cumsum construction."""
    ar = torch.arange(n_frames, device=frame_lengths.device)
    return ar[None, :] < frame_lengths[:, None]


# ----------------------------------------------------------------------------- SpecAugment
def compute_mask_indices(shape, mask_prob: float, mask_length: int,
                         attention_mask: Optional[torch.Tensor] = None, min_masks: int = 0,
                         rng=np.random) -> np.ndarray:
    """SpecAugment span sampler, a restatement of HF:101-217 ``_compute_mask_indices``.

    Consumes the numpy RNG in the same order as the reference (one ``rand(1)`` for the
    probabilistic-rounding epsilon, then one ``choice`` per batch row) so that with the same
    ``np.random.seed`` both produce the same mask.  Returns a (B, L) bool array.
    """
    batch_size, sequence_length = shape
    if mask_length < 1:
        raise ValueError("`mask_length` has to be bigger than 0.")
    if mask_length > sequence_length:
        raise ValueError(
            f"`mask_length` has to be smaller than `sequence_length`, but got `mask_length`: {mask_length}"
            f" and `sequence_length`: {sequence_length}`")

    epsilon = rng.rand(1).item()

    def num_spans(input_length):
        n = int(mask_prob * input_length / mask_length + epsilon)
        n = max(n, min_masks)
        if n * mask_length > sequence_length:
            n = sequence_length // mask_length
        if input_length - (mask_length - 1) < n:
            n = max(input_length - (mask_length - 1), 0)
        return n

    if attention_mask is not None:
        input_lengths = [int(x) for x in attention_mask.detach().sum(-1).tolist()]
    else:
        input_lengths = [sequence_length] * batch_size

    mask = np.zeros((batch_size, sequence_length), dtype=bool)
    max_spans = num_spans(sequence_length)
    if max_spans == 0:
        return mask

    rows = []
    for input_length in input_lengths:
        n = num_spans(input_length)
        idx = rng.choice(np.arange(input_length - (mask_length - 1)), n, replace=False)
        dummy = sequence_length - 1 if len(idx) == 0 else idx[0]
        idx = np.concatenate([idx, np.ones(max_spans - n, dtype=np.int32) * dummy])
        rows.append(idx)
    starts = np.array(rows)
    starts = np.broadcast_to(starts[:, :, None], (batch_size, max_spans, mask_length))
    starts = starts.reshape(batch_size, max_spans * mask_length)
    offsets = np.arange(mask_length)[None, None, :]
    offsets = np.broadcast_to(offsets, (batch_size, max_spans, mask_length)).reshape(
        batch_size, max_spans * mask_length)
    idxs = starts + offsets
    if idxs.max() > sequence_length - 1:
        idxs[idxs > sequence_length - 1] = sequence_length - 1
    np.put_along_axis(mask, idxs, 1, -1)
    return mask


# ----------------------------------------------------------------------------- optimiser schedule
def get_lr_schedule(warmup_epochs: int, static_epochs: int, lr_decay: float) -> Callable[[int], float]:
    """LambdaLR factor of train/train_aptai.py:372-386 — note the 10x during warm-up/static phases."""
    plateau_end = warmup_epochs + static_epochs

    def lambda_lr(epoch):
        if epoch >= plateau_end:                       # exponential decay from the 10x plateau
            return 10.0 * lr_decay ** (epoch - plateau_end)
        if epoch >= warmup_epochs:                     # plateau
            return 10.0
        return 10.0 * (epoch + 1) / warmup_epochs      # linear warm-up to 10x
    return lambda_lr


# ----------------------------------------------------------------------------- collate (batch dict C0)
def _pad(seqs, value):
    return torch.nn.utils.rnn.pad_sequence(seqs, batch_first=True, padding_value=value)


def collate_aptai(batch: List[dict], with_phoneme_labels: bool = False) -> Dict[str, torch.Tensor]:
    """Batch dict of train/train_aptai.py:268-332 (+ ``phoneme_labels`` of train/train_force_aptai.py:271-275).

    ``audio_inputs`` zero-padded f32, ``audio_lengths`` i64, ``phn_frames_49hz`` i64 padded with 0,
    nine TV tracks f64 padded with -100.0.
    """
    out = {
        "audio_inputs": _pad([torch.as_tensor(x["audio"]) for x in batch], 0.0),
        "audio_lengths": torch.LongTensor([int(x["audio_len"]) for x in batch]),
    }
    if with_phoneme_labels:
        out["phoneme_labels"] = _pad([torch.IntTensor(x["phoneme_label"]) for x in batch], -100)
    out["phn_frames_49hz"] = _pad([torch.LongTensor(x["phn_frames_49hz"]) for x in batch], 0)
    for name in TV_NAMES:
        out[name] = _pad([torch.from_numpy(np.asarray(x["tvs_norm_49hz"][name])) for x in batch], -100.0)
    return out


def collate_pr(batch: List[dict]) -> Dict[str, torch.Tensor]:
    """Batch dict of train/train_phoneme_recognizer.py:224-239 (labels padded with -100)."""
    return {
        "input_values": _pad([torch.as_tensor(x["audio"]) for x in batch], 0.0),
        "input_lengths": torch.LongTensor([int(x["audio_len"]) for x in batch]),
        "phoneme_labels": _pad([torch.IntTensor(x["phoneme_label"]) for x in batch], -100),
    }


def ctc_target_lengths(phoneme_labels: torch.Tensor) -> torch.Tensor:
    """Count of labels >= 0 per row — the double Python loop of models/w2v2_pr.py:62-70, vectorised."""
    return (phoneme_labels >= 0).sum(dim=-1).to(torch.long).cpu()


# ----------------------------------------------------------------------------- low-pass taps (M1)
def lowpass_taps(cutoff: float, sampling_rate: float) -> np.ndarray:
    """51-tap Hann-windowed sinc of models/modules.py:27-44 (float64, unit DC gain)."""
    fc = cutoff / sampling_rate
    if fc > 0.5:
        raise Exception('Cutoff frequency must be at least twice the sampling rate.')
    b = 0.08
    N = int(np.ceil(4 / b))
    if not N % 2:
        N += 1
    n = np.arange(N)
    h = np.sinc(fc * 2 * (n - (N - 1) / 2))
    w = 0.5 * (1 - np.cos(n * 2 * np.pi / (N - 1)))
    h = h * w
    return h / np.sum(h)


# ----------------------------------------------------------------------------- greedy CTC read-out
def ctc_best_path(logits: np.ndarray, length: int, blank: int = 0) -> np.ndarray:
    """Best-path decode: frame argmax -> collapse repeats -> drop blank.

    Stands where the reference calls torchaudio's lexicon-free beam search
    (models/w2v2_pr.py:143-159).  That decoder is not in the container: *parity unpinned*
    (SURVEY.md §8c); with no LM the top beam is the best path up to beam pruning.
    """
    ids = np.asarray(logits)[:length].argmax(axis=-1)
    keep = np.ones(len(ids), dtype=bool)
    keep[1:] = ids[1:] != ids[:-1]
    ids = ids[keep]
    return ids[ids != blank].astype(np.int64)


# ----------------------------------------------------------------------------- the reference's beam decoder, restated
def ctc_bracketed_best_path(logits: np.ndarray, length: int, blank: int = 0, sil: int = None):
    """What the reference's decoder call returns for its settings, in closed form: (tokens int64, timesteps int32).

    models/w2v2_pr.py:144-159 / utility.py:448-471 call torchaudio.models.decoder.ctc_decoder(lexicon=None, lm=None, nbest=1,
    beam_size=10, beam_threshold=50, log_add=False (default), blank_token='(blank)', sil_token='(...)'): flashlight's
    lexicon-free decoder.  With no language model and max-merging of equal hypotheses a hypothesis' score is the sum of the emissions
    along its best alignment, so the prefix of the frame-wise argmax path is the top candidate of every frame, survives every pruning
    (beam_size >= 1, beam_threshold >= 0) and the first returned hypothesis IS the best path - `ctc_beam_search` below restates the
    published algorithm and tests/test_cpu_host.py checks the equality on random emissions.  What differs from a plain best-path read-out
    is the decoder's FRAMING: it starts every hypothesis with the silence token and closes it with another one (decodeBegin /
    decodeEnd), torchaudio then collapses repeats over that T + 2 long token row and drops blanks, and reports as `timesteps` the row
    positions where a token starts - so a frame t appears as t + 1, the first silence at 0, the closing one at T + 1, and a silence
    emitted at the very first / last frame merges with the framing one.  torchaudio is not in the image: *parity unpinned*
    (SURVEY.md section 8c); the sources restated are flashlight-text LexiconFreeDecoder.cpp and torchaudio/models/decoder/_ctc_decoder.py
    as published."""
    ids = np.asarray(logits)[:length].argmax(axis=-1).astype(np.int64)
    row = ids if sil is None else np.concatenate([[sil], ids, [sil]])
    keep = np.ones(len(row), dtype=bool)
    keep[1:] = row[1:] != row[:-1]
    keep &= row != blank
    pos = np.nonzero(keep)[0]
    return row[pos].astype(np.int64), pos.astype(np.int32)


def ctc_beam_search(emissions: np.ndarray, blank: int, sil: int, beam_size: int = 10, beam_size_token: int = None,
                    beam_threshold: float = 50.0, log_add: bool = False, nbest: int = 1, sil_score: float = 0.0):
    """flashlight's lexicon-free CTC beam search without a language model, restated from the published sources (see
    ctc_bracketed_best_path), plus torchaudio's read-out of each hypothesis.  emissions [T][N] (any scores: the reference feeds raw
    logits).  Returns the `nbest` best hypotheses, best first, as (tokens int64, timesteps int32, score float).

    State of a hypothesis = (token history, last token, last-was-blank); per frame every hypothesis is extended by every token (the
    `beam_size_token` best of the frame when given): a non-blank token that differs from the last one, or follows a blank, EXTENDS
    the history; a blank keeps it and sets the flag; a repeat keeps it.  Candidates below best - beam_threshold are dropped,
    candidates of equal state are merged (max, or log-add with `log_add`), the `beam_size` best survive.  Test infrastructure for the
    closed form above and the opt-in `Wav2Vec2_PR.decoder = "flashlight"`: O(T beam N) Python, not a product hot path."""
    em = np.asarray(emissions, dtype=np.float64)
    T, N = em.shape
    k_tok = N if beam_size_token is None else min(int(beam_size_token), N)
    trie = {}                                      # (history id, token) -> history id (the ZeroLM state tree)

    def child(hist, tok):
        key = (hist, tok)
        if key not in trie:
            trie[key] = len(trie) + 1
        return trie[key]

    # a hypothesis: (score, history id, parent hypothesis or None, token, last-was-blank)
    hyps = [(0.0, 0, None, sil, False)]
    for t in range(T):
        order = np.argsort(-em[t], kind="stable")[:k_tok] if k_tok < N else np.arange(N)
        cands, best = [], -np.inf
        for h in hyps:
            hs, hist, _, prev, prev_blank = h
            for n in order:
                n = int(n)
                sc = hs + em[t, n] + (sil_score if n == sil else 0.0)
                if n != blank and (n != prev or prev_blank):
                    c = (sc, child(hist, n), h, n, False)
                elif n == blank:
                    c = (sc, hist, h, n, True)
                else:
                    c = (sc, hist, h, n, False)
                best = max(best, sc)
                if sc >= best - beam_threshold:
                    cands.append(c)
        hyps = _merge_and_prune(cands, best - beam_threshold, beam_size, log_add)
    cands = [(h[0], h[1], h, sil, False) for h in hyps]
    final = _merge_and_prune(cands, max(c[0] for c in cands) - beam_threshold, beam_size, log_add)
    out = []
    for h in final[:nbest]:
        row, node = [], h
        while node is not None:
            row.append(node[3])
            node = node[2]
        row = np.asarray(row[::-1], dtype=np.int64)          # T + 2 tokens: opening silence, one per frame, closing silence
        keep = np.ones(len(row), dtype=bool)
        keep[1:] = row[1:] != row[:-1]
        keep &= row != blank
        pos = np.nonzero(keep)[0]
        out.append((row[pos], pos.astype(np.int32), float(h[0])))
    return out


def _merge_and_prune(cands, floor, beam_size, log_add):
    cands = [c for c in cands if c[0] >= floor]
    cands.sort(key=lambda c: (c[1], c[3], c[4], -c[0]))          # equal states adjacent, the best of each first
    merged = []
    for c in cands:
        if merged and merged[-1][1] == c[1] and merged[-1][3] == c[3] and merged[-1][4] == c[4]:
            if log_add:
                m = merged[-1]
                merged[-1] = (float(np.logaddexp(m[0], c[0])),) + m[1:]
            continue
        merged.append(c)
    merged.sort(key=lambda c: -c[0])
    return merged[:beam_size]


# ----------------------------------------------------------------------------- target preparation
def interpolate_signal(org_sig, tar_len: int) -> np.ndarray:
    """Linear resampling of a [frames] or [frames][channels] track to ``tar_len`` points that span the same time range
    (data/dataset_hprc.py:2307-2313: 100 Hz trajectories -> the encoder's 49 Hz frames).  scipy's interp1d(kind='linear') on the
    integer grid, written out: out[i] = (1-w) x[floor(p)] + w x[floor(p)+1] at p = i (n-1)/(tar_len-1)."""
    x = np.asarray(org_sig, dtype=np.float64)
    n = x.shape[0]
    pos = np.linspace(0, n - 1, tar_len)
    lo = np.clip(np.floor(pos).astype(np.int64), 0, max(n - 2, 0))
    w = pos - lo
    hi = np.minimum(lo + 1, n - 1)
    if x.ndim > 1:
        w = w.reshape((-1,) + (1,) * (x.ndim - 1))
    return x[lo] + (x[hi] - x[lo]) * w


def match_phonemes_to_frames(phoneme_boundaries, phoneme_list, frame_duration: float = 0.02) -> list:
    """Frame labels from phoneme END boundaries (utility.py:317-342): frame k covers [k*step, (k+1)*step) centiseconds with
    step = int(frame_duration*100); it takes the first phoneme whose boundary falls inside, otherwise keeps the previous frame's
    label (None before the first hit).  The reference's O(frames x phonemes) scan as one searchsorted."""
    b = np.asarray(phoneme_boundaries, dtype=np.float64)
    step = int(frame_duration * 100)
    stop = int(b[-1] * 100) + 1
    out, current = [], None
    starts = np.arange(0, stop, step)
    for fs in starts:
        lo, hi = fs / 100.0, (fs + int(frame_duration * 100)) / 100.0
        hit = np.nonzero((b >= lo) & (b < hi))[0]
        if hit.size:
            current = phoneme_list[int(hit[0])]
        out.append(current)
    return out



# ----------------------------------------------------------------------------- input pipeline: 16 kHz resample + crop (f-4)
def resample(waveform, orig_freq: int, new_freq: int = 16000, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """`torchaudio.functional.resample(waveform, orig_freq, new_freq)` as data/dataset_commonphone.py:31-33 calls it (wav2vec2
    was pre-trained on 16 kHz audio): band-limited sinc interpolation with a Hann window (torchaudio's default
    `sinc_interp_hann`, lowpass_filter_width 6, rolloff 0.99), restated from the published algorithm - torchaudio is not in this
    image, so parity with it is unpinned; the test pins the properties (length, DC gain, a tone keeps its frequency, identity).
    waveform: 1-D or (channels, time) array / tensor; returns a float32 tensor of the same rank."""
    import math
    w = torch.as_tensor(np.asarray(waveform), dtype=torch.float32) if not torch.is_tensor(waveform) else waveform.float()
    squeeze = w.dim() == 1
    w = w.reshape(1, -1) if squeeze else w.reshape(-1, w.shape[-1])
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq == new_freq:
        return w[0] if squeeze else w
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window * (base / orig)
    length = w.shape[-1]
    x = torch.nn.functional.pad(w, (width, width + orig))
    y = torch.nn.functional.conv1d(x[:, None], kernels.float(), stride=orig)          # (channels, new, frames)
    y = y.transpose(1, 2).reshape(w.shape[0], -1)
    y = y[..., :math.ceil(new * length / orig)]
    return y[0] if squeeze else y


def convert_ts_float(input_string: str):
    """'[(0.0, 0.1), (0.1, 0.25)]' -> [(0.0, 0.1), (0.1, 0.25)] (utility.py:298-309)."""
    s = input_string.replace('[', '').replace(']', '').replace(' ', '')
    out = []
    for piece in s.split('),('):
        a, b = map(float, piece.strip('()').split(','))
        out.append((a, b))
    return out


def phonemes_idx(vocab: dict, phonemes: str) -> List[int]:
    """Space-separated phoneme string -> vocabulary ids (utility.py:236-244)."""
    return [vocab[p] for p in phonemes.split(' ')]


def crop_one_second(audio, phoneme_timestamps: str, phonemes: str, vocab: dict, rng=None, duration_samples: int = 16000):
    """The optional 1-second training crop of data/dataset_commonphone.py:35-70: a random window of `duration_samples` and the
    ids of the phonemes that overlap it (first = the phoneme that contains the window start, last = the one that contains its
    end).  `rng` needs `randint(a, b)` with both ends included (default: the `random` module, as the reference)."""
    import random as _random
    rng = rng or _random
    audio = torch.as_tensor(audio)
    start = rng.randint(0, len(audio) - duration_samples)
    end = start + duration_samples
    crop = audio[start:end]
    t0, t1 = start / 16000, end / 16000
    ts = convert_ts_float(phoneme_timestamps)
    hit = []
    for i, (a, b) in enumerate(ts):
        if a <= t0 < b:
            hit.append(i)
        if a < t1 <= b:
            hit.append(i)
    assert len(hit) == 2
    names = phonemes.split(' ')
    label = phonemes_idx(vocab, ' '.join(names[i] for i in range(hit[0], hit[1] + 1)))
    return {"audio": crop, "audio_len": len(crop), "phoneme_label": label}
