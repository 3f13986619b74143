"""Round-4 diagnosis of the red round-3 test (tests/test_gpu_train_loops2.py, graph-replayed vs eager Wav2Vec2_PR loop): per-parameter
gradient agreement FROM IDENTICAL PARAMETERS, which a loss-after-8-Adam-steps bound cannot give.

For each of a few reference-collated batches (train/train_phoneme_recognizer.py:224-239: padded to the batch's own longest utterance):
    eager            model(**batch); loss.backward()                      (twice: is it bit-reproducible?)
    graph, own S     BucketedGraphedStep with the batch's own length as the only bucket (no padding, same shapes)
    graph, bucket    BucketedGraphedStep with the default buckets (1.2 s -> 2 s: 60 % more frames, other tiles / split-K)
and prints per parameter family the relative L2 distance of the gradients to the eager ones, plus what one Adam step at the test's
learning rate does to those distances (the amplification the loss trajectories showed).

    python tools/pr_graph_vs_eager.py [--layers 2] [--seconds 1.2]
"""
import argparse
import collections
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def family(name: str) -> str:
    if "feature_extractor.conv_layers.0" in name:
        return "conv0"
    if "feature_extractor" in name:
        return "conv1-6"
    if "feature_projection" in name:
        return "projection"
    if "pos_conv" in name:
        return "posconv"
    if "encoder.layers" in name:
        return "layer." + name.split("encoder.layers.")[1].split(".", 1)[1].rsplit(".", 1)[0]
    if "pr_head" in name:
        return "pr_head"
    return name.rsplit(".", 1)[0]


def grads_of(model):
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def compare(ref, got, title):
    fam = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
    for n, g in ref.items():
        if n not in got:
            print(f"   !! {n} has no gradient in '{title}'")
            continue
        d = (got[n].double() - g.double())
        f = fam[family(n)]
        f[0] += float((d * d).sum()); f[1] += float((g.double() ** 2).sum()); f[2] = max(f[2], float(d.abs().max()))
    worst = 0.0
    print(f"  {title}")
    for k in sorted(fam):
        num, den, mx = fam[k]
        rel = (num / max(den, 1e-300)) ** 0.5
        worst = max(worst, rel)
        print(f"     {k:34s} rel-L2 {rel:9.3e}   max|d| {mx:9.3e}   |g| {den ** 0.5:9.3e}")
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=1.2)
    ap.add_argument("--batches", type=int, default=3)
    a = ap.parse_args()
    from aptai_amd import hostlogic, train_phoneme_recognizer as T
    from aptai_amd.config import W2V2Config
    from aptai_amd.graphed import BucketedGraphedStep
    from aptai_amd.optim import Adam
    from aptai_amd.wav2vec2 import Wav2Vec2Model
    vocab = T.default_vocab()
    w2v = W2V2Config.base(num_hidden_layers=a.layers, layerdrop=0.0, hidden_dropout=0., activation_dropout=0., attention_dropout=0.,
                          feat_proj_dropout=0., apply_spec_augment=False)
    with tempfile.TemporaryDirectory() as tmp:
        torch.manual_seed(0)
        Wav2Vec2Model(w2v).save_pretrained(tmp)
        cfg = T.default_cfg(learning_rate=2e-5, final_dropout=0.0, huggingface_model_id=tmp, pretrain_cfg=w2v)
        torch.manual_seed(3)
        model, _, _ = T.load_model_optimizer(cfg, vocab)
    model.train()
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    dl = torch.utils.data.DataLoader(T.SyntheticCommonPhone(2 * a.batches, a.seconds, len(vocab), seed=1), batch_size=2, drop_last=True,
                                     collate_fn=hostlogic.collate_pr)
    for bi, batch in enumerate(dl):
        batch = {k: v.cuda() for k, v in batch.items()}
        S = batch["input_values"].shape[1]
        print(f"== batch {bi}: S = {S}, lengths {batch['input_lengths'].reshape(-1).tolist()}, labels {tuple(batch['phoneme_labels'].shape)}")
        runs = {}
        for rep in range(2):
            model.load_state_dict(init)
            model.zero_grad(set_to_none=True)
            out = model(**batch)
            out["loss"].backward()
            runs[f"eager{rep}"] = (float(out["loss"]), grads_of(model))
        same = all(torch.equal(runs["eager0"][1][n], runs["eager1"][1][n]) for n in runs["eager0"][1])
        print(f"  eager loss {runs['eager0'][0]:.6f}; eager twice bit-identical: {same}")
        for title, buckets in (("graph, own S as the bucket", [S]), ("graph, default buckets", None)):
            model.load_state_dict(init)
            model.wav2vec2._cache_mode = None
            model.wav2vec2._cache.clear()
            opt = Adam(model.parameters(), lr=0.0).publish_to(model)
            with BucketedGraphedStep(model, opt, bucket_samples=buckets) as runner:
                losses, gs = [], []
                for rep in range(2):
                    o = runner.step(batch)
                    losses.append(float(o["loss"]))
                    gs.append(grads_of(model))
                same = all(torch.equal(gs[0][n], gs[1][n]) for n in gs[0])
                print(f"  {title}: loss {losses[0]:.6f} (eager {runs['eager0'][0]:.6f}); replay twice bit-identical: {same}")
                runs[title] = (losses[0], gs[0])
                compare(runs["eager0"][1], gs[0], f"{title} vs eager")
        # what Adam's first step makes of it: update = lr * sign(g) wherever |g| >> eps, so parameters whose gradient sign differs
        # move apart by 2 lr; report the share of elements whose sign differs and the rel-L2 of the two updates
        ge, gb = runs["eager0"][1], runs["graph, default buckets"][1]
        flips = tot = 0
        for n in ge:
            flips += int(((ge[n] > 0) != (gb[n] > 0)).sum()); tot += ge[n].numel()
        print(f"  gradient sign differs on {flips} of {tot} elements ({100.0 * flips / tot:.3f} %): Adam's first update differs by 2 lr there")
    print("done")


if __name__ == "__main__":
    main()
