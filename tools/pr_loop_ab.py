"""The body of tests/test_gpu_train_loops2.py::test_phoneme_recognizer_loop_replays_graphs_on_the_reference_collate as a tool: the eager and
the graph-replayed train_phoneme_recognizer.train over two epochs from the same initial state; prints both loss traces and their largest
relative deviation.  APTAI_HIP_LIB=tools/ab/r4cause/lib_*.so A/Bs the round-3 kernels (float atomics in the CTC gradient; GroupNorm sums
over the bucket's frames) against the round-4 ones: profiles/r04_pr_loop_cause.txt."""
import os
import random
import sys
import tempfile
from pathlib import Path

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from aptai_amd import hostlogic, train_phoneme_recognizer as T
    from aptai_amd.config import W2V2Config
    from aptai_amd.wav2vec2 import Wav2Vec2Model
    vocab = T.default_vocab()
    w2v = W2V2Config.base(num_hidden_layers=2, layerdrop=0.0, hidden_dropout=0., activation_dropout=0., attention_dropout=0.,
                          feat_proj_dropout=0., apply_spec_augment=False)
    logs = {}
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        torch.manual_seed(0)
        Wav2Vec2Model(w2v).save_pretrained(str(tmp / "w2v"))
        for graphed in (False, True):
            cfg = T.default_cfg(num_epochs=2, batch_size=2, samples_per_epoch=8, learning_rate=2e-5, final_dropout=0.0,
                                huggingface_model_id=str(tmp / "w2v"), pretrain_cfg=w2v, num_warmup_epochs=2, graphed=graphed)
            torch.manual_seed(3)
            model, opt, sched = T.load_model_optimizer(cfg, vocab)
            tr = torch.utils.data.DataLoader(T.SyntheticCommonPhone(8, 1.2, len(vocab), seed=1), batch_size=2, drop_last=True,
                                             collate_fn=hostlogic.collate_pr)
            va = torch.utils.data.DataLoader(T.SyntheticCommonPhone(2, 1.0, len(vocab), seed=2), batch_size=1, collate_fn=hostlogic.collate_pr)
            random.seed(7)
            lines = []
            sub = tmp / ("g" if graphed else "e")
            T.train(cfg, model, opt, sched, vocab, tr, va, sub / "best", sub / "last", sub / "all", log=lines.append)
            logs[graphed] = [float(l.split("train_loss:")[1]) for l in lines if l.startswith("\tepoch")]
    dev = max(abs(a - b) / abs(a) for a, b in zip(logs[False], logs[True]))
    print(f"lib {os.environ.get('APTAI_HIP_LIB', 'product')}")
    print("  eager  ", logs[False])
    print("  graphed", logs[True])
    print(f"  largest relative deviation over the 8 steps: {dev:.3e}")


if __name__ == "__main__":
    main()
