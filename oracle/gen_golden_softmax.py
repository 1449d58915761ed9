"""TEST INFRASTRUCTURE -- generator of tests/golden/softmax_head.npz (round 6): the reference's softmax student.
Runs /root/reference's FastSequenceTagger(use_crf=False) ITSELF (through oracle/ref_import.py, in the build container only):
  * _calculate_loss (sequence_tagger_model.py:2426-2453, 2523-2539) under autograd: remove_x narrowing of the mask, token-level
    cross entropy, / B (sentence_loss) and / mask.sum() (without), d loss / d features;
  * _obtain_labels (:1157-1180, 1212-1246): arg-max tags + softmax confidences at EVERY token, and the get_all_tags distributions.
Fixtures are data only (inputs + the reference's outputs)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import ref_import  # noqa: E402
from oracle.gen_golden import _Sent  # noqa: E402


def build(flair, dict_path, sentence_loss, remove_x):
    from flair.data import Dictionary
    from flair.models import FastSequenceTagger
    td = Dictionary.load_from_file(dict_path)

    class _DummyEmb(torch.nn.Module):
        embedding_length = 8
        name = "dummy"
        embeddings = []

        def embed(self, *a, **k):
            pass

    emb = _DummyEmb()
    emb.embeddings = [emb]
    torch.manual_seed(4321)
    return FastSequenceTagger(hidden_size=8, embeddings=emb, tag_dictionary=td, tag_type="ner", use_crf=False, use_rnn=False,
                              use_cnn=False, dropout=0.0, word_dropout=0.0, locked_dropout=0.0, sentence_loss=sentence_loss,
                              remove_x=remove_x, config=None), td


def main():
    flair = ref_import.load_reference()
    dict_path = os.path.join(ref_import.REFERENCE_ROOT, "resources/taggers/EN-English_x.pkl")
    rng = np.random.default_rng(20220712)
    cases, ci = {}, 0
    for sentence_loss in (True, False):
        for remove_x in (True, False):
            tagger, td = build(flair, dict_path, sentence_loss, remove_x)
            assert not hasattr(tagger, "transitions") or tagger.use_crf is False
            T = len(td)
            x_idx = td.get_idx_for_item("S-X")
            valid = [i for i in range(T) if i not in (x_idx, 0)]
            for (B, n, nreal) in ((2, 12, (4, 7)), (3, 30, (5, 1, 9)), (1, 6, (6,))):
                feats = (rng.standard_normal((B, n, T)) * 1.5).astype(np.float32)
                lengths, tags = [], np.zeros((B, n), np.int64)
                for b in range(B):
                    L = int(rng.integers(max(nreal[b], n // 2), n + 1)) if b > 0 else n
                    lengths.append(L)
                    tags[b, :nreal[b]] = rng.choice(valid, size=nreal[b])
                    tags[b, nreal[b]:L] = x_idx
                lengths = np.asarray(lengths, np.int64)
                ft = torch.from_numpy(feats).clone().requires_grad_(True)
                sents = [_Sent(int(lengths[b]), tags[b]) for b in range(B)]
                mask = (torch.arange(n)[None, :] < torch.from_numpy(lengths)[:, None]).float()
                loss = tagger._calculate_loss(ft, sents, mask)
                loss.backward()
                with torch.no_grad():
                    labels, all_tags = tagger._obtain_labels(torch.from_numpy(feats), sents, get_all_tags=True)
                cases["c%d_sentence_loss" % ci] = np.int64(sentence_loss)
                cases["c%d_remove_x" % ci] = np.int64(remove_x)
                cases["c%d_feats" % ci] = feats
                cases["c%d_lengths" % ci] = lengths
                cases["c%d_tags" % ci] = tags
                cases["c%d_loss" % ci] = loss.detach().numpy()
                cases["c%d_dfeats" % ci] = ft.grad.numpy()
                pt = np.full((B, n), -1, np.int64)
                pc = np.zeros((B, n), np.float32)
                pd = np.zeros((B, n, T), np.float32)
                for b in range(B):
                    for i, lab in enumerate(labels[b]):
                        pt[b, i] = td.get_idx_for_item(lab.value)
                        pc[b, i] = lab.score
                        pd[b, i] = [x.score for x in all_tags[b][i]]
                cases["c%d_pred_tags" % ci] = pt
                cases["c%d_pred_conf" % ci] = pc
                if ci < 2:   # (the distributions of the first two cases only: fixture size)
                    cases["c%d_pred_dist" % ci] = pd
                ci += 1
    cases["n_cases"] = np.int64(ci)
    cases["x_idx"] = np.int64(x_idx)
    np.savez_compressed(os.path.join(GOLD, "softmax_head.npz"), **cases)
    print("softmax_head.npz:", ci, "cases")


if __name__ == "__main__":
    main()
