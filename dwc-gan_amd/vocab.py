"""Token vocabulary and padding of the text commands (the reference's ``vocab`` module API: ``Vocab``, ``ListsToTensor``,
``getTextLists``; reference vocab.py:157-240).

The CelebA token table is data, not code: ``data_ios/celeba_vocab.json`` holds the tokens in the index order every
checkpoint's ``enc_txt.embed_tokens.weight`` was trained with (4 specials + 98 words).  Other datasets hand their own token
list to ``Vocab(tokens=...)``.
"""
import json
import os

import numpy as np

PAD, BOS, EOS, UNK = "<_>", "<bos>", "<eos>", "<unk>"
_HERE = os.path.dirname(os.path.abspath(__file__))


def _celeba_tokens():
    with open(os.path.join(_HERE, "data_ios", "celeba_vocab.json")) as f:
        return json.load(f)["itos"]


class Vocab(object):
    def __init__(self, dataset="CelebA", with_SE=True, tokens=None):
        if tokens is None:
            if dataset != "CelebA":
                raise NotImplementedError("only the CelebA token table ships with this package; pass tokens=[...]")
            itos = _celeba_tokens()
            if not with_SE:
                itos = [PAD, UNK] + itos[4:]
        else:
            itos = ([PAD, BOS, EOS, UNK] if with_SE else [PAD, UNK]) + list(tokens)
        self.itos = itos
        self.stoi = {w: i for i, w in enumerate(itos)}
        self.size = len(self.stoi)
        self.padding_idx, self.unk_idx = self.stoi[PAD], self.stoi[UNK]
        self.start_idx, self.end_idx = self.stoi.get(BOS, -1), self.stoi.get(EOS, -1)

    def random_sample(self):
        return self.idx2token(1 + np.random.randint(self.size - 1))

    def idx2token(self, x):
        return [self.idx2token(i) for i in x] if isinstance(x, list) else self.itos[x]

    def token2idx(self, x):
        return [self.token2idx(i) for i in x] if isinstance(x, list) else self.stoi.get(x, self.unk_idx)


def ListsToTensor(xs, vocab, with_S=True, with_E=True, mx_len=50):
    """Word lists -> (int array [batch, mx_len], lengths [batch]): words beyond mx_len are cut BEFORE <bos>/<eos> are added
    and the row is padded to mx_len with <_>, exactly like the reference (so a sentence of >= mx_len-1 words yields a row
    longer than mx_len there too; its generator never produces one)."""
    extra = int(with_S) + int(with_E)
    rows, lens = [], []
    for x in xs:
        x = list(x)[:mx_len]
        n = len(x) + extra
        row = ([vocab.start_idx] if with_S else []) + [vocab.token2idx(w) for w in x] + ([vocab.end_idx] if with_E else [])
        rows.append(row + [vocab.padding_idx] * (mx_len - n))
        lens.append(max(1, n))
    return np.array(rows), np.array(lens)


def getTextLists(x, with_S=True, with_E=True, mx_len=50):
    x = list(x)[:mx_len]
    n = len(x) + int(with_S) + int(with_E)
    return ([BOS] if with_S else []) + x + ([EOS] if with_E else []) + [PAD] * (mx_len - n), n
