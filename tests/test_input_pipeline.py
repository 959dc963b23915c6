"""Input pipeline: vocabulary / token padding against vectors recorded from the reference's
``vocab.ListsToTensor`` on sentences of its own generator, the train/test split against the reference's ``CelebA`` class on a
synthetic attribute file, the PIL transform chain against its definition, and the properties the re-implemented text
grammar must have (every token in the vocabulary, the target value of every changed attribute stated)."""
import json
import os
import random

import numpy as np
import torch
from PIL import Image

import data_loader
import vocab as V
from data_ios import celeba_text as T
from data_ios.celeba_data import CelebA

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "text_pipeline.json")) as f:
    GOLD = json.load(f)


def test_vocab_order_and_token_padding_match_reference():
    v = V.Vocab(dataset="CelebA")
    assert v.itos == GOLD["itos"] and v.size == 102
    assert (v.padding_idx, v.start_idx, v.end_idx, v.unk_idx) == (0, 1, 2, 3)
    toks, lens = V.ListsToTensor([s.split() for s in GOLD["sentences"]], v, mx_len=80)
    assert np.array_equal(toks, np.array(GOLD["tokens"])) and np.array_equal(lens, np.array(GOLD["lens"]))
    assert toks.shape == (len(GOLD["sentences"]), 80)
    row, n = V.getTextLists("make hair black".split(), mx_len=8)
    assert row == ["<bos>", "make", "hair", "black", "<eos>", "<_>", "<_>", "<_>"] and n == 5
    # truncation happens before <bos>/<eos> are added (reference vocab.py:221-224)
    t2, l2 = V.ListsToTensor([["hair"] * 100], v, mx_len=10)
    assert l2[0] == 12 and len(t2[0]) == 12


def _attr_file(path):
    names = GOLD["attr_names"]
    r2 = np.random.RandomState(GOLD["attr_seed"])
    lines = ["2500", " ".join(names)]
    for i in range(2500):
        vals = np.where(r2.rand(40) < 0.4, "1", "-1")
        lines.append("%06d.jpg %s" % (i + 1, " ".join(vals)))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def test_split_matches_reference(tmp_path):
    ap = str(tmp_path / "attr.txt")
    _attr_file(ap)
    ds = CelebA(str(tmp_path), ap, GOLD["selected"], None, "train")
    assert (len(ds.test_dataset), len(ds.train_dataset)) == (GOLD["n_test"], GOLD["n_train"])
    assert ds.test_dataset[:50] == GOLD["test_head"]
    assert ds.train_dataset[:50] == GOLD["train_head"] and ds.train_dataset[-5:] == GOLD["train_tail"]
    assert len(ds) == GOLD["n_train"] and len(CelebA(str(tmp_path), ap, GOLD["selected"], None, "test")) == GOLD["n_test"]
    assert len(ds.all_domains) == 256 and ds.all_domains[5] == [0, 0, 0, 0, 0, 1, 0, 1]


def test_text_grammar_properties():
    v = V.Vocab()
    rng = random.Random(3)
    nr = np.random.RandomState(5)
    longest = 0
    for k in range(600):
        src, trg = (nr.rand(8) < 0.5).astype(int), (nr.rand(8) < 0.5).astype(int)
        if k % 7 == 0:
            trg = src.copy()
        s = T.labels2text(src, trg, rng)
        words = s.split()
        longest = max(longest, len(words))
        assert all(w in v.stoi for w in words), (s, [w for w in words if w not in v.stoi])
        flat = " " + " ".join(words) + " "
        if list(src[:3]) != list(trg[:3]):
            assert all((" " + T.HAIR[i] + " ") in flat for i in range(3) if trg[i]) or (not trg[:3].any() and " unknown " in flat), s
        if src[3] != trg[3]:
            assert any((" " + w + " ") in flat for w in (T.MAN if trg[3] else T.WOMAN)), s
        if src[4] != trg[4]:
            assert any((" " + w + " ") in flat for w in (T.SMILE if trg[4] else T.NOSMILE)), s
        if src[6] != trg[6]:
            assert any((" " + w + " ") in flat for w in T.GLASSES), s
        if src[7] != trg[7]:
            assert any((" " + w + " ") in flat for w in T.BEARD), s
    assert 10 <= longest <= 60                 # far below the 80-token pad width
    same = np.array([1, 0, 0, 1, 0, 1, 0, 1])
    assert isinstance(T.labels2text(same, same.copy(), random.Random(1)), str)


def test_dataset_items_and_loader(tmp_path):
    ap = str(tmp_path / "attr.txt")
    _attr_file(ap)
    g = np.random.RandomState(1)            # stand-ins for the 178x218 aligned CelebA jpegs (only a few are ever opened)
    ds_probe = CelebA(str(tmp_path), ap, GOLD["selected"], None, "train")
    for name, _ in ds_probe.train_dataset[:6]:
        Image.fromarray(g.randint(0, 256, (218, 178, 3), dtype=np.uint8)).save(str(tmp_path / name), quality=95)
    tf = data_loader.ImageTransform(178, 128, flip=False, square=False)
    ds = CelebA(str(tmp_path), ap, GOLD["selected"], tf, "train")
    random.seed(9)
    img, src, trg, txt, ln = ds[0]
    assert img.shape == (3, 128, 128) and img.dtype == torch.float32 and -1.0 <= float(img.min()) and float(img.max()) <= 1.0
    assert src.shape == (8,) and trg.shape == (8,) and set(src.tolist()) <= {0.0, 1.0}
    assert txt.shape == (80,) and txt.dtype == torch.int64 and int(txt[0]) == 1 and int(txt[int(ln) - 1]) == 2
    assert int((txt != 0).sum()) == int(ln) and 3 not in txt.tolist()
    # the transform is crop -> resize -> [0,1] -> [-1,1]; with image_size == crop_size it is the centre crop itself
    raw = np.asarray(Image.open(str(tmp_path / ds.train_dataset[0][0])).convert("RGB"))
    crop = data_loader.ImageTransform(178, 178, flip=False, square=False)(Image.fromarray(raw))
    want = torch.from_numpy(raw[20:198]).permute(2, 0, 1).float() / 255.0 * 2.0 - 1.0
    assert torch.allclose(crop, want, atol=1e-6)
    flipped = data_loader.ImageTransform(178, 178, flip=True, square=False)
    random.seed(0)
    outs = [flipped(Image.fromarray(raw)) for _ in range(8)]
    assert any(torch.equal(o, want) for o in outs) and any(torch.equal(o, want.flip(2)) for o in outs)
    # a loader batch has the 5-tuple layout train.py unpacks (reference train.py:92-100)
    sub = torch.utils.data.Subset(ds, list(range(4)))
    batch = next(iter(torch.utils.data.DataLoader(sub, batch_size=4, shuffle=False, num_workers=0)))
    assert [tuple(t.shape) for t in batch] == [(4, 3, 128, 128), (4, 8), (4, 8), (4, 80), (4,)]
    loader = data_loader.get_loader(str(tmp_path), 178, 128, 2, ap, GOLD["selected"], "CelebA", "train", num_workers=0)
    assert len(loader.dataset) == GOLD["n_train"]
