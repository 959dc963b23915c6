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


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("n_critic", [1, 2])
def test_train_loop_end_to_end_on_synthetic_celeba(tmp_path, n_critic):
    """The body of the reference's training loop (train.py:89-147: loader 5-tuple -> asign_label -> dis_update -> gen_update every
    n_critic-th iteration -> smooth_moving -> update_learning_rate -> update_attention_status -> sample -> save / resume), driven
    end to end on the MI355X from a synthetic CelebA directory through data_loader.get_loader and the drop-in Solver: real
    loader batches (PIL decode, crop, resize, text synthesis, token padding) reach the HIP kernels, every loss the logger reads is
    finite, the weights move, a snapshot round-trips, and n_critic = 2 (with and without telling the solver) gives the same
    numbers as explicit taping decisions."""
    from hipdwc import host, synth
    from solver import Solver
    from tools import asign_label
    dev = torch.device("cuda:0")
    ap = str(tmp_path / "attr.txt")
    _attr_file(ap)
    g = np.random.RandomState(2)
    probe = CelebA(str(tmp_path), ap, GOLD["selected"], None, "train")
    for name, _ in probe.train_dataset[:12]:
        Image.fromarray(g.randint(0, 256, (218, 178, 3), dtype=np.uint8)).save(str(tmp_path / name), quality=95)
    cfg = synth.make_config(image_size=32, tiny=True, lstm_dropout=0.1)
    loader = data_loader.get_loader(str(tmp_path), 178, 32, 4, ap, GOLD["selected"], "CelebA", "train", num_workers=0)
    loader = torch.utils.data.DataLoader(torch.utils.data.Subset(loader.dataset, list(range(12))), batch_size=4, shuffle=False,
                                         num_workers=0)

    def run(tell):
        torch.manual_seed(1234)
        torch.cuda.manual_seed(1234)
        random.seed(5)
        host.set_noise(host.DeviceNoise())
        trainer = Solver(cfg, dev, None).to(dev)
        trainer.copy_nets()
        if tell == "set":
            trainer.set_n_critic(n_critic)
        w0 = trainer.gen.dec.model[2].conv.weight.detach().clone()
        log = []
        iterations = 0
        for data_iter in loader:
            x_real, label_src, label_trg, txt, lens = data_iter
            c_src = asign_label(label_src, cfg["c_dim"], "CelebA").to(dev)
            c_trg = asign_label(label_trg, cfg["c_dim"], "CelebA").to(dev)
            x_real, label_src, label_trg, txt, lens = (t.to(dev) for t in (x_real, label_src, label_trg, txt, lens))
            gen_now = (iterations + 1) % n_critic == 0
            kw = {"tape_content": gen_now} if tell == "arg" else {}
            trainer.dis_update(x_real, c_src, c_trg, txt, lens, label_src, label_trg, cfg, iterations, **kw)
            if gen_now:
                trainer.gen_update(x_real, c_src, c_trg, txt, lens, label_src, label_trg, cfg, iterations)
            torch.cuda.synchronize()
            trainer.smooth_moving()
            trainer.update_learning_rate()
            trainer.update_attention_status(iterations)
            row = {k: float(torch.as_tensor(getattr(trainer, k)).detach()) for k in dir(trainer)
                   if "loss" in k and not k.startswith("_") and not callable(getattr(trainer, k))}
            assert row and all(np.isfinite(v) for v in row.values()), row
            log.append(row)
            iterations += 1
        assert not torch.equal(w0, trainer.gen.dec.model[2].conv.weight.detach())
        return trainer, log, (x_real, txt, lens), iterations

    trainer, log, (x_real, txt, lens), iterations = run("set")
    assert len(log) == 3 and "loss_gen_total" in log[-1] and "loss_dis_all" in log[0]
    # explicit per-call taping decisions and the detect-and-disable route of an unmodified caller give the same numbers
    for tell in ("arg", "none"):
        _, log2, _, _ = run(tell)
        for a, b in zip(log, log2):
            for k in a:
                assert abs(a[k] - b[k]) <= 1e-5 * max(1.0, abs(a[k])), (tell, k, a[k], b[k])
    outs = trainer.sample(x_real, txt, lens)
    assert len(outs) >= 4 and all(o.shape[0] == x_real.shape[0] and torch.isfinite(o).all() for o in outs)
    ck = tmp_path / "ckpt"
    ck.mkdir()
    trainer.save(str(ck), iterations - 1)
    fresh = Solver(cfg, dev, None).to(dev)
    assert fresh.resume(str(ck), cfg) == iterations
    # (the newest file that matches "gen" is the EMA snapshot gen_<iter>_avg.pt, as in the reference's get_model_list)
    src = trainer.gen_copy if "avg" in Solver._latest(str(ck), "gen") else trainer.gen
    for (ka, va), (kb, vb) in zip(src.state_dict().items(), fresh.gen.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
