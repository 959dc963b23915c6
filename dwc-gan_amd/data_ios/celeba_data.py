"""CelebA dataset feeding ``train.py`` (the reference's ``data_ios.celeba_data.CelebA`` API, reference
celeba_data.py:21-112): attribute-file parsing and the seeded train/test split, a random target label vector per sample,
the text command for the (source, target) pair and its token tensor.  Images are decoded with PIL; the transform is any
callable PIL.Image -> tensor (``data_loader.get_loader`` builds the reference's crop/resize/normalise chain without
torchvision, which the target image does not ship).
"""
import os
import random

import numpy as np
import torch
from PIL import Image
from torch.utils import data

from vocab import ListsToTensor, Vocab

from .celeba_text import labels2text as lab2txt


class CelebA(data.Dataset):
    TEST_ITEMS = 1999        # the first 1999 lines of the shuffled attribute file are the test split

    def __init__(self, image_dir, attr_path, selected_attrs, transform, mode):
        self.image_dir, self.attr_path, self.selected_attrs = image_dir, attr_path, list(selected_attrs)
        self.transform, self.mode = transform, mode
        self.train_dataset, self.test_dataset, self.attr2idx, self.idx2attr = [], [], {}, {}
        self.preprocess()
        self.vocab = Vocab(dataset="CelebA")
        self.all_domains = [[int(c) for c in format(v, "0%db" % len(self.selected_attrs))]
                            for v in range(2 ** len(self.selected_attrs))]
        self.labels2text = lab2txt
        self.num_images = len(self.train_dataset if mode == "train" else self.test_dataset)

    def preprocess(self):
        """list_attr_celeba format: line 0 count, line 1 the 40 attribute names, then '<file> <+-1> x 40'.  The body is
        shuffled with Python's generator seeded 1234 (the split every reference checkpoint was trained on)."""
        with open(self.attr_path, "r") as f:
            lines = [ln.rstrip() for ln in f]
        for i, name in enumerate(lines[1].split()):
            self.attr2idx[name], self.idx2attr[i] = i, name
        body = lines[2:]
        random.seed(1234)
        random.shuffle(body)
        cols = [self.attr2idx[a] for a in self.selected_attrs]
        for i, ln in enumerate(body):
            fields = ln.split()
            item = [fields[0], [int(fields[1 + c] == "1") for c in cols]]
            (self.test_dataset if i < self.TEST_ITEMS else self.train_dataset).append(item)

    def __getitem__(self, index):
        items = self.train_dataset if self.mode == "train" else self.test_dataset
        filename, src_label = items[index]
        _, trg_label = random.choice(items)                       # the target domain: another sample's attribute vector
        words = self.labels2text(np.array(src_label), np.array(trg_label)).split()
        tokens, lens = ListsToTensor([words], self.vocab, mx_len=80)
        image = Image.open(os.path.join(self.image_dir, filename)).convert("RGB")
        if self.transform is not None:
            image = self.transform(image)
        return (image, torch.tensor(src_label).float(), torch.tensor(trg_label).float(),
                torch.from_numpy(tokens).squeeze(0).long(), torch.from_numpy(lens).squeeze(0).long())

    def __len__(self):
        return self.num_images
