"""``get_loader`` with the reference's signature (reference data_loader.py:6-31) on PIL + torch only.

The reference composes torchvision transforms; torchvision is not part of the target image, so the same chain is
spelled out: RandomHorizontalFlip (train) -> CenterCrop(crop_size) -> Resize(image_size) (shorter side, bilinear with
PIL's antialiasing, what ``T.Resize`` does on PIL images) -> ToTensor ([0,1] CHW fp32) -> Normalize(0.5, 0.5) = [-1, 1].
Batches come out as the 5-tuple ``train.py`` unpacks (reference train.py:92-100); the trainer packs images into its
internal NHWC form on the device (``hipdwc.ops.pack_image``), so the host side stays plain NCHW fp32.
"""
import random

import numpy as np
import torch
from PIL import Image
from torch.utils import data


class ImageTransform:
    def __init__(self, crop_size, image_size, flip, square):
        self.crop, self.size, self.flip, self.square = crop_size, image_size, flip, square

    def __call__(self, img):
        if self.flip and random.random() < 0.5:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        if self.square:                                           # non-CelebA branch: Resize((S, S))
            img = img.resize((self.size, self.size), Image.BILINEAR)
        else:
            w, h = img.size
            left, top = int(round((w - self.crop) / 2.0)), int(round((h - self.crop) / 2.0))
            img = img.crop((left, top, left + self.crop, top + self.crop))
            w, h = img.size
            if w <= h:
                img = img.resize((self.size, max(1, int(self.size * h / w))), Image.BILINEAR)
            else:
                img = img.resize((max(1, int(self.size * w / h)), self.size), Image.BILINEAR)
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
        return x.sub_(0.5).div_(0.5)


def _seed_worker(worker_id):
    random.seed(torch.initial_seed() % (2 ** 32))


def get_loader(image_dir, crop_size=178, image_size=128, batch_size=16, attr_path=None, selected_attrs=None,
               dataset="CelebA", mode="train", num_workers=4):
    """Build and return a data loader (same arguments and batch layout as the reference's)."""
    transform = ImageTransform(crop_size, image_size, flip=(mode == "train"), square=(dataset != "CelebA"))
    if dataset != "CelebA":
        raise NotImplementedError("only the CelebA dataset class is provided (the shipped configuration)")
    from data_ios.celeba_data import CelebA
    cur = CelebA(image_dir, attr_path, selected_attrs, transform, mode)
    return data.DataLoader(dataset=cur, batch_size=batch_size, shuffle=True, num_workers=num_workers,
                           worker_init_fn=_seed_worker, pin_memory=torch.cuda.is_available())
