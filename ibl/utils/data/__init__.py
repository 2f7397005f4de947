"""Data-side helpers test.py imports (ibl/utils/data/__init__.py): IterLoader and the test/train
transforms, written without torchvision (absent from this image): PIL resize + to-tensor +
mean/std normalisation with the reference's constants (std = 1/255, i.e. mean-subtracted 0..255
pixels)."""
from __future__ import absolute_import

import numpy as np
import torch

from .dataset import Dataset
from .preprocessor import Preprocessor

MEAN = [0.48501960784313836, 0.4579568627450961, 0.4076039215686255]
STD = [0.00392156862745098, 0.00392156862745098, 0.00392156862745098]


class IterLoader:
    def __init__(self, loader, length=None):
        self.loader = loader
        self.length = length
        self.iter = None

    def __len__(self):
        return self.length if self.length is not None else len(self.loader)

    def new_epoch(self):
        self.iter = iter(self.loader)

    def next(self):
        try:
            return next(self.iter)
        except (StopIteration, TypeError):
            self.iter = iter(self.loader)
            return next(self.iter)


class _TestTransform(object):
    """Resize((h, w)) — or Resize(max(h, w)) on the shorter side for Tokyo queries — then
    ToTensor and Normalize."""

    def __init__(self, height, width, keep_aspect=False, as_uint8=False):
        self.size = (height, width)
        self.keep_aspect = keep_aspect
        self.as_uint8 = as_uint8
        self.mean = torch.tensor(MEAN, dtype=torch.float32).view(3, 1, 1)
        self.std = torch.tensor(STD, dtype=torch.float32).view(3, 1, 1)

    def __call__(self, img):
        from PIL import Image
        if self.keep_aspect:
            s = max(self.size)
            w, h = img.size
            if w <= h:
                nw, nh = s, int(s * h / w)
            else:
                nw, nh = int(s * w / h), s
        else:
            nh, nw = self.size
        img = img.resize((nw, nh), Image.BILINEAR)
        if self.as_uint8:   # raw [H][W][3] bytes: the model normalises them inside its first kernel
            return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        return (x - self.mean) / self.std


def get_transformer_test(height, width, tokyo=False, as_uint8=False):
    """as_uint8=True (extension): skip ToTensor + Normalize and hand the decoded uint8 HWC image to
    the model, which applies exactly that arithmetic on the GPU — same descriptors, 4x less PCIe."""
    return _TestTransform(height, width, keep_aspect=tokyo, as_uint8=as_uint8)


def get_transformer_train(height, width):
    # colour jitter belongs to training, which this package does not implement
    return _TestTransform(height, width)
