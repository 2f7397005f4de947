from __future__ import absolute_import

import os.path as osp

from torch.utils.data import Dataset as _TorchDataset


class Preprocessor(_TorchDataset):
    """(fname, pid, x, y) records -> (image tensor, fname, pid, x, y)."""

    def __init__(self, dataset, root=None, transform=None):
        super(Preprocessor, self).__init__()
        self.dataset = dataset
        self.root = root
        self.transform = transform

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, indices):
        if isinstance(indices, (tuple, list)):
            return [self._load(i) for i in indices]
        return self._load(indices)

    def _load(self, index):
        from PIL import Image
        fname, pid, x, y = self.dataset[index]
        path = fname if self.root is None else osp.join(self.root, fname)
        img = Image.open(path).convert('RGB')
        if self.transform is not None:
            img = self.transform(img)
        return img, fname, pid, x, y
