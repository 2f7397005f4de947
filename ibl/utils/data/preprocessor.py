"""`Preprocessor(records, root, transform)`: the torch Dataset examples/test.py wraps its record
lists in (test.py:40-54).  Item i is `(image, fname, pid, utm_x, utm_y)`; an index list yields the
list of items (the tuple samplers of the reference ask for several images at once)."""
from __future__ import absolute_import

import os

from torch.utils.data import Dataset as _TorchDataset


def _open_rgb(path):
    from PIL import Image      # imported on first use: the package imports without Pillow
    with Image.open(path) as im:
        return im.convert('RGB')


class Preprocessor(_TorchDataset):
    def __init__(self, dataset, root=None, transform=None):
        super(Preprocessor, self).__init__()
        self.dataset, self.root, self.transform = dataset, root, transform

    def __len__(self):
        return len(self.dataset)

    def _item(self, i):
        record = self.dataset[i]
        fname = record[0]
        image = _open_rgb(fname if self.root is None else os.path.join(self.root, fname))
        if self.transform is not None:
            image = self.transform(image)
        return (image,) + tuple(record)

    def __getitem__(self, indices):
        many = isinstance(indices, (tuple, list))
        return list(map(self._item, indices)) if many else self._item(indices)
