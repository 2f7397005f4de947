"""Pittsburgh / Tokyo 24/7 records from the json files written by the reference's parsers."""
from __future__ import print_function, absolute_import

import os.path as osp

from ..utils.data.dataset import Dataset


class _JsonDataset(Dataset):
    name = 'dataset'

    def __init__(self, root, scale=None, verbose=True):
        super(_JsonDataset, self).__init__(root)
        self.scale = scale
        self.load(verbose, scale)          # Dataset.load: raises RuntimeError("Dataset not found.")


class Pittsburgh(_JsonDataset):
    name = registry_name = 'pitts'

    def __init__(self, root, scale='250k', verbose=True):
        super(Pittsburgh, self).__init__(root, scale=scale, verbose=verbose)


class Tokyo(_JsonDataset):
    name = registry_name = 'tokyo'

    def __init__(self, root, scale=None, verbose=True):
        super(Tokyo, self).__init__(root, scale=None, verbose=verbose)
