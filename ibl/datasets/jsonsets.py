"""Pittsburgh / Tokyo 24/7 records from the json files written by the reference's parsers."""
from __future__ import print_function, absolute_import

import os.path as osp

from ..utils.data.dataset import Dataset, get_groundtruth
from ..utils.serialization import read_json


def _pluck(identities, utm, indices):
    out = []
    for pid in indices:
        for fname in identities[pid]:
            x, y = utm[pid]
            out.append((fname, pid, x, y))
    return sorted(out)


class _JsonDataset(Dataset):
    name = 'dataset'

    def __init__(self, root, scale=None, verbose=True):
        super(_JsonDataset, self).__init__(root)
        self.scale = scale
        suffix = '' if scale is None else '_' + str(scale)
        meta_f = osp.join(root, 'meta' + suffix + '.json')
        splits_f = osp.join(root, 'splits' + suffix + '.json')
        if not (osp.isfile(meta_f) and osp.isfile(splits_f)):
            raise RuntimeError("Dataset not found.")
        meta, splits = read_json(meta_f), read_json(splits_f)
        ident, utm = meta['identities'], meta['utm']
        # examples/test.py:37-38 reads pitts.q_train / pitts.db_train (the PCA training set);
        # train = q_train + db_train before the queries without positives are dropped
        # (ibl/utils/data/dataset.py:75-88)
        self.q_train = _pluck(ident, utm, sorted(splits.get('q_train', [])))
        self.db_train = _pluck(ident, utm, sorted(splits.get('db_train', [])))
        self.train = self.q_train + self.db_train
        if self.q_train and self.db_train:
            self.train_pos, self.train_neg, sel = get_groundtruth(
                self.q_train, self.db_train, self.intra_thres, self.inter_thres)
            self.train_neg = [self.train_neg[i] for i in sel]
            self.q_train = [self.q_train[i] for i in sel]
        self.q_val = _pluck(ident, utm, sorted(splits.get('q_val', [])))
        self.db_val = _pluck(ident, utm, sorted(splits.get('db_val', [])))
        self.q_test = _pluck(ident, utm, sorted(splits.get('q_test', [])))
        self.db_test = _pluck(ident, utm, sorted(splits.get('db_test', [])))
        if self.q_val and self.db_val:
            self.val_pos, sel = get_groundtruth(self.q_val, self.db_val, self.inter_thres)
            self.q_val = [self.q_val[i] for i in sel]
        if self.q_test and self.db_test:
            self.test_pos, sel = get_groundtruth(self.q_test, self.db_test, self.inter_thres)
            self.q_test = [self.q_test[i] for i in sel]
        if verbose:
            print(self.__class__.__name__, "dataset loaded: {} test queries, {} test gallery".format(
                len(self.q_test), len(self.db_test)))


class Pittsburgh(_JsonDataset):
    name = registry_name = 'pitts'

    def __init__(self, root, scale='250k', verbose=True):
        super(Pittsburgh, self).__init__(root, scale=scale, verbose=verbose)


class Tokyo(_JsonDataset):
    name = registry_name = 'tokyo'

    def __init__(self, root, scale=None, verbose=True):
        super(Tokyo, self).__init__(root, scale=None, verbose=verbose)
