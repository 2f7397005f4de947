"""Dataset base: query / database records `(fname, pid, utm_x, utm_y)` and position ground truth."""
from __future__ import print_function

import numpy as np


def get_groundtruth(query, gallery, intra_thres, inter_thres=None):
    """For each query the gallery positions within `intra_thres` metres (other place id); queries
    without any positive are dropped (`select_pos` lists the kept ones)."""
    from sklearn.neighbors import NearestNeighbors
    utm_q = np.array([[u[2], u[3]] for u in query], dtype=np.float64).reshape(-1, 2)
    utm_g = np.array([[u[2], u[3]] for u in gallery], dtype=np.float64).reshape(-1, 2)
    nn_ = NearestNeighbors(n_jobs=-1).fit(utm_g)
    _, nbrs = nn_.radius_neighbors(utm_q, radius=intra_thres)
    pos, select_pos = [], []
    for qi, cand in enumerate(nbrs):
        keep = [int(i) for i in cand.tolist() if gallery[i][1] != query[qi][1]]
        if keep:
            pos.append(keep)
            select_pos.append(qi)
    if inter_thres is None:
        return pos, select_pos
    _, nbrs = nn_.radius_neighbors(utm_q, radius=inter_thres)
    return pos, [n.tolist() for n in nbrs], select_pos


class Dataset(object):
    def __init__(self, root, intra_thres=10, inter_thres=25):
        self.root = root
        self.intra_thres = intra_thres
        self.inter_thres = inter_thres
        self.train = []
        self.q_val, self.db_val = [], []
        self.q_test, self.db_test = [], []
        self.train_pos, self.train_neg = [], []
        self.val_pos, self.val_neg = [], []
        self.test_pos, self.test_neg = [], []

    @property
    def images_dir(self):
        import os.path as osp
        return osp.join(self.root, 'raw')

    def load(self, verbose, scale=None):
        """Fill the record lists from the json files the reference's dataset parsers write under
        `root` (meta[_scale].json: identities + utm; splits[_scale].json: place ids per split) —
        the loader the reference's Pittsburgh / Tokyo classes call from their constructors
        (ibl/utils/data/dataset.py:57-115).  Training queries without a positive within `intra_thres` are
        dropped; for val / test the reference's literal 25 m radius applies and — as there — every query is
        ASSERTED to have a positive inside it (a split with such a query raises AssertionError, it is not
        filtered)."""
        import os.path as osp
        from ..serialization import read_json
        suffix = '' if scale is None else '_' + str(scale)
        meta_f = osp.join(self.root, 'meta' + suffix + '.json')
        splits_f = osp.join(self.root, 'splits' + suffix + '.json')
        if not (osp.isfile(meta_f) and osp.isfile(splits_f)):
            raise RuntimeError("Dataset not found.")
        meta, splits = read_json(meta_f), read_json(splits_f)
        ident, utm = meta['identities'], meta['utm']

        def pluck(pids):
            return sorted((fname, pid, utm[pid][0], utm[pid][1]) for pid in pids for fname in ident[pid])

        # examples/test.py:37-38 reads q_train / db_train (the PCA training set); train = q_train +
        # db_train BEFORE the queries without positives are dropped
        self.q_train = pluck(sorted(splits.get('q_train', [])))
        self.db_train = pluck(sorted(splits.get('db_train', [])))
        self.train = self.q_train + self.db_train
        if self.q_train and self.db_train:
            self.train_pos, self.train_neg, sel = get_groundtruth(
                self.q_train, self.db_train, self.intra_thres, self.inter_thres)
            self.train_neg = [self.train_neg[i] for i in sel]
            self.q_train = [self.q_train[i] for i in sel]
        self.q_val = pluck(sorted(splits.get('q_val', [])))
        self.db_val = pluck(sorted(splits.get('db_val', [])))
        self.q_test = pluck(sorted(splits.get('q_test', [])))
        self.db_test = pluck(sorted(splits.get('db_test', [])))
        # validation / test positives: the 25 m radius is a literal in the reference, independent of
        # inter_thres, and no query may be left without a positive (ibl/utils/data/dataset.py:89-92)
        if self.q_val and self.db_val:
            self.val_pos, sel = get_groundtruth(self.q_val, self.db_val, 25)
            assert len(sel) == len(self.q_val), "validation queries without a positive within 25 m"
        if self.q_test and self.db_test:
            self.test_pos, sel = get_groundtruth(self.q_test, self.db_test, 25)
            assert len(sel) == len(self.q_test), "test queries without a positive within 25 m"
        if verbose:
            print(self.__class__.__name__, "dataset loaded: {} test queries, {} test gallery".format(
                len(self.q_test), len(self.db_test)))
