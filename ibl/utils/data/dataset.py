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
