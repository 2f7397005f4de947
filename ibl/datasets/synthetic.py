"""A self-contained stand-in dataset: seeded synthetic images written as PNG files under `root`,
with UTM positions such that every query has gallery positives.  Lets examples/test.py-style
evaluation run end to end without Pittsburgh / Tokyo."""
from __future__ import print_function, absolute_import

import os
import os.path as osp

import numpy as np

from ..utils.data.dataset import Dataset, get_groundtruth


class Synthetic(Dataset):
    registry_name = 'synthetic'

    def __init__(self, root, scale=None, verbose=True, num_query=8, num_gallery=24, height=96,
                 width=128, seed=5):
        super(Synthetic, self).__init__(root)
        from PIL import Image
        rng = np.random.default_rng(seed)
        os.makedirs(self.images_dir, exist_ok=True)

        def make(n, tag, base_pid):
            recs = []
            for i in range(n):
                fname = '{}_{:04d}.png'.format(tag, i)
                path = osp.join(self.images_dir, fname)
                if not osp.isfile(path):
                    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
                    img = np.stack([127 + 120 * np.sin(rng.uniform(0.02, 0.3) * xx +
                                                       rng.uniform(0.02, 0.3) * yy + rng.uniform(0, 6))
                                    for _ in range(3)], axis=-1)
                    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(path)
                # places are 40 m apart along x; queries sit 5 m from gallery place i % G
                recs.append((fname, base_pid + i))
            return recs

        g = make(num_gallery, 'db', 0)
        q = make(num_query, 'q', 100000)
        self.db_test = [(f, pid, 40.0 * i, 0.0) for i, (f, pid) in enumerate(g)]
        self.q_test = [(f, pid, 40.0 * (i % num_gallery) + 5.0, 3.0) for i, (f, pid) in enumerate(q)]
        self.db_val, self.q_val = list(self.db_test), list(self.q_test)
        self.q_train, self.db_train = list(self.q_test), list(self.db_test)   # what test.py:37-38 reads
        self.train = self.q_train + self.db_train
        self.test_pos, sel = get_groundtruth(self.q_test, self.db_test, self.inter_thres)
        self.q_test = [self.q_test[i] for i in sel]
        self.val_pos = list(self.test_pos)
        if verbose:
            print("Synthetic dataset: {} queries, {} gallery images".format(
                len(self.q_test), len(self.db_test)))
