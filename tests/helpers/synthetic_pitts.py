"""A tiny Pittsburgh-FORMAT dataset (the json files the reference's own parsers write —
ibl/datasets/pitts.py: meta_<scale>.json with `identities` / `utm`, splits_<scale>.json with the six
pid lists — plus PNG images under raw/), so that `datasets.create('pitts', root, scale='30k')` and
everything examples/test.py builds on it can run without the real data."""
import json
import os
import os.path as osp

import numpy as np


def make(root, n_db=14, n_q=6, height=72, width=96, seed=3, scale="30k"):
    """Places 40 m apart along x; query i stands 6 m from database place (2 i) % n_db, so every
    query has exactly one positive inside the 25 m radius.  The train / val / test splits use the
    same geometry with disjoint pids.  Returns the dataset root."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    raw = osp.join(root, "raw")
    os.makedirs(raw, exist_ok=True)
    identities, utm = [], []
    splits = {}

    def image(fname, base=None):
        path = osp.join(raw, fname)
        yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
        if base is None:
            fx, fy, ph = rng.uniform(0.03, 0.4, 3), rng.uniform(0.03, 0.4, 3), rng.uniform(0, 6, 3)
            img = np.stack([127 + 110 * np.sin(fx[c] * xx + fy[c] * yy + ph[c]) for c in range(3)], -1)
        else:
            img = base + rng.normal(0, 6.0, base.shape)
        img = np.clip(img, 0, 255)
        Image.fromarray(img.astype(np.uint8)).save(path)
        return img

    for split in ("train", "val", "test"):
        db_pids, q_pids, db_imgs = [], [], []
        for i in range(n_db):
            pid = len(identities)
            fname = "{}_db_{:03d}.png".format(split, i)
            db_imgs.append(image(fname))
            identities.append([fname])
            utm.append([1000.0 * len(splits) + 40.0 * i, 0.0])
            db_pids.append(pid)
        for i in range(n_q):
            pid = len(identities)
            j = (2 * i) % n_db
            fname = "{}_q_{:03d}.png".format(split, i)
            image(fname, base=db_imgs[j])           # a noisy view of its positive
            identities.append([fname])
            utm.append([utm[db_pids[j]][0] + 6.0, 3.0])
            q_pids.append(pid)
        splits["q_" + split] = q_pids
        splits["db_" + split] = db_pids
    with open(osp.join(root, "meta_{}.json".format(scale)), "w") as f:
        json.dump({"name": "pitts_" + scale, "identities": identities, "utm": utm}, f)
    with open(osp.join(root, "splits_{}.json".format(scale)), "w") as f:
        json.dump(splits, f)
    return root
