"""Checkpoint / json helpers with the reference's names (ibl/utils/serialization.py)."""
from __future__ import print_function, absolute_import

import json
import os.path as osp
import shutil

import torch

from .osutils import mkdir_if_missing


def _rank():
    try:
        import torch.distributed as dist
        return dist.get_rank()
    except Exception:
        return 0


def read_json(fpath):
    with open(fpath, 'r') as f:
        return json.load(f)


def write_json(obj, fpath):
    mkdir_if_missing(osp.dirname(fpath))
    with open(fpath, 'w') as f:
        json.dump(obj, f, indent=4, separators=(',', ': '))


def read_mat(path, key='dbStruct'):
    from scipy.io import loadmat
    return loadmat(path)[key].item()


def save_checkpoint(state, is_best, fpath='checkpoint.pth.tar'):
    mkdir_if_missing(osp.dirname(fpath))
    torch.save(state, fpath)
    if is_best:
        shutil.copy(fpath, osp.join(osp.dirname(fpath), 'model_best.pth.tar'))


def load_checkpoint(fpath):
    if not osp.isfile(fpath):
        raise ValueError("=> No checkpoint found at '{}'".format(fpath))
    checkpoint = torch.load(fpath, map_location=torch.device('cpu'))
    if _rank() == 0:
        print("=> Loaded checkpoint '{}'".format(fpath))
    return checkpoint


def copy_state_dict(state_dict, model, strip=None):
    """Name-matched, size-checked, tolerant copy into `model` (extra / missing keys are reported,
    not fatal); `strip` removes a prefix such as 'module.' from the source names."""
    target = model.state_dict()
    done = set()
    for name, value in state_dict.items():
        if strip is not None and name.startswith(strip):
            name = name[len(strip):]
        dst = target.get(name)
        if dst is None:
            continue
        value = value.data if isinstance(value, torch.nn.Parameter) else value
        if tuple(value.shape) != tuple(dst.shape):
            if _rank() == 0:
                print('mismatch:', name, value.size(), dst.size())
            continue
        dst.copy_(value)
        done.add(name)
    missing = set(target.keys()) - done
    if missing and _rank() == 0:
        print("missing keys in state_dict:", missing)
    return model
