from __future__ import absolute_import

import numpy as np
import torch


def to_numpy(tensor):
    if torch.is_tensor(tensor):
        return tensor.detach().cpu().numpy()
    if isinstance(tensor, np.ndarray):
        return tensor
    raise ValueError("Cannot convert {} to numpy array".format(type(tensor)))


def to_torch(ndarray):
    if torch.is_tensor(ndarray):
        return ndarray
    if isinstance(ndarray, np.ndarray):
        return torch.from_numpy(ndarray)
    raise ValueError("Cannot convert {} to torch tensor".format(type(ndarray)))
