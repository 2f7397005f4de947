from __future__ import absolute_import

import numpy as np
import torch


def to_numpy(x):
    if torch.is_tensor(x):
        return x.detach().cpu().numpy()
    if isinstance(x, np.ndarray):
        return x
    raise ValueError("Cannot convert {} to numpy array".format(type(x)))


def to_torch(x):
    if torch.is_tensor(x):
        return x
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x)
    raise ValueError("Cannot convert {} to torch tensor".format(type(x)))
