"""Dataset registry with the reference's factory surface (ibl/datasets/__init__.py).

Dataset parsing (Pittsburgh / Tokyo .mat -> records) is outside the accelerated path and the
datasets are not available offline; 'pitts' and 'tokyo' load the json files the reference's own
tooling writes (meta*.json / splits*.json) and raise the reference's RuntimeError otherwise.
'synthetic' is a self-contained stand-in used by the tests and examples."""
from __future__ import absolute_import

import warnings

from .jsonsets import Pittsburgh, Tokyo
from .synthetic import Synthetic

__factory = {
    'pitts': Pittsburgh,
    'tokyo': Tokyo,
    'synthetic': Synthetic,
}


def names():
    return sorted(__factory.keys())


def create(name, root, *args, **kwargs):
    if name not in __factory:
        raise KeyError("Unknown dataset:", name)
    return __factory[name](root, *args, **kwargs)


def get_dataset(name, root, *args, **kwargs):
    warnings.warn("get_dataset is deprecated. Use create instead.")
    return create(name, root, *args, **kwargs)
