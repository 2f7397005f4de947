"""Dataset registry with the reference's factory surface (`names()`, `create(name, root, ...)`,
ibl/datasets/__init__.py).

Dataset parsing (Pittsburgh / Tokyo .mat -> records) is outside the accelerated path and the
datasets are not available offline; 'pitts' and 'tokyo' load the json files the reference's own
tooling writes (meta*.json / splits*.json) and raise the reference's RuntimeError otherwise.
'synthetic' is a self-contained stand-in used by the tests and examples."""
from __future__ import absolute_import

from .jsonsets import Pittsburgh, Tokyo
from .synthetic import Synthetic

_REGISTRY = {cls.registry_name: cls for cls in (Pittsburgh, Tokyo, Synthetic)}


def names():
    return sorted(_REGISTRY)


def create(name, root, *args, **kwargs):
    try:
        cls = _REGISTRY[name]
    except KeyError:
        raise KeyError("Unknown dataset:", name)
    return cls(root, *args, **kwargs)


def get_dataset(name, root, *args, **kwargs):
    import warnings
    warnings.warn("get_dataset is deprecated. Use create instead.")
    return create(name, root, *args, **kwargs)
