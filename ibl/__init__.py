"""Drop-in `ibl` package: the reference's module names (ibl.models, ibl.pca, ibl.evaluators,
ibl.utils.*, ibl.datasets) over the MI355X-native implementation in `openibl_amd`, so that the
reference's examples/test.py runs against it unchanged.  Training (ibl.trainers) is out of scope."""
from __future__ import absolute_import

from . import datasets
from . import models
from . import utils
from . import evaluators
from . import pca

__version__ = '0.1.0'
