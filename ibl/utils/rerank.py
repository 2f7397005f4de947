"""`ibl.utils.rerank.re_ranking` (ibl/utils/rerank.py:32-100 of the reference): k-reciprocal
re-ranking, same name and signature.  The implementation lives in openibl_amd/rerank.py (a
re-derivation pinned to the reference's outputs, tests/golden/rerank_small.npz)."""
from __future__ import absolute_import

from openibl_amd.rerank import re_ranking  # noqa: F401

__all__ = ["re_ranking"]
