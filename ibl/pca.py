"""ibl.pca — PCA / whitening with the reference's interface (ibl/pca.py)."""
from openibl_amd.pca import PCA

__all__ = ['PCA']
