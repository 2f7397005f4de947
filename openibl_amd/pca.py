"""Mirror of ibl.pca.PCA (ibl/pca.py): same constructor / train / load / infer surface.

`load` builds the projection  W = (U diag(lams^-1/2))^T  (whitening) or U^T, bias = -W mu
(pca.py:96-106) on the host in float32 and moves it to the GPU; `infer` is one call into the HIP
PCA kernel (oibl_pca_forward): normalize(W v + b) (pca.py:117-121).

Parameter files: the reference writes HDF5 (datasets U, lams, mu, Utmu; pca.py:77-84).  h5py is
imported lazily; a `.npz` holding the same four arrays is accepted as well (and is what `train`
writes when h5py is unavailable).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from . import ops
from .models import default_precision


def _rank() -> int:
    try:
        import torch.distributed as dist
        return dist.get_rank()
    except Exception:
        return 0


def _read_params(path: str):
    if path.endswith(".npz") or (not os.path.exists(path) and os.path.exists(path + ".npz")):
        z = np.load(path if path.endswith(".npz") else path + ".npz")
        return z["U"], z["lams"], z["mu"], z["Utmu"]
    try:
        import h5py
    except ImportError as e:
        raise ImportError(f"PCA.load: reading {path} needs h5py (or provide the same arrays as "
                          f"{path}.npz)") from e
    with h5py.File(path, "r") as f:
        return f["U"][...], f["lams"][...], f["mu"][...], f["Utmu"][...]


def _write_params(path: str, U, lams, mu, Utmu) -> str:
    try:
        import h5py
        if not hasattr(h5py, "File"):
            raise ImportError
        with h5py.File(path, "w") as f:
            for k, v in (("U", U), ("lams", lams), ("mu", mu), ("Utmu", Utmu)):
                f.create_dataset(k, data=v)
        return path
    except ImportError:
        out = path if path.endswith(".npz") else path + ".npz"
        np.savez(out, U=U, lams=lams, mu=mu, Utmu=Utmu)
        return out


class PCA:
    def __init__(self, pca_n_components=4096, pca_whitening=True,
                 pca_parameters_path="./logs/pca_params.h5", precision: Optional[str] = None):
        self.pca_n_components = pca_n_components
        self.pca_whitening = pca_whitening
        self.pca_parameters_path = pca_parameters_path
        self.precision = precision
        self.weight = None
        self.bias = None
        self._w_dev = None

    # ---- offline: parameter estimation (ibl/pca.py:28-84) ------------------------------------
    def train(self, x: torch.Tensor):
        """x [N][dim] float tensor.  Same algorithm as the reference (covariance or its dual,
        eigendecomposition, keep the top components); `torch.symeig` (removed from torch) is
        replaced by torch.linalg.eigh.  Host-side, one-off: not part of the accelerated path."""
        print("calculating PCA parameters...")
        x = x.detach().cpu().float().t()
        n_dims, n_points = x.size(0), x.size(1)
        mu = x.mean(1, keepdim=True)
        x = x - mu
        dual = n_dims > n_points
        x2 = (x.t() @ x if dual else x @ x.t()) / (n_points - 1)
        L, U = torch.linalg.eigh(x2)
        if self.pca_n_components < x2.size(0):
            keep = torch.argsort(L, descending=True)[: self.pca_n_components]
            L, U = L[keep], U[:, keep]
        lams = L.clamp_min(1e-9)
        if dual:
            U = x @ (U @ torch.diag(1.0 / torch.sqrt(lams)) / np.sqrt(n_points - 1))
        Utmu = U.t() @ mu
        saved = _write_params(self.pca_parameters_path, U.numpy(), lams.numpy(), mu.numpy(),
                              Utmu.numpy())
        print("PCA parameters (U {}, lams {}, mu {}) saved to {}".format(
            tuple(U.shape), tuple(lams.shape), tuple(mu.shape), saved))

    # ---- hot path --------------------------------------------------------------------------
    def load(self, gpu=None):
        if _rank() == 0:
            print("load PCA parameters...")
        U, lams, mu, _ = _read_params(self.pca_parameters_path)
        U = np.asarray(U)[:, : self.pca_n_components]
        lams = np.asarray(lams)[: self.pca_n_components]
        mu = np.asarray(mu)
        if self.pca_whitening:
            U = np.matmul(U, np.diag(1.0 / np.sqrt(lams)))
        Utmu = np.matmul(U.T, mu)
        dev = torch.device("cuda", torch.cuda.current_device() if gpu is None else gpu)
        # same tensors and shapes as the reference keeps (pca.py:105-106)
        self.weight = torch.from_numpy(np.ascontiguousarray(U.T)).view(
            self.pca_n_components, -1, 1, 1).float().to(dev)
        self.bias = torch.from_numpy(-Utmu).view(-1).float().to(dev)
        self._w_dev = None

    def _kernel_weight(self) -> torch.Tensor:
        prec = self.precision or default_precision()
        key = (prec, self.weight.data_ptr(), self.weight._version)
        if self._w_dev is None or self._w_dev[0] != key:
            w2 = self.weight.reshape(self.pca_n_components, -1).contiguous()
            self._w_dev = (key, ops.cast(w2, prec))
        return self._w_dev[1]

    def infer(self, data: torch.Tensor) -> torch.Tensor:
        """[N][dim] -> [N][pca_n_components], L2-normalised rows (pca.py:108-123)."""
        if self.weight is None:
            raise RuntimeError("PCA.infer called before PCA.load")
        out = ops.pca(data.float().contiguous(), self._kernel_weight(), self.bias, l2norm=True)
        assert out.size(1) == self.pca_n_components
        return out
