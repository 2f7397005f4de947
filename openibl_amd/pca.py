"""Mirror of ibl.pca.PCA (ibl/pca.py): same constructor / train / load / infer surface.

`load` builds the projection  W = (U diag(lams^-1/2))^T  (whitening) or U^T, bias = -W mu
(pca.py:96-106) on the host in float32 and moves it to the GPU; `infer` is one call into the HIP
PCA kernel (oibl_pca_forward): normalize(W v + b) (pca.py:117-121).

Parameter files: the reference writes HDF5 (datasets U, lams, mu, Utmu; pca.py:77-84).  h5py is
imported lazily; an npz archive holding the same four arrays is accepted as well, recognised by
content, and is what `train` writes — at the requested path — when h5py is unavailable.
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


_HDF5_MAGIC = b"\x89HDF\r\n\x1a\n"


def _read_params(path: str):
    """(U, lams, mu, Utmu) from the reference's HDF5 file or from an npz archive holding the same
    four arrays.  The format is sniffed from the file's first bytes, not from its name: without
    h5py `train` writes the archive AT the path the caller chose (examples/test.py:109-111 decides
    with osp.isfile(<...>.h5) whether to train again)."""
    if not os.path.exists(path) and os.path.exists(path + ".npz"):
        path = path + ".npz"                   # files written by earlier versions of this package
    with open(path, "rb") as f:
        magic = f.read(8)
    if magic != _HDF5_MAGIC:
        z = np.load(path)
        return z["U"], z["lams"], z["mu"], z["Utmu"]
    try:
        import h5py
    except ImportError as e:
        raise ImportError(f"PCA.load: {path} is an HDF5 file and h5py is not installed (an npz "
                          "archive with the arrays U, lams, mu, Utmu at the same path is accepted "
                          "as well)") from e
    with h5py.File(path, "r") as f:
        return f["U"][...], f["lams"][...], f["mu"][...], f["Utmu"][...]


def _write_params(path: str, U, lams, mu, Utmu) -> str:
    folder = os.path.dirname(path)
    if folder:
        os.makedirs(folder, exist_ok=True)
    try:
        import h5py
        if not hasattr(h5py, "File"):
            raise ImportError
        with h5py.File(path, "w") as f:
            for k, v in (("U", U), ("lams", lams), ("mu", mu), ("Utmu", Utmu)):
                f.create_dataset(k, data=v)
    except ImportError:
        with open(path, "wb") as f:            # same four arrays, npz container, SAME path
            np.savez(f, U=U, lams=lams, mu=mu, Utmu=Utmu)
    return path


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
            U.shape[1], -1, 1, 1).float().to(dev)
        self.bias = torch.from_numpy(-Utmu).view(-1).float().to(dev)
        self._w_dev = None

    def _kernel_params(self):
        """(weight in the precision's element type, bias), both padded with zero rows to the next
        multiple of 128 outputs — the projection kernel's tile — which leaves W v + b and its L2
        norm unchanged (the padded outputs are exactly 0 and are sliced off again)."""
        prec = self.precision or default_precision()
        key = (prec, self.weight.data_ptr(), self.weight._version, self.bias.data_ptr())
        if self._w_dev is None or self._w_dev[0] != key:
            n = int(self.weight.shape[0])
            w2 = self.weight.reshape(n, -1).contiguous()
            b = self.bias
            pad = (-n) % 128
            if pad:
                w2 = torch.cat([w2, w2.new_zeros((pad, w2.shape[1]))]).contiguous()
                b = torch.cat([b, b.new_zeros(pad)]).contiguous()
            self._w_dev = (key, ops.PcaWeight(ops.cast(w2, prec)), b)
        return self._w_dev[1], self._w_dev[2]

    def infer(self, data: torch.Tensor) -> torch.Tensor:
        """[N][dim] -> [N][n], L2-normalised rows (pca.py:108-123); n = pca_n_components, or the
        number of components the parameter file holds if that is smaller."""
        if self.weight is None:
            raise RuntimeError("PCA.infer called before PCA.load")
        n = int(self.weight.shape[0])
        w, b = self._kernel_params()
        out = ops.pca(data.float().contiguous(), w, b, l2norm=True)
        if out.size(1) != n:
            out = out[:, :n].contiguous()
        return out
