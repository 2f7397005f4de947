"""openibl_amd — MI355X (gfx950) native descriptor + matching path behind OpenIBL's API.

Layout
  csrc/            hand-written HIP kernels + the C ABI (include/openibl_amd.h)
  build.py         hipcc driver (in-tree libopenibl_amd.so)
  lib.py           ctypes binding, fails loudly when the library is absent
  ops.py           tensor-level wrappers over the C ABI
  models.py        nn.Module mirror of ibl.models (vgg16 / netvlad / embednet / embednetpca ...)
  pca.py           mirror of ibl.pca.PCA (load / infer)
  evaluators.py    mirror of ibl.evaluators (extract_*, pairwise_distance, evaluate_all, Evaluator)
  sharded.py       gallery-sharded top-k matching over torch.distributed (RCCL)

The `ibl` package at the repository root re-exports these under the reference's module names so
that the reference's examples/test.py runs unchanged.
"""

__version__ = "0.1.0"
