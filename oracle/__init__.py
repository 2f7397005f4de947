"""CPU oracle for the descriptor + matching hot path.  TEST INFRASTRUCTURE ONLY.

A restatement, in plain torch-CPU / numpy, of what the reference (yxgeee/OpenIBL) computes on
this path; every function cites the reference lines it follows.  It exists to check the HIP
path and to provide `bench.py`'s cpu_baseline; the only allowed importers are `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg.  Nothing under `openibl_amd/` or
`ibl/` imports it, and no product path ever falls back to it.

Where the arithmetic lives.  The reference has no kernels of its own: the numbers come out of
third-party PyTorch (ATen CPU: oneDNN/MKL convolution and GEMM, vectorised softmax / norms).
The reference pins no versions (setup.py:10-13 lists 'torch', 'torchvision' bare; docs/INSTALL.md:3
says "tested on PyTorch 1.1.0"); the container's torch 2.10.0 CPU path is therefore the
reference arithmetic, and the restatement below is written against torch.nn.functional so that
it differs from the reference only in structure (no nn.Modules, no 157 MB residual tensor), not
in the kernels that produce the numbers.  The VGG16 layer list lives in torchvision (absent
from this image): it is restated from ibl/models/vgg.py:40-42 plus the standard cfg-D.

Parity pinning.  The reference ships no tests, golden vectors or fixtures for this path
("parity unpinned" by the reference itself).  The oracle is instead pinned
against outputs of THE REFERENCE ITSELF, imported from /root/reference in the build container
through `oracle/refshim.py` by `oracle/make_golden.py`; the resulting vectors are committed
under `tests/golden/` and `tests/test_oracle_golden.py` checks the oracle against them on every
run (CPU-only, no access to /root/reference needed).
"""
