"""Executed in a subprocess by tests/test_host_logic.py when /root/reference is present: loads the
reference's OWN examples/cluster.py (unmodified, from where it lies) against THIS repo's `ibl`
package, runs its get_data() on a Pittsburgh-format synthetic dataset and builds the model the way
its get_model() does (without the .cuda() / DataParallel wrap: no GPU here).  h5py is stubbed."""
import sys
sys.dont_write_bytecode = True   # nothing may be written under /root/reference
import argparse
import os
import runpy
import sys
import types

import numpy as np
import torch

root, ref_script, repo = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, repo)
sys.path.insert(0, os.path.join(repo, "tests"))
sys.modules.setdefault("h5py", types.ModuleType("h5py"))
mod = runpy.run_path(ref_script, run_name="reference_examples_cluster")   # not __main__: no parser, no main()
import ibl
assert os.path.abspath(ibl.__file__).startswith(os.path.abspath(repo)), ibl.__file__
np.random.seed(43)
args = argparse.Namespace(data_dir=root, dataset="pitts", height=72, width=96, batch_size=4, workers=0)
dataset, loader = mod["get_data"](args, 8)
n = 0
for imgs, fnames, _, _, _ in loader:
    assert tuple(imgs.shape[1:]) == (3, 72, 96) and imgs.dtype == torch.float32
    n += len(fnames)
assert n == 8
# get_model(): models.create(arch, pretrained=True, cut_at_pooling=True, matconvnet=...) and .feature_dim
# (the off-the-shelf matconvnet backbone file is read at construction, vgg.py:56-60: write one)
mc = os.path.join(root, "vd16_offtheshelf_conv5_3_max.pth")
torch.save(mod["models"].create("vgg16", pretrained=False, cut_at_pooling=True).base.state_dict(), mc)
model = mod["models"].create("vgg16", pretrained=True, cut_at_pooling=True, matconvnet=mc)
assert model.feature_dim == 512 and model.pretrained
assert mod["KMeans"].__module__.startswith("sklearn")
print("REFERENCE_CLUSTER_OK")
