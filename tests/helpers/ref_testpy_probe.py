"""Executed in a subprocess by tests/test_host_logic.py when /root/reference is present: loads the
reference's OWN examples/test.py (unmodified, from where it lies) against THIS repo's `ibl`
package and runs its get_data() on a Pittsburgh-format synthetic dataset.  h5py (imported at the
top of test.py, absent from the image) is stubbed; nothing else is."""
import sys
sys.dont_write_bytecode = True   # nothing may be written under /root/reference
import argparse
import os
import runpy
import sys
import types

import torch
import torch.distributed as dist

root, ref_test, repo = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, repo)
sys.path.insert(0, os.path.join(repo, "tests"))
sys.modules.setdefault("h5py", types.ModuleType("h5py"))
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29613", rank=0, world_size=1)
mod = runpy.run_path(ref_test, run_name="reference_examples_test")     # not __main__: no parser, no main()
import ibl
assert os.path.abspath(ibl.__file__).startswith(os.path.abspath(repo)), ibl.__file__
args = argparse.Namespace(data_dir=root, dataset="pitts", scale="30k", height=72, width=96,
                          test_batch_size=4, workers=0)
dataset, pitts_train, train_loader, loader_q, loader_db = mod["get_data"](args)
assert len(pitts_train) == 20 and len(dataset.q_test) == 6 and len(dataset.db_test) == 14
assert all(len(p) == 1 for p in dataset.test_pos)
n = 0
for batch in loader_db:
    imgs, fnames = batch[0], batch[1]
    assert tuple(imgs.shape[1:]) == (3, 72, 96) and imgs.dtype == torch.float32
    n += len(fnames)
assert n == 14
assert sum(len(b[1]) for b in train_loader) == 20 and sum(len(b[1]) for b in loader_q) == 6
print("REFERENCE_GET_DATA_OK")
dist.destroy_process_group()
