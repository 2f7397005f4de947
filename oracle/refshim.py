"""Import the real reference from /root/reference in THIS container (golden generation only).

The reference needs torchvision (for the vgg16 layer list), h5py and cv2, none of which are
installed.  Minimal stand-ins are registered in sys.modules before `ibl` is imported; no file
of the reference is modified or copied.  /root/reference does not exist on the GPU box, so
nothing that runs there may import this module.
"""
from __future__ import annotations

import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = "/root/reference"
_CFG_D = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


class _VGGFeatures(nn.Module):
    """torchvision.models.vgg16().features: conv3x3(pad 1)+ReLU(inplace) per entry, MaxPool2d(2,2)
    per 'M' (31 modules; the reference drops the last two, vgg.py:41)."""

    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in _CFG_D:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)


def _vgg16(pretrained=False, **kwargs):
    if pretrained:
        raise RuntimeError("no network: ImageNet weights are not available")
    return _VGGFeatures()


def install():
    """Register the stand-in modules and put the reference on sys.path.  Idempotent."""
    sys.dont_write_bytecode = True  # never create __pycache__ inside /root/reference
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.models = types.ModuleType("torchvision.models")
        tv.models.vgg16 = _vgg16
        tv.transforms = types.ModuleType("torchvision.transforms")
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.models"] = tv.models
        sys.modules["torchvision.transforms"] = tv.transforms
    for name in ("h5py", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def init_process_group():
    """ibl.evaluators calls dist.get_rank() unconditionally (evaluators.py:116,147)."""
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29591", rank=0, world_size=1)


def reference_model(state_dict):
    """hubconf.vgg16_netvlad(pretrained=False) with `state_dict` loaded, in eval mode."""
    install()
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_hubconf", REFERENCE_ROOT + "/hubconf.py")
    hub = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hub)
    torch.manual_seed(0)
    model = hub.vgg16_netvlad(pretrained=False)
    model.load_state_dict(state_dict)
    return model.eval()
