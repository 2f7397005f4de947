"""One 480x640 image through the replayed forward, 60 times (for rocprofv3 --kernel-trace --stats: where a
single image's latency goes).      python tests/gpu_single_image_prof.py [precision]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import hubconf  # noqa: E402
from openibl_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
m = hubconf.vgg16_netvlad(pretrained=False)
m.load_state_dict(synth.embednetpca_state(0))
m = m.to(dev).eval().set_precision(prec)
x = synth.images(1, 480, 640, seed=3).to(dev)
with torch.no_grad():
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(60):
        m(x)
    b.record()
    torch.cuda.synchronize()
print(f"{prec}: {a.elapsed_time(b) / 60:.4f} ms per image (eager, back to back)")
