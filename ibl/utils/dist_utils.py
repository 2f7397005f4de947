"""Process-group bootstrap behind the reference's entry point `init_dist(launcher, args)`
(ibl/utils/dist_utils.py:11-45) — one process per GPU, backend 'nccl' = RCCL on ROCm.

Each launcher is a function that reads its environment into a `_Placement` (global rank, world
size, local device, rendezvous variables to export); `init_dist` applies it to `args` the way the
reference's scripts expect (`args.rank`, `args.gpu`, `args.world_size`, `args.ngpus_per_node`) and
creates the group.  `--launcher none` is an error, as in the reference."""
import os
import subprocess
from collections import namedtuple

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

_Placement = namedtuple("_Placement", "rank world_size gpu env")


def _torchrun(args, ngpus):
    # single node: the local rank doubles as the global rank, as in the reference; WORLD_SIZE (set by
    # torch.distributed.run) wins over the visible device count when fewer processes are started
    local = int(os.environ['LOCAL_RANK'])
    return _Placement(local, int(os.environ.get('WORLD_SIZE', ngpus)), local, {})


def _slurm(args, ngpus):
    rank, world = int(os.environ['SLURM_PROCID']), int(os.environ['SLURM_NTASKS'])
    head = subprocess.getoutput(
        'scontrol show hostname {} | head -n1'.format(os.environ['SLURM_NODELIST']))
    env = {'MASTER_PORT': str(args.tcp_port), 'MASTER_ADDR': head, 'WORLD_SIZE': str(world),
           'RANK': str(rank)}
    return _Placement(rank, world, rank % ngpus, env)


_LAUNCHERS = {'pytorch': _torchrun, 'slurm': _slurm}


def init_dist(launcher, args, backend='nccl'):
    if launcher not in _LAUNCHERS:
        raise ValueError('Invalid launcher type: {}'.format(launcher))
    if mp.get_start_method(allow_none=True) is None:
        mp.set_start_method('spawn')
    ngpus = torch.cuda.device_count()
    place = _LAUNCHERS[launcher](args, ngpus)
    os.environ.update(place.env)
    args.rank, args.world_size, args.gpu, args.ngpus_per_node = place.rank, place.world_size, place.gpu, ngpus
    torch.cuda.set_device(args.gpu)
    dist.init_process_group(backend=backend)
    if launcher == 'slurm':
        args.total_gpus = dist.get_world_size()


def init_dist_pytorch(args, backend="nccl"):
    init_dist('pytorch', args, backend)


def init_dist_slurm(args, backend="nccl"):
    init_dist('slurm', args, backend)


def synchronize():
    """Barrier across all processes (no-op without a multi-process group)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def simple_group_split(world_size, rank, num_groups):
    """Process groups of world_size / num_groups consecutive ranks each; returns the one `rank` belongs
    to (ibl/utils/dist_utils.py:44-52: the SyncBN groups of the training scripts — VGG16 itself has
    no normalisation layers, the evaluation path never calls this)."""
    size = world_size // num_groups
    members = [list(range(g * size, (g + 1) * size)) for g in range(num_groups)]
    groups = [dist.new_group(m) for m in members]          # every rank creates every group, in order
    mine = rank // size
    print("Rank no.{} start sync BN on the process group of {}".format(rank, members[mine]))
    return groups[mine]


def convert_sync_bn(model, process_group=None, gpu=None):
    """Replace the batch-norm layers below `model` by SyncBatchNorm, in place (dist_utils.py:54-62)."""
    for name, child in list(model.named_children()):
        converted = torch.nn.SyncBatchNorm.convert_sync_batchnorm(child, process_group)
        if converted is not child:
            setattr(model, name, converted if gpu is None else converted.cuda(gpu))
        elif gpu is not None and any(isinstance(m, torch.nn.SyncBatchNorm) for m in child.modules()):
            child.cuda(gpu)
