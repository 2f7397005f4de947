"""Process-group bootstrap with the reference's entry points (ibl/utils/dist_utils.py).
backend='nccl' selects RCCL on ROCm; one process per GPU."""
import os
import subprocess

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def init_dist(launcher, args, backend='nccl'):
    if mp.get_start_method(allow_none=True) is None:
        mp.set_start_method('spawn')
    if launcher == 'pytorch':
        init_dist_pytorch(args, backend)
    elif launcher == 'slurm':
        init_dist_slurm(args, backend)
    else:
        raise ValueError('Invalid launcher type: {}'.format(launcher))


def init_dist_pytorch(args, backend="nccl"):
    # single-node: the local rank doubles as the global rank, as in the reference
    args.rank = int(os.environ['LOCAL_RANK'])
    args.ngpus_per_node = torch.cuda.device_count()
    args.gpu = args.rank
    args.world_size = int(os.environ.get('WORLD_SIZE', args.ngpus_per_node))
    torch.cuda.set_device(args.gpu)
    dist.init_process_group(backend=backend)


def init_dist_slurm(args, backend="nccl"):
    args.rank = int(os.environ['SLURM_PROCID'])
    args.world_size = int(os.environ['SLURM_NTASKS'])
    args.ngpus_per_node = torch.cuda.device_count()
    args.gpu = args.rank % args.ngpus_per_node
    torch.cuda.set_device(args.gpu)
    addr = subprocess.getoutput(
        'scontrol show hostname {} | head -n1'.format(os.environ['SLURM_NODELIST']))
    os.environ['MASTER_PORT'] = str(args.tcp_port)
    os.environ['MASTER_ADDR'] = addr
    os.environ['WORLD_SIZE'] = str(args.world_size)
    os.environ['RANK'] = str(args.rank)
    dist.init_process_group(backend=backend)
    args.total_gpus = dist.get_world_size()


def synchronize():
    """Barrier across all processes (no-op without a multi-process group)."""
    if not dist.is_available() or not dist.is_initialized():
        return
    if dist.get_world_size() == 1:
        return
    dist.barrier()
