from __future__ import absolute_import

import torch.distributed as dist
from torch.utils.data.sampler import Sampler

from openibl_amd.sharded import slice_bounds


class DistributedSliceSampler(Sampler):
    """Rank r iterates the contiguous index range [r * per, (r + 1) * per), per = ceil(L / W);
    indices past the end wrap to the start (padding that extract_features drops after the
    gather).  Same dealing as the reference's sampler, which is what makes the gallery shard of
    rank r a contiguous block of gallery positions."""

    def __init__(self, dataset, num_replicas=None, rank=None):
        if num_replicas is None:
            num_replicas = dist.get_world_size()
        if rank is None:
            rank = dist.get_rank()
        self.dataset = dataset
        self.num_replicas = num_replicas
        self.rank = rank
        start, per, _ = slice_bounds(len(dataset), rank, num_replicas)
        self.num_samples = per
        self.total_size = per * num_replicas
        n = len(dataset)
        self._indices = [(start + i) % n for i in range(per)] if n else []

    def __iter__(self):
        return iter(self._indices)

    def __len__(self):
        return self.num_samples
