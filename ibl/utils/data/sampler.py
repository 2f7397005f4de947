from __future__ import absolute_import

import torch.distributed as dist
# examples/cluster.py imports SubsetRandomSampler from this module (the reference's sampler.py:9-11
# pulls torch's samplers into its namespace)
from torch.utils.data import BatchSampler  # noqa: F401
from torch.utils.data.sampler import (  # noqa: F401
    RandomSampler, Sampler, SequentialSampler, SubsetRandomSampler, WeightedRandomSampler)

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


class DistributedRandomTupleSampler(Sampler):
    """Tuples for hard-negative mining — `[anchor, easiest positive, neg_num hardest negatives]`, the
    gallery entries offset by len(query_source) — with the contract of the reference's sampler
    (ibl/utils/data/sampler.py:15-86; examples/netvlad_img.py:42-46).

    What is accelerated is `sort_gallery`: the reference calls `torch.argsort(distmat, dim=1)` over
    the whole query x gallery matrix on the host (sampler.py:49); here the full-row ranking runs on
    the GPU (`oibl_row_argsort`, stable: ties go to the lowest gallery index).  The tuple bookkeeping
    below is host logic restated with numpy; it draws from `random` exactly like the reference
    (one `random.sample(range(#candidates), min(neg_pool, #candidates))` per anchor), so a seeded
    run yields the same tuples (tests/golden/tuple_sampler.npz)."""

    def __init__(self, query_source, gallery_source, pos_list, neg_list, neg_num=10, neg_pool=1000,
                 sub_length=None, num_replicas=None, rank=None):
        self.num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
        self.rank = dist.get_rank() if rank is None else rank
        self.epoch = 0
        self.query_source, self.gallery_source = query_source, gallery_source
        self.pos_list, self.neg_list = pos_list, neg_list
        self.neg_num, self.neg_pool = neg_num, neg_pool
        self.sort_idx = None
        self.neg_cache = [[] for _ in range(len(query_source))]
        self._set_subset(list(range(len(query_source))))
        if sub_length is not None:
            self.sub_length = sub_length      # (the reference leaves the derived sizes to sort_gallery)

    def _set_subset(self, sub_set):
        self.sub_set = list(sub_set)
        self.sub_length = len(self.sub_set)
        self.sub_length_dist = -(-self.sub_length // self.num_replicas)
        self.total_size = self.sub_length_dist * self.num_replicas

    def sort_gallery(self, distmat, sub_set):
        """Rank the whole gallery for every query (ascending distance) and select this epoch's anchors."""
        assert distmat.shape[0] == len(self.query_source) and distmat.shape[1] == len(self.gallery_source)
        import torch
        from openibl_amd import ops
        d = torch.as_tensor(distmat, dtype=torch.float32)
        if not d.is_cuda:
            d = d.to(torch.device("cuda", torch.cuda.current_device()))
        self.sort_idx = ops.row_argsort(d.contiguous()).cpu().long()
        self._set_subset(sub_set)

    def __len__(self):
        return self.sub_length_dist

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _my_slots(self):
        slots = list(range(self.sub_length))
        slots += slots[: self.total_size - len(slots)]        # wrap-around padding
        return slots[self.rank: self.total_size: self.num_replicas]

    def __iter__(self):
        import random
        import numpy as np
        offset = len(self.query_source)
        for slot in self._my_slots():
            anchor = self.sub_set[slot]
            ranked = np.asarray(self.sort_idx[anchor])
            # easiest positive: the best-ranked gallery entry among the anchor's positives
            positive = int(ranked[np.isin(ranked, self.pos_list[anchor])][0])
            # hardest negatives: best-ranked entries outside the anchor's exclusion zone, restricted
            # to a random pool plus the negatives used last time
            cand = ranked[~np.isin(ranked, self.neg_list[anchor])]
            pool = set(random.sample(range(len(cand)), min(self.neg_pool, len(cand))))
            if self.neg_cache[anchor]:
                where = {int(g): i for i, g in enumerate(cand.tolist())}
                pool |= {where[g] for g in self.neg_cache[anchor]}
            picked = [int(cand[i]) for i in sorted(pool)[: self.neg_num]]
            assert len(picked) == self.neg_num
            self.neg_cache[anchor] = picked
            yield [anchor, positive + offset] + [g + offset for g in picked]


class DistributedRandomDiffTupleSampler(DistributedRandomTupleSampler):
    """The SFRS mining sampler (ibl/utils/data/sampler.py:92-190 of the reference;
    examples/netvlad_img_sfrs.py): tuples `[anchor, easiest positive, neg_num hardest negatives,
    up to pos_num "difficult" positives]`.  As above, what runs on the GPU is the full-row ranking of
    `sort_gallery` (sampler.py:130: `torch.argsort(distmat, dim=1)`); the k-reciprocal matrix
    `distmat_jac` is only indexed.

    Difficult positives (sampler.py:158-178): of the anchor's best-ranked `pos_pool` positives, those
    that the Jaccard distance ranks HIGHER than the descriptor distance does, largest promotion first,
    then the ones both rankings agree on.  The two small sorts are `torch.argsort` calls on the same
    values as in the reference (its tie behaviour on equal promotions is torch's, so torch is asked);
    a seeded run yields the same tuples (tests/golden/diff_tuple_sampler.npz)."""

    def __init__(self, query_source, gallery_source, pos_list, neg_list, pos_num=10, pos_pool=20,
                 neg_num=10, neg_pool=1000, sub_length=None, num_replicas=None, rank=None):
        super().__init__(query_source, gallery_source, pos_list, neg_list, neg_num=neg_num, neg_pool=neg_pool,
                         sub_length=sub_length, num_replicas=num_replicas, rank=rank)
        self.pos_num, self.pos_pool = pos_num, pos_pool
        self.distmat_jac = None

    def sort_gallery(self, distmat, distmat_jac, sub_set):
        super().sort_gallery(distmat, sub_set)
        self.distmat_jac = distmat_jac

    def _difficult_positives(self, anchor, ranked_pos):
        import torch
        pool = torch.as_tensor(ranked_pos[: self.pos_pool], dtype=torch.long)
        jac = torch.as_tensor(self.distmat_jac[anchor])[pool]
        by_jac = torch.argsort(jac, dim=0)                  # by_jac[i]: descriptor rank of the i-th by Jaccard
        gap = torch.arange(by_jac.size(0)) - by_jac         # < 0: promoted by the Jaccard distance
        slots = torch.arange(by_jac.size(0))
        promoted = slots[gap < 0][torch.argsort(gap[gap < 0], dim=0)]
        chosen = torch.cat((promoted, slots[gap == 0]), dim=0)[: self.pos_num]
        return pool[by_jac[chosen]].tolist()

    def __iter__(self):
        import numpy as np
        offset = len(self.query_source)
        base = super().__iter__()
        for slot in self._my_slots():
            anchor = self.sub_set[slot]
            ranked = np.asarray(self.sort_idx[anchor])
            ranked_pos = ranked[np.isin(ranked, self.pos_list[anchor])].tolist()
            extra = self._difficult_positives(anchor, ranked_pos)   # (no RNG: before or after `base` alike)
            yield next(base) + [p + offset for p in extra]
