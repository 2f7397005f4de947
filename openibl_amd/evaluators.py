"""Mirror of ibl.evaluators (ibl/evaluators.py): same names, arguments and return values, with the
device work routed to the HIP kernels.

    extract_cnn_feature   evaluators.py:22-34    forward + F.normalize -> oibl_l2_normalize_rows
    extract_features      evaluators.py:36-103   batched extraction, cross-rank gather, fname dict
    pairwise_distance     evaluators.py:105-130  host GEMM -> oibl_pairwise_sqdist on the GPU
    spatial_nms           evaluators.py:132-140
    evaluate_all          evaluators.py:142-167  full argsort -> oibl_row_topk of the needed prefix
    Evaluator.evaluate    evaluators.py:176-201

`Evaluator.evaluate` additionally has a device-resident route (default when re-ranking is off):
descriptors stay on the GPU that produced them, the gallery is matched shard-by-shard
(openibl_amd.sharded) and only top-k lists are exchanged; its recalls equal the reference flow's.
"""
from __future__ import annotations

import time
from collections import OrderedDict
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import ops, sharded
from .models import default_precision


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _device(gpu=None) -> torch.device:
    return torch.device("cuda", torch.cuda.current_device() if gpu is None else gpu)


def _to_tensor(x):
    if torch.is_tensor(x):
        return x
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x)
    raise ValueError("Cannot convert {} to torch tensor".format(type(x)))


class _Meter:
    def __init__(self):
        self.val = self.sum = self.count = 0.0

    def update(self, v):
        self.val = v
        self.sum += v
        self.count += 1

    @property
    def avg(self):
        return self.sum / max(self.count, 1)


def extract_cnn_feature(model, inputs, vlad=True, gpu=None, scales=None):
    """Forward one batch and L2-normalise the selected output (evaluators.py:22-34).
    `scales` (extension, BASELINE.json configs[4]): multi-scale extraction, see multiscale.py."""
    model.eval()
    x = _to_tensor(inputs).to(_device(gpu), non_blocking=True)
    if scales is not None:
        from .multiscale import extract_multiscale
        return extract_multiscale(model, x, scales, vlad)
    out = model(x)
    if isinstance(out, (list, tuple)):
        pool_x, vlad_x = out
        out = vlad_x if vlad else pool_x
    return ops.l2_normalize(out.float().contiguous())


FAST_EXTRACTION = True   # False: batch-by-batch eager launches (the cross-check of the replayed path)


def _extract_local(model, data_loader, vlad, pca, gpu, print_freq, rank, scales=None, store_dtype=None,
                   use_graphs=True):
    """Run the loader of this rank; returns the [n_local][d] descriptor matrix ON DEVICE, in
    `store_dtype` (None = float32; float16 / bfloat16 = 16-bit descriptor storage).

    Embed* models take the replayed route of openibl_amd.extract (hipGraph per batch shape, batches
    alternating between two lanes, H2D on a copy stream, descriptors written straight into a
    pre-sized matrix); other modules and multi-scale extraction run batch by batch.  Same kernels,
    same bits."""
    model.eval()
    if pca is not None:
        pca.load(gpu=gpu)
    from . import extract
    if FAST_EXTRACTION and scales is None and extract.fast_path_supported(model):
        return extract.extract_descriptors(model, data_loader, vlad=vlad, pca=pca, gpu=gpu,
                                           print_freq=print_freq, rank=rank, store_dtype=store_dtype,
                                           use_graphs=use_graphs)
    batch_t, data_t = _Meter(), _Meter()
    chunks = []
    end = time.time()
    with torch.no_grad():
        for i, batch in enumerate(data_loader):
            imgs = batch[0]
            data_t.update(time.time() - end)
            out = extract_cnn_feature(model, imgs, vlad, gpu=gpu, scales=scales)
            if pca is not None:
                out = pca.infer(out)
            chunks.append(ops.store_descriptors(out, store_dtype))
            batch_t.update(time.time() - end)
            end = time.time()
            if (i + 1) % print_freq == 0 and rank == 0:
                print("Extract Features: [{}/{}]\tTime {:.3f} ({:.3f})\tData {:.3f} ({:.3f})\t".format(
                    i + 1, len(data_loader), batch_t.val, batch_t.avg, data_t.val, data_t.avg))
    if not chunks:
        return torch.empty((0, 0), device=_device(gpu))
    return torch.cat(chunks)


def _host_f32(t: torch.Tensor) -> torch.Tensor:
    """Device descriptors (any storage type) -> float32 on the host; 16-bit rows are widened by the
    HIP cast before they leave the device."""
    if t.is_cuda and t.dtype != torch.float32 and t.numel():
        t = ops.load_descriptors(t.contiguous())
    return t.to("cpu", copy=True).float()


def _gather_all(local: torch.Tensor, sync_gather: bool, rank: int, world: int) -> torch.Tensor:
    """Every rank's [per][d] block -> [world * per][d] on the host, rank-major."""
    if world == 1:
        return _host_f32(local)
    if sync_gather:  # one all_gather (evaluators.py:76-88)
        return _host_f32(sharded.all_gather_rows(local))
    parts = []  # rank-by-rank broadcast, one block resident at a time (evaluators.py:89-101)
    buf = torch.empty_like(local)
    for k in range(world):
        if k == rank:
            buf.copy_(local)
        if rank == 0:
            print("gathering features from rank no.{}".format(k))
        dist.broadcast(buf, k)
        parts.append(_host_f32(buf))   # (a copy: the buffer is reused for the next rank)
    return torch.cat(parts)


def extract_features(model, data_loader, dataset, print_freq=10, vlad=True, pca=None, gpu=None,
                     sync_gather=False, scales=None, store_dtype=None, use_graphs=True):
    """OrderedDict fname -> CPU descriptor for every item of `dataset` (evaluators.py:36-103).
    `use_graphs=False`: the two-lane route without hipGraph capture (loaders whose batch shapes rarely
    repeat; `openibl_amd.extract.MAX_CACHED_SHAPES` bounds the captured shapes kept otherwise).

    Each rank extracts the slice its DistributedSliceSampler yields; slices are gathered
    rank-major and truncated to len(dataset) (the wrap-around padding of the last slices)."""
    rank, world = _rank_world()
    local = _extract_local(model, data_loader, vlad, pca, gpu, print_freq, rank, scales, store_dtype, use_graphs)
    allf = _gather_all(local, sync_gather, rank, world)[: len(dataset)]
    features = OrderedDict()
    for item, row in zip(dataset, allf):
        features[item[0]] = row
    return features


def pairwise_distance(features, query=None, gallery=None, metric=None, precision=None):
    """Squared-L2 matrix between query and gallery descriptors (evaluators.py:105-130).

    Returns (dist_m [m][n] float32 CPU tensor, x.numpy(), y.numpy()) like the reference; the
    contraction runs on the GPU.  With query = gallery = None: all-pairs mode (evaluators.py:106-114,
    which assumes equal row norms: 2|x_i|^2 - 2 x_i.x_j)."""
    prec = precision or default_precision()
    dev = _device()
    if query is None and gallery is None:
        n = len(features)
        x = torch.stack([v.reshape(-1) for v in features.values()]).float()
        if metric is not None:
            x = metric.transform(x)
        xd = x.to(dev).contiguous()
        d = ops.pairwise_sqdist(xd, xd, prec)
        # |x_i|^2 + |x_j|^2 - 2 x.x  ->  reference's 2|x_i|^2 - 2 x.x
        sq = (xd * xd).sum(dim=1)
        d = d + (sq[:, None] - sq[None, :])
        return d.cpu(), None, None

    rank, _ = _rank_world()
    if rank == 0:
        print("===> Start calculating pairwise distances")
    x = torch.stack([features[f].reshape(-1) for f, _, _, _ in query]).float()
    y = torch.stack([features[f].reshape(-1) for f, _, _, _ in gallery]).float()
    if metric is not None:
        x = metric.transform(x)
        y = metric.transform(y)
    d = ops.pairwise_sqdist(x.to(dev).contiguous(), y.to(dev).contiguous(), prec)
    return d.cpu(), x.numpy(), y.numpy()


def spatial_nms(pred, db_ids, topN):
    """Keep, among the first topN predictions, the first occurrence of every place id
    (evaluators.py:132-140)."""
    assert len(pred) == len(db_ids)
    seen = set()
    keep = []
    for i in pred[:topN]:
        pid = db_ids[i]
        if pid not in seen:
            seen.add(pid)
            keep.append(i)
    return keep


def recalls_from_topk(topk_idx: np.ndarray, gt: Sequence[Sequence[int]],
                      gallery_pids: Optional[Sequence[int]] = None,
                      recall_topk: Sequence[int] = (1, 5, 10), nms: bool = False) -> np.ndarray:
    """Recall@N from the ranked prefix of every row (the counting of evaluators.py:149-160).

    topk_idx [m][k] holds, per query, the k nearest gallery positions in ascending distance;
    k >= max(recall_topk), or >= 12 * max(recall_topk) when nms is on (evaluators.py:152-153)."""
    correct = np.zeros(len(recall_topk))
    for q, pred in enumerate(topk_idx):
        pred = [int(p) for p in pred if p >= 0]
        if nms:
            seen, keep = set(), []
            for i in pred[: max(recall_topk) * 12]:
                pid = gallery_pids[i]
                if pid not in seen:
                    seen.add(pid)
                    keep.append(i)
            pred = keep
        truth = set(int(t) for t in gt[q])
        for i, n in enumerate(recall_topk):
            if any(p in truth for p in pred[:n]):
                correct[i:] += 1
                break
    return correct / len(gt)


def recalls_from_topk_device(topk_idx: torch.Tensor, gt: Sequence[Sequence[int]],
                             gallery_pids: Optional[Sequence[int]] = None,
                             recall_topk: Sequence[int] = (1, 5, 10), nms: bool = False) -> np.ndarray:
    """Same numbers as recalls_from_topk with the per-query loop on the GPU
    (oibl_first_hit_rank): the ranked lists never leave the device, only m integers do."""
    dev = topk_idx.device
    lens = np.fromiter((len(g) for g in gt), dtype=np.int64, count=len(gt))
    off = np.zeros(len(gt) + 1, dtype=np.int32)
    np.cumsum(lens, out=off[1:])
    vals = np.fromiter((int(t) for g in gt for t in g), dtype=np.int32, count=int(off[-1]))
    pids = None
    if nms:
        pids = torch.as_tensor(np.asarray(gallery_pids, dtype=np.int64).astype(np.int32), device=dev)
    ranks = ops.first_hit_rank(topk_idx.to(torch.int32), torch.from_numpy(off).to(dev),
                               torch.from_numpy(vals if len(vals) else np.zeros(1, np.int32)).to(dev),
                               pids, nms_window=max(recall_topk) * 12).cpu().numpy()
    correct = np.zeros(len(recall_topk))
    for r in ranks:                                # evaluators.py:155-159, on the first-hit rank
        if r < 0:
            continue
        for i, n in enumerate(recall_topk):
            if r < n:
                correct[i:] += 1
                break
    return correct / len(gt)


MAX_RANK_PREFIX = 1024   # longest ranked prefix oibl_row_topk / oibl_first_hit_rank produce


def _check_prefix(k: int) -> None:
    """The reference counts over max(recall_topk) predictions, 12x that with nms
    (evaluators.py:152-153).  A prefix the kernels cannot produce is an error, never a silent
    truncation: truncated lists would change Recall@N."""
    if k > MAX_RANK_PREFIX:
        raise ValueError(f"recall counting needs the {k} nearest gallery entries per query; the "
                         f"device ranking is limited to {MAX_RANK_PREFIX} (max(recall_topk) <= "
                         f"{MAX_RANK_PREFIX}, or <= {MAX_RANK_PREFIX // 12} with nms=True)")


def _print_recalls(recalls, recall_topk):
    print("Recall Scores:")
    for i, k in enumerate(recall_topk):
        print("  top-{:<4}{:12.1%}".format(k, recalls[i]))


def evaluate_all(distmat, gt, gallery, recall_topk=[1, 5, 10], nms=False):
    """Recall@N from a distance matrix (evaluators.py:142-167).  The reference argsorts every full
    row; only the first max(recall_topk) (x12 with nms) entries are ever read, so the rows are
    reduced on the GPU by the top-k kernel (ties: lowest gallery index first)."""
    rank, _ = _rank_world()
    d = _to_tensor(distmat).float()
    k = min(max(recall_topk) * (12 if nms else 1), d.shape[1])
    dev = _device()
    if rank == 0:
        print("===> Start calculating recalls")
    if k <= MAX_RANK_PREFIX:
        _, idx = ops.row_topk(d.to(dev).contiguous(), k)
        recalls = recalls_from_topk_device(idx, gt, [g[1] for g in gallery], recall_topk, nms)
    else:
        # more ranks than the selection kernels keep: full-row ranking on the device
        # (oibl_row_argsort), counting with the host loop of the reference
        idx = ops.row_argsort(d.to(dev).contiguous())[:, :k].cpu().numpy()
        recalls = recalls_from_topk(idx, gt, [g[1] for g in gallery], recall_topk, nms)
    if rank == 0:
        _print_recalls(recalls, recall_topk)
    return recalls


class Evaluator(object):
    def __init__(self, model, precision: Optional[str] = None, scales=None, descriptor_dtype=None):
        """`scales` / `descriptor_dtype` are the BASELINE.json configs[4] extensions (multi-scale
        extraction, 16-bit descriptor storage); the defaults reproduce the reference."""
        super(Evaluator, self).__init__()
        self.model = model
        self.rank = _rank_world()[0]
        self.precision = precision
        self.scales = scales
        self.descriptor_dtype = descriptor_dtype

    # -- reference flow: fname dict on the host, full distance matrix ----------------------------
    def _evaluate_host(self, query_loader, dataset, query, gallery, ground_truth, gallery_loader,
                       vlad, pca, rerank, gpu, sync_gather, nms, rr_topk, lambda_value):
        if gallery_loader is not None:
            ext = dict(vlad=vlad, pca=pca, gpu=gpu, sync_gather=sync_gather, scales=self.scales,
                       store_dtype=self.descriptor_dtype)
            features = extract_features(self.model, query_loader, query, **ext)
            features.update(extract_features(self.model, gallery_loader, gallery, **ext))
        else:
            features = extract_features(self.model, query_loader, dataset, vlad=vlad, pca=pca,
                                        gpu=gpu, sync_gather=sync_gather, scales=self.scales,
                                        store_dtype=self.descriptor_dtype)
        distmat, _, _ = pairwise_distance(features, query, gallery, precision=self.precision)
        recalls = evaluate_all(distmat, ground_truth, gallery, nms=nms)
        if not rerank:
            return recalls
        from .rerank import re_ranking
        if self.rank == 0:
            print("Applying re-ranking ...")
        distmat_gg, _, _ = pairwise_distance(features, gallery, gallery, precision=self.precision)
        distmat_qq, _, _ = pairwise_distance(features, query, query, precision=self.precision)
        distmat = re_ranking(distmat.numpy(), distmat_qq.numpy(), distmat_gg.numpy(), k1=rr_topk,
                             k2=1, lambda_value=lambda_value)
        return evaluate_all(distmat, ground_truth, gallery, nms=nms)

    # -- device-resident flow: gallery stays sharded, only top-k lists travel ----------------------
    def _evaluate_device(self, query_loader, query, gallery, ground_truth, gallery_loader, vlad,
                         pca, gpu, nms, recall_topk=(1, 5, 10)):
        rank, world = _rank_world()
        prec = self.precision or default_precision()
        q_local = _extract_local(self.model, query_loader, vlad, pca, gpu, 10, rank, self.scales,
                                 self.descriptor_dtype)
        g_local = _extract_local(self.model, gallery_loader, vlad, pca, gpu, 10, rank, self.scales,
                                 self.descriptor_dtype)
        # This flow relies on the dealing of DistributedSliceSampler (contiguous slices of
        # ceil(L / W) items, wrap-around padding): global gallery indices come from slice_bounds.
        # A loader with another sampler / drop_last would silently shift them — refuse instead.
        for what, local, items in (("query", q_local, query), ("gallery", g_local, gallery)):
            per = sharded.slice_bounds(len(items), rank, world)[1]
            if local.shape[0] != per:
                raise ValueError(
                    f"Evaluator.evaluate(device_resident=True): the {what} loader of rank {rank} "
                    f"yielded {local.shape[0]} items, DistributedSliceSampler would yield {per}; "
                    "use DistributedSliceSampler(dataset) without drop_last, or pass "
                    "device_resident=False for the reference's host flow")
        # queries: prepared where they were extracted, exchanged in prepared form; gallery shard:
        # prepared once and resident
        k = min(max(recall_topk) * (12 if nms else 1), len(gallery))
        _check_prefix(k)
        # an f16mx model's fp32 descriptors are matched by the fp16 filter + exact rescoring (fp32-exact lists,
        # faster than f16mx on every pair): ops.topk_precision
        prec = ops.topk_precision(prec, q_local.dtype if q_local.dtype == g_local.dtype else None, k)
        q_all = sharded.gather_prepared_queries(q_local, len(query), prec)
        start, _, n_valid = sharded.slice_bounds(len(gallery), rank, world)
        g_local = ops.PreparedRows(g_local[:n_valid].contiguous(), prec)
        if rank == 0:
            print("===> Start calculating pairwise distances")
        _, idx = sharded.sharded_topk(q_all, g_local, k, start, prec)
        if rank == 0:
            print("===> Start calculating recalls")
        recalls = recalls_from_topk_device(idx, ground_truth, [g[1] for g in gallery],
                                           recall_topk, nms)
        if rank == 0:
            _print_recalls(recalls, recall_topk)
        return recalls

    def evaluate(self, query_loader, dataset, query, gallery, ground_truth, gallery_loader=None,
                 vlad=True, pca=None, rerank=False, gpu=None, sync_gather=False, nms=False,
                 rr_topk=25, lambda_value=0, device_resident=True):
        if device_resident and gallery_loader is not None and not rerank:
            return self._evaluate_device(query_loader, query, gallery, ground_truth,
                                         gallery_loader, vlad, pca, gpu, nms)
        return self._evaluate_host(query_loader, dataset, query, gallery, ground_truth,
                                   gallery_loader, vlad, pca, rerank, gpu, sync_gather, nms,
                                   rr_topk, lambda_value)
