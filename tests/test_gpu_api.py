"""The reference's evaluation API end to end on the GPU: PCA.load / PCA.infer through the class,
Evaluator.evaluate(pca=...), the call sequence of examples/test.py's main_worker on a
Pittsburgh-format dataset, BASELINE.json configs[1] (32 distinct 480x640 images) against the oracle,
and the replayed extraction route against the batch-by-batch one."""
import argparse
import os
import os.path as osp

import numpy as np
import pytest
import torch
import torch.distributed as dist

from conftest import assert_rel_l2, load_golden, rel_l2
from openibl_amd import ops, synth
from oracle import descriptor as od
from oracle import matching as om

pytestmark = pytest.mark.gpu
_CACHE = {}


@pytest.fixture(scope="module")
def group(dev):
    """One-rank RCCL group (what `--launcher pytorch` creates): Evaluator / samplers call dist.get_rank()
    unconditionally and test.py wraps the model in DistributedDataParallel."""
    torch.cuda.set_device(dev)
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29581", rank=0, world_size=1,
                                device_id=dev)
    yield
    dist.destroy_process_group()


def _write_npz_params(path, g):
    os.makedirs(osp.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        np.savez(f, U=g["U"], lams=g["lams"], mu=g["mu"], Utmu=g["Utmu"])


@pytest.mark.parametrize("whiten", [True, False])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_pca_class_load_infer_matches_reference(dev, tmp_path, whiten, precision):
    """ibl.pca.PCA(...).load(gpu); .infer(data) against the output of the reference's own
    PCA.load / PCA.infer on the same parameter arrays (tests/golden/pca.npz, ibl/pca.py:86-123)."""
    from ibl.pca import PCA
    g = load_golden("pca")
    path = str(tmp_path / "logs" / "pca_params.h5")
    _write_npz_params(path, g)
    pca = PCA(int(g["n_components"]), whiten, path, precision=precision)
    with pytest.raises(RuntimeError):
        pca.infer(torch.from_numpy(g["data"]).to(dev))
    pca.load(gpu=dev.index)
    tag = "whiten" if whiten else "nowhiten"
    assert tuple(pca.weight.shape) == (int(g["n_components"]), 256, 1, 1) and pca.weight.is_cuda
    assert_rel_l2(f"PCA.weight {tag}", pca.weight.reshape(-1, 256).cpu(), g[f"weight_{tag}"], 1e-6)
    assert_rel_l2(f"PCA.bias {tag}", pca.bias.cpu(), g[f"bias_{tag}"], 1e-5)
    out = pca.infer(torch.from_numpy(g["data"]).to(dev))
    assert tuple(out.shape) == (300, int(g["n_components"]))
    assert_rel_l2(f"PCA.infer {tag} {precision}", out.cpu(), g[f"out_{tag}"], 1e-5)


class _Records(torch.utils.data.Dataset):
    def __init__(self, images, records):
        self.images, self.records = images, records

    def __len__(self):
        return len(self.records)

    def __getitem__(self, i):
        f, pid, x, y = self.records[i]
        return self.images[i], f, pid, x, y


def _oracle_pca(vlad_norm, path, whiten):
    """normalize(W v + b) from a parameter file, restating ibl/pca.py:96-106, 117-121 in fp64."""
    import openibl_amd.pca as pmod
    U, lams, mu, _ = pmod._read_params(path)
    U, lams, mu = torch.from_numpy(U).double(), torch.from_numpy(lams).double(), torch.from_numpy(mu).double()
    if whiten:
        U = U @ torch.diag(1.0 / torch.sqrt(lams))
    y = (vlad_norm.double() - mu.t()) @ U
    return torch.nn.functional.normalize(y, dim=1).float()


def test_evaluate_with_external_pca_matches_oracle(group, state_dict, dev, tmp_path):
    """The reference's default evaluation path (examples/test.py:108-131, scripts/test_dist.sh:
    --vlad --reduction): EmbedNet + ibl.pca.PCA handed to Evaluator.evaluate -> pca.load(gpu) once,
    pca.infer per batch (ibl/evaluators.py:47-57).  Both flows against the oracle."""
    from ibl import models
    from ibl.evaluators import Evaluator, extract_features
    from ibl.pca import PCA
    from ibl.utils.data.sampler import DistributedSliceSampler
    emb_sd = synth.embednet_state(0)
    base = models.create("vgg16", pretrained=False)
    model = models.create("embednet", base, models.create("netvlad", dim=base.feature_dim))
    model.load_state_dict(emb_sd)
    model = model.to(dev).eval()
    nq, ng = 5, 13
    imgs = synth.images(nq + ng, 64, 96, seed=51)
    for i in range(nq):
        imgs[i] = imgs[nq + 2 * i] + 2.0 * torch.randn_like(imgs[i])
    query = [(f"q{i}.png", 1000 + i, 0.0, 0.0) for i in range(nq)]
    gallery = [(f"g{j}.png", j // 2, 0.0, 0.0) for j in range(ng)]
    gt = [[2 * i] for i in range(nq)]
    qset, gset = _Records(imgs[:nq], query), _Records(imgs[nq:], gallery)

    def loader(ds):
        return torch.utils.data.DataLoader(ds, batch_size=4, num_workers=0, shuffle=False,
                                           sampler=DistributedSliceSampler(ds), pin_memory=True)

    with torch.no_grad():
        vlad = od.extract_cnn_feature(imgs, emb_sd, vlad=True, with_pca=False)
    path = str(tmp_path / "pca_params_model_best.h5")
    pca = PCA(8, True, path)
    feats = extract_features(model, loader(gset), gallery, vlad=True, gpu=dev.index)
    assert_rel_l2("extract_features (EmbedNet, vlad)", torch.stack(list(feats.values())), vlad[nq:], 1e-4)
    pca.train(torch.stack(list(feats.values())))
    assert osp.isfile(path)
    want_desc = _oracle_pca(vlad, path, True)
    d = om.pairwise_distance(want_desc[:nq], want_desc[nq:]).numpy()
    want = om.evaluate_all(d, gt, [g[1] for g in gallery])
    ev = Evaluator(model)
    r_dev = ev.evaluate(loader(qset), query + gallery, query, gallery, gt, gallery_loader=loader(gset),
                        vlad=True, pca=pca, gpu=dev.index)
    r_host = ev.evaluate(loader(qset), query + gallery, query, gallery, gt, gallery_loader=loader(gset),
                         vlad=True, pca=pca, gpu=dev.index, device_resident=False)
    print("recalls with external PCA", r_dev, r_host, want)
    assert np.array_equal(r_dev, want) and np.array_equal(r_host, want)
    got = extract_features(model, loader(gset), gallery, vlad=True, pca=pca, gpu=dev.index)
    assert_rel_l2("extract_features (EmbedNet + PCA.infer)", torch.stack(list(got.values())),
                  want_desc[nq:], 1e-4)


def test_examples_test_py_main_worker_sequence(group, dev, tmp_path, capsys):
    """The statements of examples/test.py:77-133 (main_worker) and :29-72 (get_data / get_model), in
    order, against this `ibl` on a Pittsburgh-format dataset: DDP-wrapped EmbedNet, checkpoint
    through copy_state_dict, PCA trained because its parameter file is missing, evaluate — and a
    second run that finds the file and does not train again.  Recalls are checked against the
    oracle fed with the same images and the same PCA parameters."""
    from torch import nn
    from torch.utils.data import DataLoader
    from helpers import synthetic_pitts
    from ibl import datasets, models
    from ibl.evaluators import Evaluator, extract_features
    from ibl.pca import PCA
    from ibl.utils.data import get_transformer_test
    from ibl.utils.data.preprocessor import Preprocessor
    from ibl.utils.data.sampler import DistributedSliceSampler
    from ibl.utils.dist_utils import synchronize
    from ibl.utils.logging import Logger
    from ibl.utils.serialization import copy_state_dict, load_checkpoint, save_checkpoint
    import random
    import sys

    data_dir = str(tmp_path / "data")
    synthetic_pitts.make(osp.join(data_dir, "pitts"))
    emb_sd = synth.embednet_state(0)
    resume = str(tmp_path / "logs" / "model_best.pth.tar")
    save_checkpoint({"state_dict": {"module." + k: v for k, v in emb_sd.items()}, "epoch": 4,
                     "best_recall5": 0.5}, False, fpath=resume)
    args = argparse.Namespace(data_dir=data_dir, dataset="pitts", scale="30k", height=72, width=96,
                              test_batch_size=4, workers=0, arch="vgg16", vlad=True, reduction=True,
                              nowhiten=False, features=16, resume=resume, sync_gather=False,
                              rerank=False, rr_topk=25, lambda_value=0, gpu=dev.index, rank=0,
                              world_size=1)

    def run():
        # ---- get_data (test.py:29-56)
        root = osp.join(args.data_dir, args.dataset)
        dataset = datasets.create(args.dataset, root, scale=args.scale)
        t_db = get_transformer_test(args.height, args.width)
        t_q = get_transformer_test(args.height, args.width, tokyo=(args.dataset == "tokyo"))
        pitts = datasets.create("pitts", osp.join(args.data_dir, "pitts"), scale="30k", verbose=False)
        pitts_train = sorted(list(set(pitts.q_train) | set(pitts.db_train)))

        def mk(recs, tf, root_dir):
            return DataLoader(Preprocessor(recs, root=root_dir, transform=tf), batch_size=args.test_batch_size,
                              num_workers=args.workers, sampler=DistributedSliceSampler(recs),
                              shuffle=False, pin_memory=True)
        train_extract_loader = mk(pitts_train, t_db, pitts.images_dir)
        test_loader_q = mk(dataset.q_test, t_q, dataset.images_dir)
        test_loader_db = mk(dataset.db_test, t_db, dataset.images_dir)
        # ---- get_model (test.py:58-72)
        base_model = models.create(args.arch)
        pool_layer = models.create("netvlad", dim=base_model.feature_dim)
        model = models.create("embednet", base_model, pool_layer)
        model.cuda(args.gpu)
        model = nn.parallel.DistributedDataParallel(model, device_ids=[args.gpu], output_device=args.gpu,
                                                    find_unused_parameters=True)
        # ---- main_worker (test.py:86-131)
        log_dir = osp.dirname(args.resume)
        old_stdout = sys.stdout
        sys.stdout = Logger(osp.join(log_dir, "log_test_" + args.dataset + ".txt"))
        try:
            checkpoint = load_checkpoint(args.resume)
            copy_state_dict(checkpoint["state_dict"], model)
            evaluator = Evaluator(model)
            pca_parameters_path = osp.join(osp.dirname(args.resume),
                                           "pca_params_" + osp.basename(args.resume).split(".")[0] + ".h5")
            pca = PCA(args.features, (not args.nowhiten), pca_parameters_path)
            trained = False
            if not osp.isfile(pca_parameters_path):
                dict_f = extract_features(model, train_extract_loader, pitts_train, vlad=args.vlad,
                                          gpu=args.gpu, sync_gather=args.sync_gather)
                features = list(dict_f.values())
                if len(features) > 10000:
                    features = random.sample(features, 10000)
                features = torch.stack(features)
                pca.train(features)
                synchronize()
                trained = True
            recalls = evaluator.evaluate(
                test_loader_q, sorted(list(set(dataset.q_test) | set(dataset.db_test))), dataset.q_test,
                dataset.db_test, dataset.test_pos, gallery_loader=test_loader_db, vlad=args.vlad, pca=pca,
                rerank=args.rerank, gpu=args.gpu, sync_gather=args.sync_gather,
                nms=(True if args.dataset == "tokyo" else False), rr_topk=args.rr_topk,
                lambda_value=args.lambda_value)
            synchronize()
        finally:
            sys.stdout.close()
            sys.stdout = old_stdout
        return dataset, recalls, trained, pca_parameters_path, t_db

    dataset, recalls, trained, pca_path, tf = run()
    assert trained and osp.isfile(pca_path)
    log = open(osp.join(osp.dirname(resume), "log_test_pitts.txt")).read()
    assert "Recall Scores:" in log and "load PCA parameters" in log
    _, recalls2, trained2, _, _ = run()
    assert not trained2 and np.array_equal(recalls, recalls2)

    # the oracle on the same files: loader transform -> reference restatement -> same PCA file
    from PIL import Image

    def load(recs):
        return torch.stack([tf(Image.open(osp.join(dataset.images_dir, r[0])).convert("RGB")) for r in recs])
    with torch.no_grad():
        vq = od.extract_cnn_feature(load(dataset.q_test), emb_sd, vlad=True, with_pca=False)
        vg = od.extract_cnn_feature(load(dataset.db_test), emb_sd, vlad=True, with_pca=False)
    dq, dg = _oracle_pca(vq, pca_path, True), _oracle_pca(vg, pca_path, True)
    want = om.evaluate_all(om.pairwise_distance(dq, dg).numpy(), dataset.test_pos,
                           [g[1] for g in dataset.db_test])
    print("test.py sequence recalls", recalls, "oracle", want)
    assert np.array_equal(recalls, want)
    assert want[-1] == 1.0          # every query finds its noisy view within the top 10


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("f16mx", 1e-4), ("bf16x3", 1e-4), ("bf16", 6e-3)])
def test_configs1_batch32_480x640_against_oracle(dev, state_dict, precision, tol):
    """BASELINE.json configs[1] itself: 32 DISTINCT 480x640 images in one batch — the regime of the
    benchmark (tile counts, 32-bit buffer offsets, multiply-shift pixel decomposition) — against the
    oracle, eagerly and through the pipelined hipGraph replay bench.py times."""
    import hubconf
    x = synth.images(32, 480, 640, seed=321)
    assert (x[0] - x[1]).abs().max() > 1 and (x[7] - x[31]).abs().max() > 1
    if "b32" not in _CACHE:                  # ~16 s of host time, once for the three precisions
        with torch.no_grad():
            _CACHE["b32"] = torch.cat([od.embednetpca(x[i:i + 8], state_dict) for i in range(0, 32, 8)])
    want = _CACHE["b32"]
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(state_dict)
    model = model.to(dev).eval().set_precision(precision)
    xd = x.to(dev)
    got = model(xd).clone()
    assert_rel_l2(f"configs[1] batch 32 {precision}", got.cpu(), want, tol)
    worst = max(rel_l2(got[i].cpu(), want[i]) for i in range(32))
    worst_abs = float(((got.cpu().double() - want.double()).abs().amax(1) / want.double().abs().amax(1)).max())
    print(f"{precision}: worst single-image rel-L2 {worst:.3e}, worst max|diff| / max|want| {worst_abs:.3e}")
    assert worst < tol and worst_abs < tol          # north_star's bound holds for EVERY image, not on average
    if precision == "f16mx":
        assert model.base_model.effective_precision(xd) == "f16mx"     # the benchmark batch runs the MX kernels
    fwd = model.graphed(xd, pipeline=True)
    a, b = fwd(), fwd(xd)
    fwd.wait()
    torch.cuda.synchronize()
    assert torch.equal(a, got) and torch.equal(b, got)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
def test_replayed_extraction_equals_batch_by_batch(group, state_dict, dev, precision):
    """extract_features through the replayed two-lane route == the eager batch-by-batch route,
    bit for bit: mixed batch shapes (graph per repeated shape, eager for singletons), a ragged last
    batch, pinned and pageable batches, raw uint8 batches, external PCA, 16-bit storage."""
    import hubconf
    from openibl_amd import evaluators as ev
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(state_dict)
    model = model.to(dev).eval().set_precision(precision)
    g = torch.Generator().manual_seed(9)

    class Loader:
        """in-memory loader: a list of (images, names) batches"""
        def __init__(self, batches):
            self.batches = batches
            self.sampler = range(sum(int(b[0].shape[0]) for b in batches))

        def __iter__(self):
            return iter(self.batches)

        def __len__(self):
            return len(self.batches)

    shapes = [(3, 64, 96)] * 4 + [(2, 64, 96)] + [(1, 70, 90)] + [(3, 80, 64)] * 3 + [(3, 64, 96)] + [(1, 80, 64)]
    batches = []
    for k, (n, h, w) in enumerate(shapes):
        x = synth.images(n, h, w, seed=500 + k)
        if k % 2 == 0:
            x = x.pin_memory()
        batches.append((x, [f"im{k}_{i}" for i in range(n)]))
    names = [(f, 0, 0.0, 0.0) for b in batches for f in b[1]]
    for flag in (False, True):
        ev.FAST_EXTRACTION = flag
        try:
            feats = ev.extract_features(model, Loader(batches), names, gpu=dev.index)
        finally:
            ev.FAST_EXTRACTION = True
        m = torch.stack(list(feats.values()))
        if flag:
            assert torch.equal(m, want), "replayed extraction differs from the eager route"
        else:
            want = m
    assert list(feats.keys()) == [n[0] for n in names]
    # raw uint8 batches (ToTensor + Normalize inside the first kernel), 16-bit storage
    u8 = [(torch.randint(0, 256, (3, 64, 96, 3), generator=g, dtype=torch.uint8), None) for _ in range(4)]
    u8n = [(f"u{i}", 0, 0.0, 0.0) for i in range(12)]
    for flag in (False, True):
        ev.FAST_EXTRACTION = flag
        try:
            f8 = ev.extract_features(model, Loader(u8), u8n, gpu=dev.index, store_dtype=torch.float16)
        finally:
            ev.FAST_EXTRACTION = True
        m8 = torch.stack(list(f8.values()))
        if flag:
            assert torch.equal(m8, want8)
        else:
            want8 = m8


def test_captured_forwards_survive_between_extractions_and_die_with_their_state(group, state_dict, dev):
    """The graphs an extraction captured are kept on the model for the next one (same shapes: no eager batch,
    no capture), serve its FIRST batch too, and are dropped the moment anything they were captured for
    changes: a parameter written in place, the precision, the head's options."""
    import hubconf
    from openibl_amd import evaluators as ev
    from openibl_amd.extract import unwrap_model, _GRAPH_STORES, release_graphs
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(state_dict)
    model = model.to(dev).eval().set_precision("bf16x3")
    core = unwrap_model(model)

    class Loader:
        def __init__(self, batches):
            self.batches = batches
            self.sampler = range(sum(int(b[0].shape[0]) for b in batches))

        def __iter__(self):
            return iter(self.batches)

        def __len__(self):
            return len(self.batches)

    def run(seed, **kw):
        batches = [(synth.images(3, 64, 96, seed=seed + k), [f"s{seed}_{k}_{i}" for i in range(3)]) for k in range(4)]
        names = [(f, 0, 0.0, 0.0) for b in batches for f in b[1]]
        feats = ev.extract_features(model, Loader(batches), names, gpu=dev.index, **kw)
        eager = torch.cat([ev.extract_cnn_feature(model, b[0], gpu=dev.index).cpu() for b in batches])
        if kw.get("store_dtype") is not None:
            eager = eager.to(kw["store_dtype"])
        assert torch.equal(torch.stack(list(feats.values())), eager)

    run(700)
    store = _GRAPH_STORES[core]
    assert len(store[1]) == 1
    fwd = next(iter(store[1].values()))
    calls = fwd.calls
    run(800)                                            # other images, same shape: all four batches replayed
    assert _GRAPH_STORES[core] is store and next(iter(store[1].values())) is fwd
    assert fwd.calls == calls + 4
    run(900, store_dtype=torch.float16)                 # another head: another store
    assert _GRAPH_STORES[core] is not store
    store = _GRAPH_STORES[core]
    with torch.no_grad():
        core.net_vlad.centroids.mul_(1.0)               # written in place: the version counter moves
    run(1000, store_dtype=torch.float16)
    assert _GRAPH_STORES[core] is not store
    store = _GRAPH_STORES[core]
    model.set_precision("fp32")
    run(1100, store_dtype=torch.float16)
    assert _GRAPH_STORES[core] is not store
    # a precision ROUND TRIP with no extraction in between frees the packed weights the kept
    # graphs point into although every key component but the cache generation is back to its old value
    store = _GRAPH_STORES[core]
    model.set_precision("bf16")
    model.set_precision("fp32")
    run(1200, store_dtype=torch.float16)
    assert _GRAPH_STORES[core] is not store
    # ... and p.data.copy_() + invalidate() (the documented idiom) changes weights without a version bump
    store = _GRAPH_STORES[core]
    core.net_vlad.centroids.data.copy_(core.net_vlad.centroids.data * 1.5)
    model.invalidate()
    run(1300, store_dtype=torch.float16)                # (compares against the eager forward of the NEW weights)
    assert _GRAPH_STORES[core] is not store
    # the model stays copyable / picklable with captured forwards around, and they can be released
    import copy
    copy.deepcopy(core.net_vlad)
    release_graphs(model)
    assert core not in _GRAPH_STORES


def test_default_precision_is_the_fast_parity_mode(dev, state_dict, monkeypatch):
    """What a drop-in user gets (round 6): `torch.hub.load(<repo>, 'vgg16_netvlad')` with no
    environment variable and no set_precision() runs the backbone in f16mx — and its descriptor of the reference's
    480x640 vector is within north_star's 1e-4 of what the reference computed (tests/golden/desc_480x640.npz); fp32
    stays selectable.  (conftest.py pins the variable to 'fp32' for the rest of the suite.)"""
    from pathlib import Path
    from openibl_amd import models
    monkeypatch.delenv("OPENIBL_AMD_PRECISION", raising=False)
    assert models.default_precision() == "f16mx"
    repo = str(Path(__file__).resolve().parent.parent)
    model = torch.hub.load(repo, "vgg16_netvlad", source="local", pretrained=False)
    model.load_state_dict(state_dict)
    model = model.to(dev).eval()
    assert model.precision == "f16mx" and model.base_model.precision == "f16mx"
    g = load_golden("desc_480x640")
    x = synth.images(1, 480, 640, seed=int(g["image_seed"])).to(dev)
    desc = model(x)
    assert model.base_model.precision_runs == {"f16mx": 1} and model.base_model.range_fallbacks == 0
    assert rel_l2(desc.cpu(), g["desc"]) <= 1e-4
    print(f"default precision f16mx: desc rel-L2 vs the reference {rel_l2(desc.cpu(), g['desc']):.2e}")
    monkeypatch.setenv("OPENIBL_AMD_PRECISION", "fp32")
    assert models.default_precision() == "fp32" and model.precision == "fp32"
    assert rel_l2(model(x).cpu(), g["desc"]) <= 4e-6


def test_batch_beyond_the_32bit_offsets_runs_f16mx_in_image_groups(dev, state_dict):
    """128 images of 480x640: conv2_2's input is 5 GB, beyond the 32-bit buffer offsets of the f16mx kernels (95
    images) — rounds 1-5 ran such a batch in bf16x3 without saying so; it now runs f16mx in two groups of 64, each a
    pass of its own into its rows of the map (round 6): bit-equal to the two halves on their own."""
    import hubconf
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(state_dict)
    model = model.to(dev).eval().set_precision("f16mx")
    x = synth.images(128, 480, 640, seed=77).to(dev)
    assert model.base_model.effective_precision(x) == "f16mx"
    assert model.base_model.f16mx_groups(x) == [(0, 64), (64, 64)]
    desc = model(x)
    runs = dict(model.base_model.precision_runs)
    assert runs.get("f16mx(groups)") == 2 and "bf16x3" not in runs and model.base_model.range_fallbacks == 0
    # the conv5_3 map of the whole batch IS the two groups' maps (bit for bit: each group is a pass of its own); the
    # head of 128 rows sums in another order than the head of 64 (the PCA's split-K follows the row count): 1.1e-6 seen
    feat = model.base_model.features_nhwc(x)
    fa, fb = model.base_model.features_nhwc(x[:64].contiguous()), model.base_model.features_nhwc(x[64:].contiguous())
    assert torch.equal(feat[:64], fa) and torch.equal(feat[64:], fb)
    a, b = model(x[:64].contiguous()), model(x[64:].contiguous())
    assert rel_l2(desc[:64].cpu(), a.cpu()) <= 3e-6 and rel_l2(desc[64:].cpu(), b.cpu()) <= 3e-6
    want = od.embednetpca(x[125:128].cpu(), state_dict)
    assert rel_l2(desc[125:128].cpu(), want) <= 1e-4


def test_forward_settles_the_range_flag_behind_the_head(dev, state_dict):
    """`model(x)` in f16mx: the flag is read once, BEHIND the head's launches (round 6) — an ordinary batch and a batch
    that leaves the fp16 range both come out as before: the flagged one recomputed in bf16x3, head included."""
    import hubconf
    from test_gpu_range import _scaled, ACT_HEADROOM
    sd = _scaled(state_dict, 2000.0 * ACT_HEADROOM)
    model = hubconf.vgg16_netvlad(pretrained=False)
    model.load_state_dict(sd)
    model = model.to(dev).eval().set_precision("f16mx")
    model.base_model.F16MX_MIN_TILES = 0
    x = synth.images(2, 64, 96, seed=5)
    ok = model(x.to(dev))
    assert model.base_model.range_fallbacks == 0
    big = model((x * 2000.0 * ACT_HEADROOM).to(dev))
    assert model.base_model.range_fallbacks == 1
    model.set_precision("bf16x3")
    want_ok, want_big = model(x.to(dev)), model((x * 2000.0 * ACT_HEADROOM).to(dev))
    assert torch.equal(big, want_big)                       # the fallback IS the bf16x3 forward
    assert rel_l2(ok.cpu(), want_ok.cpu()) <= 1e-4
    # the other forwards share the helper: EmbedNet and the bare backbone
    from ibl import models
    emb = models.create("embednet", model.base_model, model.net_vlad).eval().set_precision("f16mx")
    emb.base_model.F16MX_MIN_TILES = 0
    p1, v1 = emb((x * 2000.0 * ACT_HEADROOM).to(dev))
    emb.set_precision("bf16x3")
    p2, v2 = emb((x * 2000.0 * ACT_HEADROOM).to(dev))
    assert torch.equal(v1, v2) and torch.equal(p1, p2)
