"""End-to-end Evaluator flows on one GPU with a synthetic loader (GPU)."""
import numpy as np
import pytest
import torch
import torch.distributed as dist

from openibl_amd import synth
from oracle import descriptor as od
from oracle import matching as om

pytestmark = pytest.mark.gpu


class _Records(torch.utils.data.Dataset):
    def __init__(self, images, records):
        self.images, self.records = images, records

    def __len__(self):
        return len(self.records)

    def __getitem__(self, i):
        f, pid, x, y = self.records[i]
        return self.images[i], f, pid, x, y


@pytest.fixture(scope="module")
def group():
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1)
    yield
    dist.destroy_process_group()


def test_evaluator_flows_agree_with_oracle(group, state_dict, dev):
    import hubconf
    from ibl.evaluators import Evaluator
    from ibl.utils.data.sampler import DistributedSliceSampler
    model = hubconf.vgg16_netvlad()
    model.load_state_dict(state_dict)
    model = model.to(dev).eval()
    nq, ng = 5, 13
    imgs = synth.images(nq + ng, 64, 96, seed=41)
    # queries are noisy copies of some gallery images
    for i in range(nq):
        imgs[i] = imgs[nq + 2 * i] + 2.0 * torch.randn_like(imgs[i])
    query = [(f"q{i}.png", 1000 + i, 0.0, 0.0) for i in range(nq)]
    gallery = [(f"g{j}.png", j // 2, 0.0, 0.0) for j in range(ng)]
    gt = [[2 * i] for i in range(nq)]
    qset, gset = _Records(imgs[:nq], query), _Records(imgs[nq:], gallery)

    def loader(ds):
        return torch.utils.data.DataLoader(ds, batch_size=4, num_workers=0, shuffle=False,
                                           sampler=DistributedSliceSampler(ds))

    ev = Evaluator(model)
    r_dev = ev.evaluate(loader(qset), query + gallery, query, gallery, gt, gallery_loader=loader(gset))
    r_host = ev.evaluate(loader(qset), query + gallery, query, gallery, gt,
                         gallery_loader=loader(gset), device_resident=False)
    r_nms = ev.evaluate(loader(qset), query + gallery, query, gallery, gt,
                        gallery_loader=loader(gset), nms=True)
    with torch.no_grad():
        desc = od.extract_cnn_feature(imgs, state_dict)
    d = om.pairwise_distance(desc[:nq], desc[nq:]).numpy()
    want = om.evaluate_all(d, gt, [g[1] for g in gallery])
    want_nms = om.evaluate_all(d, gt, [g[1] for g in gallery], nms=True)
    print("recalls", r_dev, r_host, want, "nms", r_nms, want_nms)
    assert np.array_equal(r_dev, want) and np.array_equal(r_host, want)
    assert np.array_equal(r_nms, want_nms)
    assert want[0] == 1.0     # the planted copies are found
    # rerank=True takes the reference's host flow: k-reciprocal re-ranking of the three matrices
    r_rr = ev.evaluate(loader(qset), query + gallery, query, gallery, gt, gallery_loader=loader(gset),
                       rerank=True, rr_topk=6, lambda_value=0.3)
    from openibl_amd.rerank import re_ranking
    d_rr = re_ranking(d, om.pairwise_distance(desc[:nq], desc[:nq]).numpy(),
                      om.pairwise_distance(desc[nq:], desc[nq:]).numpy(), k1=6, k2=1, lambda_value=0.3)
    want_rr = om.evaluate_all(d_rr, gt, [g[1] for g in gallery])
    print("re-ranked recalls", r_rr, want_rr)
    assert np.array_equal(r_rr, want_rr)


def test_evaluator_extensions_16bit_storage_and_multiscale(group, state_dict, dev):
    """BASELINE.json configs[4] extensions through the Evaluator: descriptors stored in 16 bits
    (device-resident and host flows agree with each other and with the oracle on the widened
    values), and multi-scale extraction against the oracle's definition."""
    import hubconf
    from ibl.evaluators import Evaluator
    from ibl.utils.data.sampler import DistributedSliceSampler
    model = hubconf.vgg16_netvlad()
    model.load_state_dict(state_dict)
    model = model.to(dev).eval()
    nq, ng = 4, 11
    imgs = synth.images(nq + ng, 64, 96, seed=43)
    for i in range(nq):
        imgs[i] = imgs[nq + 2 * i + 1] + 2.0 * torch.randn_like(imgs[i])
    query = [(f"q{i}.png", 1000 + i, 0.0, 0.0) for i in range(nq)]
    gallery = [(f"g{j}.png", j // 3, 0.0, 0.0) for j in range(ng)]
    gt = [[2 * i + 1] for i in range(nq)]
    qset, gset = _Records(imgs[:nq], query), _Records(imgs[nq:], gallery)

    def loader(ds):
        return torch.utils.data.DataLoader(ds, batch_size=3, num_workers=0, shuffle=False,
                                           sampler=DistributedSliceSampler(ds))

    pids = [g[1] for g in gallery]
    with torch.no_grad():
        desc = od.extract_cnn_feature(imgs, state_dict)
        desc_ms = od.multiscale_descriptor(imgs, state_dict, (1.0, 0.5))
    for dt in (torch.float16, torch.bfloat16):
        ev = Evaluator(model, descriptor_dtype=dt)
        r_dev = ev.evaluate(loader(qset), query + gallery, query, gallery, gt, gallery_loader=loader(gset))
        r_host = ev.evaluate(loader(qset), query + gallery, query, gallery, gt,
                             gallery_loader=loader(gset), device_resident=False)
        w = desc.to(dt).float()
        want = om.evaluate_all(om.pairwise_distance(w[:nq], w[nq:]).numpy(), gt, pids)
        print(dt, "recalls", r_dev, r_host, want)
        assert np.array_equal(r_dev, want) and np.array_equal(r_host, want)
    ev = Evaluator(model, scales=(1.0, 0.5))
    r_ms = ev.evaluate(loader(qset), query + gallery, query, gallery, gt, gallery_loader=loader(gset))
    want_ms = om.evaluate_all(om.pairwise_distance(desc_ms[:nq], desc_ms[nq:]).numpy(), gt, pids)
    print("multi-scale recalls", r_ms, want_ms)
    assert np.array_equal(r_ms, want_ms)


def test_device_resident_flow_refuses_a_loader_it_cannot_index(group, state_dict, dev):
    """The device-resident flow derives global gallery indices from DistributedSliceSampler's dealing;
    a loader that yields another number of items (drop_last here) must be refused, not mis-indexed —
    and the reference's host flow still takes it."""
    import hubconf
    from ibl.evaluators import Evaluator
    from ibl.utils.data.sampler import DistributedSliceSampler
    model = hubconf.vgg16_netvlad()
    model.load_state_dict(state_dict)
    model = model.to(dev).eval()
    nq, ng = 4, 10
    imgs = synth.images(nq + ng, 64, 96, seed=47)
    query = [(f"q{i}.png", 1000 + i, 0.0, 0.0) for i in range(nq)]
    gallery = [(f"g{j}.png", j, 0.0, 0.0) for j in range(ng)]
    gt = [[i] for i in range(nq)]
    qset, gset = _Records(imgs[:nq], query), _Records(imgs[nq:], gallery)
    ql = torch.utils.data.DataLoader(qset, batch_size=4, sampler=DistributedSliceSampler(qset))
    bad = torch.utils.data.DataLoader(gset, batch_size=4, sampler=DistributedSliceSampler(gset), drop_last=True)
    with pytest.raises(ValueError, match="DistributedSliceSampler"):
        Evaluator(model).evaluate(ql, query + gallery, query, gallery, gt, gallery_loader=bad)


@pytest.mark.parametrize("precision", ["f16mx", "bf16x3"])
def test_images_to_recall_at_480x640_in_the_headline_arithmetic(group, state_dict, dev, precision):
    """images -> Recall@1/5/10 through `Evaluator.evaluate` (ibl/evaluators.py:176-201) in the
    1e-4 arithmetics at the benchmark resolution: 74 planted 480x640 images (24 queries, 50 gallery), batches of
    32 with ragged last batches, DistributedSliceSampler, the device-resident flow (f16mx: top-k by the fp16
    filter + exact rescoring) and the reference's host flow (full matrix), with and without spatial NMS — all equal
    to the oracle's recalls on its own fp32 descriptors, and the gathered descriptors themselves within 1e-4."""
    import hubconf
    from ibl.evaluators import Evaluator, extract_features
    from ibl.utils.data.sampler import DistributedSliceSampler
    model = hubconf.vgg16_netvlad()
    model.load_state_dict(state_dict)
    model = model.to(dev).eval().set_precision(precision)
    nq, ng = 24, 50
    base = synth.images(ng // 2, 480, 640, seed=61)
    gen = torch.Generator().manual_seed(62)
    # gallery: two views per place (the second a noisy copy); queries: noisier copies of a view, half of them hard
    gal = torch.empty((ng, 3, 480, 640))
    for j in range(ng):
        gal[j] = base[j // 2] + (0.0 if j % 2 == 0 else 12.0) * torch.randn((3, 480, 640), generator=gen)
    qry = torch.empty((nq, 3, 480, 640))
    for i in range(nq):
        qry[i] = gal[(2 * i + 1) % ng] + (10.0 if i % 2 else 45.0) * torch.randn((3, 480, 640), generator=gen)
    imgs = torch.cat([qry, gal])
    query = [(f"q{i}.png", 1000 + i, 0.0, 0.0) for i in range(nq)]
    gallery = [(f"g{j}.png", j // 2, 0.0, 0.0) for j in range(ng)]
    gt = [[(2 * i + 1) % ng] for i in range(nq)]
    qset, gset = _Records(qry, query), _Records(gal, gallery)

    def loader(ds):
        return torch.utils.data.DataLoader(ds, batch_size=32, num_workers=0, shuffle=False,
                                           sampler=DistributedSliceSampler(ds))

    with torch.no_grad():
        desc = torch.cat([od.extract_cnn_feature(imgs[i:i + 8], state_dict) for i in range(0, nq + ng, 8)])
    pids = [g[1] for g in gallery]
    d = om.pairwise_distance(desc[:nq], desc[nq:]).numpy()
    want, want_nms = om.evaluate_all(d, gt, pids), om.evaluate_all(d, gt, pids, nms=True)
    ev = Evaluator(model)
    r_dev = ev.evaluate(loader(qset), query + gallery, query, gallery, gt, gallery_loader=loader(gset))
    r_host = ev.evaluate(loader(qset), query + gallery, query, gallery, gt, gallery_loader=loader(gset),
                         device_resident=False)
    r_nms = ev.evaluate(loader(qset), query + gallery, query, gallery, gt, gallery_loader=loader(gset), nms=True)
    print(precision, "recalls", r_dev, r_host, want, "nms", r_nms, want_nms)
    assert np.array_equal(r_dev, want) and np.array_equal(r_host, want) and np.array_equal(r_nms, want_nms)
    assert 0.3 < want[0] and want[2] <= 1.0
    feats = extract_features(model, loader(gset), gallery, gpu=dev.index)
    got = torch.stack([feats[g[0]] for g in gallery]).double()
    err = ((got - desc[nq:].double()).abs().amax(1) / desc[nq:].double().abs().amax(1)).max().item()
    print(f"{precision}: gathered gallery descriptors, worst image max|diff| / max|want| = {err:.2e}")
    assert err <= 1e-4
    assert model.base_model.range_fallbacks == 0
