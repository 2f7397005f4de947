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
