"""The replayed, two-lane forward behind `ibl.evaluators.extract_features`
(ibl/evaluators.py:36-103) and behind `EmbedNetPCA.graphed()`.

`GraphedForward` captures a forward of a fixed batch shape once into two hipGraphs — backbone (the
matrix-core launches) and head (everything after the conv5_3 map) — and replays them: a batch costs
the host two graph launches instead of ~30 kernel launches.  With `pipeline=True` there are two
LANES, each a HIP stream with its own input buffer, activation workspaces, feature map and output;
batch i runs wholly on lane i % 2:

    copy      H2D of batch i+1 into slot (i+1) % 2 as soon as that slot's last backbone has read it
    lane 0    backbone | head | hand-off of batch 0          backbone | head | hand-off of batch 2 ...
    lane 1              backbone | head | hand-off of batch 1          backbone | head | ...

Two hardware queues feed the chip: the head's latency / HBM-bound kernels of one lane hide behind
the matrix-core work of the other, and so do the partial last rounds of every convolution launch
(300 tiles on 256 CUs…) — measured +4 % (bf16x3) / +6.6 % (bf16) over one lane
(`tests/gpu_dual_stream.py`; a third lane adds nothing).  The PCIe copy has its own stream: with
the copy on the lanes the two lanes fall into step (both copy, then both compute) and the copy is
no longer hidden (measured: 4275 instead of 5500 images/s from fp32 host batches).

`extract_descriptors` is the loop of `extract_features` over one rank's loader built on it: a batch
shape is run eagerly the first time it is seen (which also packs weights and sizes workspaces) and
captured the second time — and kept on the model for later calls for as long as the state it was captured
in lasts (`_graph_store`); descriptors land in a pre-sized device matrix in loader order.  The
kernels, their order and their arithmetic are those of the eager path, so the descriptors are
bit-identical to `extract_cnn_feature` batch by batch (tested).
"""
from __future__ import annotations

import time
import weakref
from collections import OrderedDict
from typing import Callable, Optional

import torch

from . import ops

__all__ = ["GraphedForward", "EagerLanes", "extract_descriptors", "unwrap_model", "release_graphs", "MAX_CACHED_SHAPES"]

GUARD_REPLAYS = True      # diagnostic switch (tests/gpu_api_guard_ab.py): False = replayed forwards do not check the
                          # f16mx range flag (the eager path still does)
MAX_CACHED_SHAPES = 3     # captured (shape, dtype) entries kept per extraction (each owns its
                          # activation workspaces: ~2.5 GB for 32 x 480x640 in bf16)


def unwrap_model(model):
    """The module behind DistributedDataParallel / DataParallel wrappers (examples/test.py:67-69)."""
    while hasattr(model, "module") and isinstance(getattr(model, "module"), torch.nn.Module):
        model = model.module
    return model


_LANE_POOL = {}
NUM_LANES = 2             # lanes of a pipelined replay.  Three are slower: f16mx 3570 / 3697 / 3665 images/s on
                          # 1 / 2 / 3 lanes, bf16 6261 / 6812 / 6612 (tests/gpu_lanes_ab.py, profiles/r06_n_lanes_ab.txt)


def _lane_streams(dev: torch.device):
    """The two lane streams and the copy stream of a device, created ONCE per process.  HIP maps
    streams onto a few hardware queues; two streams created at different times can land on the same
    queue, and lanes that share a queue run one after the other — the back-fill is gone without any
    error (measured: a second pair of fresh streams fell back to the one-lane rate, a pool created
    once held the two-lane rate in every run, tests/gpu_dual_stream.py).  All GraphedForward objects
    of a device therefore share one pair of lanes (they are replayed one at a time anyway)."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _LANE_POOL:
        _LANE_POOL[key] = ([torch.cuda.Stream(device=dev) for _ in range(max(2, NUM_LANES))],
                           torch.cuda.Stream(device=dev))
    return _LANE_POOL[key]


class GraphedForward:
    """Two-graph replay of `head_fn(backbone_fn(x))` for one input shape.

        fwd = GraphedForward(backbone_fn, head_fn, example, pipeline=True)
        out = fwd(x)              # x: device tensor or (pinned) host tensor of the example's shape
        out = fwd()               # again on the batch resident in the slot
        fwd.wait()                # the current stream waits for everything launched so far

    backbone_fn(x) -> feature map and head_fn(feat) -> output tensor must only launch work on the
    current stream, allocate through torch (graph-pool allocations) and take their scratch from
    ops.workspace (keyed by stream: every lane captures on its own stream and so owns its scratch).
    The returned tensor is the slot's static output: complete after `wait()` (or a device
    synchronisation), overwritten by the call `depth` calls later.  `dest=` hands the result off
    instead: the rows are copied into `dest` on the lane right behind the head, and the slot is free
    again without the caller synchronising.  A device tensor `x` must stay unchanged until the lane
    has copied it (`wait()`); host tensors are read before the call returns unless pinned.  A host batch
    crosses PCIe into one of two device staging buffers of its slot (its transfer may run while the slot's
    previous batch is still being computed) and reaches the graph's input with a device copy on the lane.
    `events`: optional (start, stop) torch.cuda.Event pair recorded on the launching lane right
    around the backbone graph (bench.py's matrix-core span: meaningful with ONE lane — with two the
    spans of consecutive batches overlap).

    f16mx range guard.  A backbone that runs in f16mx raises a device flag when an activation is beyond
    fp16 (models.VGG.features_nhwc); a replayed graph cannot branch on it, so the flag travels to a pinned
    host word behind every backbone replay (one word per call, a ring of RING) and every call first POLLS the
    batches in flight: one whose `done` event has fired has its flag read, and a flagged batch is recomputed
    eagerly in bf16x3 — from the caller's tensor `x`, which the queue keeps referenced (a pinned loader batch, a
    device tensor), into its `dest` (and into the slot's output while the slot still holds that batch).  The
    host only BLOCKS on a batch when its slot is needed again and the batch has no other source than the
    slot's input (`x=None`, or `stable_src=False`: a staging buffer the caller reuses), when RING - 2 batches
    are unchecked, in `wait()`, and — one lane — before the call returns.  So the input copies of the next
    batches are enqueued as far ahead as without the guard.  `range_guard`: the object with
    `last_range_flag()` / `features_fallback(x)` (default: the object `backbone_fn` is bound to, if it has them).
    `keep`: optional callable returning tensors (in nested dicts / lists) the captured kernels point into —
    packed weights — evaluated after the capture; they stay referenced for the life of this object."""

    def __init__(self, backbone_fn: Callable, head_fn: Callable, example: torch.Tensor,
                 pipeline: bool = False, range_guard=None, keep: Optional[Callable] = None):
        if not example.is_cuda or example.dim() != 4 or example.dtype not in (torch.float32, torch.uint8):
            raise ValueError("graphed forward: example must be a CUDA tensor, float32 [N][3][H][W] "
                             "or uint8 [N][H][W][3]")
        dev = example.device
        self.device = dev
        self.pipeline = bool(pipeline)
        self.depth = NUM_LANES if self.pipeline else 1
        self.calls = 0
        self.lanes, self.copy = _lane_streams(dev) if self.pipeline else ([None], None)
        self.in_ready = [torch.cuda.Event() for _ in range(self.depth)]
        self.bb_done = [torch.cuda.Event() for _ in range(self.depth)]
        self.last_stream = None               # the stream the last call's input copy ran on
        # never aliases a caller's tensor; every slot starts with the example in place
        self.static_in = [example.clone(memory_format=torch.contiguous_format) for _ in range(self.depth)]
        self.done = [torch.cuda.Event() for _ in range(self.depth)]
        self.stage_in = {}                    # (slot, half) -> (device staging buffer for host batches, its free event)
        self.g_backbone, self.g_head, self.out = [], [], []
        self._keep = []
        if range_guard is None and GUARD_REPLAYS:
            owner = getattr(backbone_fn, "__self__", None)
            if hasattr(owner, "last_range_flag") and hasattr(owner, "features_fallback"):
                range_guard = owner
        self.guard = range_guard
        self.head_fn = head_fn
        self.flag_dev = [None] * self.depth       # the lane's range flag (device), None: not an f16mx capture
        self.flag_ring = None                     # pinned host words: call c's flag lands in word c % RING
        self.ev_ring = []                         # call c's completion event (flag copy + hand-off behind it)
        self.pending = []                         # calls whose flag has not been checked: oldest first
        self.slot_call = [-1] * self.depth        # the last call that ran on each slot
        self.range_fallbacks = 0
        with torch.no_grad():
            head_fn(backbone_fn(self.static_in[0]))   # packs weights, sizes every workspace, warms up
            torch.cuda.synchronize(dev)               # (also: the shared lanes are idle before a capture)
            for j in range(self.depth):
                # Every graph captures into its OWN memory pool (a temporary that head_fn frees during
                # its capture must never be handed to another lane's capture), and every lane captures
                # ON ITS OWN STREAM: ops.workspace keys scratch buffers by stream, so the two lanes —
                # which run concurrently — record disjoint activation / partial-sum workspaces.
                kw = {"stream": self.lanes[j]} if self.pipeline else {}
                gb = torch.cuda.CUDAGraph()
                # thread_local: a communicator's watchdog thread (RCCL, one process per GPU) may
                # touch the HIP runtime while this thread captures
                with torch.cuda.graph(gb, capture_error_mode="thread_local", **kw):
                    feat = backbone_fn(self.static_in[j])
                flag = self.guard.last_range_flag() if self.guard is not None else None
                if flag is not None:
                    self.flag_dev[j] = flag
                    self._keep.append(flag)
                    if self.flag_ring is None:
                        self.flag_ring = torch.zeros(self.RING, dtype=torch.int32).pin_memory()
                        self.ev_ring = [torch.cuda.Event() for _ in range(self.RING)]
                gh = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gh, capture_error_mode="thread_local", **kw):
                    out = head_fn(feat)
                self.g_backbone.append(gb)
                self.g_head.append(gh)
                self.out.append(out)
                self._keep += [feat, out]
            self._keep += ops.workspaces_snapshot()   # scratch the graphs recorded pointers into
            if keep is not None:
                self._keep += _tensors_in(keep())     # packed weights the graphs recorded pointers into
            torch.cuda.synchronize(dev)

    RING = 16     # range-guard bookkeeping: calls that may be in flight with their flag unchecked

    def _settle(self, entry, block: bool) -> bool:
        """Check the range flag of one call in flight (block: wait for it); recompute its batch in bf16x3 if
        the flag is set.  False = not finished yet (and not blocking)."""
        c, j, src, dest = entry
        ev = self.ev_ring[c % self.RING]
        if block:
            ev.synchronize()
        elif not ev.query():
            return False
        if int(self.flag_ring[c % self.RING]) == 0:
            return True
        self.range_fallbacks += 1
        lane = self.lanes[j] if self.pipeline else torch.cuda.current_stream(self.device)
        with torch.no_grad(), torch.cuda.stream(lane):
            x = self.static_in[j] if src is None else src.to(self.device, non_blocking=True)
            out = self.head_fn(self.guard.features_fallback(x))
            if self.slot_call[j] == c:            # the slot still holds this batch: its static output too
                self.out[j].copy_(out)
            if dest is not None:
                dest.copy_(out)
            self.done[j].record(lane)
        return True

    def _poll(self, need_slot: Optional[int] = None) -> None:
        """Settle what has finished; block on the calls that must be settled before slot `need_slot` is
        overwritten (their only source is the slot's input) and on the oldest ones when the ring is full."""
        keep = []
        for n, entry in enumerate(self.pending):
            c, j, src, dest = entry
            must = (need_slot is not None and j == need_slot and src is None) or \
                   len(self.pending) - n > self.RING - 2
            if not self._settle(entry, must):
                keep.append(entry)
        self.pending = keep

    @property
    def shape(self):
        return self.static_in[0].shape

    def _stage_in(self, j: int, h: int):
        """(device staging buffer, event of the device copy that last read it) for host batches of slot j."""
        key = (j, h)
        if key not in self.stage_in:
            self.stage_in[key] = (torch.empty_like(self.static_in[j]), torch.cuda.Event())
            # the block may have had an earlier life on the caller's stream
            self.copy.wait_stream(torch.cuda.current_stream(self.device))
        return self.stage_in[key]

    def __call__(self, x: Optional[torch.Tensor] = None, events=None,
                 dest: Optional[torch.Tensor] = None, stable_src: bool = False) -> torch.Tensor:
        """stable_src=False (default): `x` only has to stay unchanged until the lane has copied it (the
        contract above) — with the range guard the batch is then settled from the SLOT's copy before the slot
        is used again.  stable_src=True: the caller promises that `x` stays alive and unchanged until the
        batch is settled (up to RING - 2 calls later; `extract_descriptors` says so for the loader batches it
        owns): the queue keeps `x` referenced and re-reads it if the f16mx range flag fires, and the host never
        blocks on a slot."""
        j = self.calls % self.depth
        c = self.calls
        self.calls += 1
        guarded = self.flag_dev[j] is not None
        if guarded:
            self._poll(need_slot=j)
        main = torch.cuda.current_stream(self.device)
        if x is not None and (x.shape != self.static_in[j].shape or x.dtype != self.static_in[j].dtype):
            raise ValueError(f"graphed forward was captured for {tuple(self.static_in[j].shape)} "
                             f"{self.static_in[j].dtype}")
        if not self.pipeline:
            self.last_stream = main
            if x is not None:                 # x=None: run again on the batch resident in the slot
                self.static_in[j].copy_(x, non_blocking=True)
            if events is not None:
                events[0].record()
            self.g_backbone[j].replay()
            if events is not None:
                events[1].record()
            if guarded:
                ops.flag_to_host(self.flag_dev[j], self.flag_ring, c % self.RING)
            self.g_head[j].replay()
            if dest is not None:
                dest.copy_(self.out[j], non_blocking=True)
            self.slot_call[j] = c
            if guarded:                           # one lane: the result is final when the call returns
                self.ev_ring[c % self.RING].record(main)
                self._settle((c, j, None, dest), True)
            return self.out[j]
        lane = self.lanes[j]
        if x is not None and x.is_cuda:
            # the slot's last backbone has read its input; whatever produced x on the caller's
            # stream has finished
            self.copy.wait_event(self.bb_done[j])
            self.copy.wait_stream(main)
            with torch.cuda.stream(self.copy):
                self.static_in[j].copy_(x, non_blocking=True)
                self.in_ready[j].record(self.copy)
            # the caller may drop x right after this call: its block must not be handed out again
            # (on the caller's stream) before the copy stream has read it
            x.record_stream(self.copy)
            lane.wait_event(self.in_ready[j])
            self.last_stream = self.copy
        elif x is not None:
            # A host batch crosses PCIe into a STAGING buffer of the slot (two per slot, call c in buffer
            # (c / depth) % 2) and moves into the graph's input with a device copy on the lane.  Copying straight
            # into the slot's input had to wait for the slot's previous backbone (it reads that input), i.e. the
            # transfer of batch c + 2 started when batch c was half done and the lane idled until it arrived: with
            # 118 MB fp32 batches extract_features ran 10 % below the eager lanes, whose inputs are fresh tensors
            # (profiles/r06_n_eager_lanes.txt).  Now the transfer waits only for the device copy that last read its
            # staging buffer (four calls back) and runs as far ahead as the host does.
            buf, free = self._stage_in(j, (c // self.depth) % 2)
            self.copy.wait_event(free)
            with torch.cuda.stream(self.copy):
                buf.copy_(x, non_blocking=True)
                self.in_ready[j].record(self.copy)
            lane.wait_event(self.in_ready[j])
            with torch.cuda.stream(lane):          # behind the slot's previous batch on the same lane
                self.static_in[j].copy_(buf, non_blocking=True)
                free.record(lane)
            self.last_stream = self.copy
        if dest is not None:
            lane.wait_stream(main)           # dest was allocated / last written on the caller's stream
        # the slot's previous batch (feature map, output, hand-off) is behind us on the same lane
        with torch.cuda.stream(lane):
            if events is not None:
                events[0].record(lane)
            self.g_backbone[j].replay()
            self.bb_done[j].record(lane)
            if events is not None:
                events[1].record(lane)
            self.g_head[j].replay()
            if dest is not None:
                dest.copy_(self.out[j], non_blocking=True)
            self.done[j].record(lane)
            if guarded:
                # behind the head and the hand-off (the flag stays valid until this lane's next backbone clears
                # it): nothing sits between the two graph launches of a batch
                ops.flag_to_host(self.flag_dev[j], self.flag_ring, c % self.RING)
                self.ev_ring[c % self.RING].record(lane)
        self.slot_call[j] = c
        if guarded:
            self.pending.append((c, j, x if (x is not None and stable_src) else None, dest))
        return self.out[j]

    def wait(self) -> None:
        """Make the current stream wait for every batch (and hand-off copy) launched so far.  With the f16mx
        range guard this first checks the flags of the batches in flight (the host waits for them)."""
        for entry in self.pending:
            self._settle(entry, True)
        self.pending = []
        if self.pipeline:
            main = torch.cuda.current_stream(self.device)
            for ev in self.done:
                main.wait_event(ev)


class EagerLanes:
    """GraphedForward's two-lane schedule with EAGER launches: any batch shape, nothing captured.

        lanes = EagerLanes(core.base_model, head_fn, dev)
        lanes(x, dest)            # x: device or (pinned) host batch; the result lands in `dest` on the lane
        lanes.wait()              # the current stream waits for everything launched; flags settled

    Batch i runs wholly on lane i % 2 (its own stream and therefore its own ops.workspace scratch), its input copy on
    the copy stream — the schedule of the class above, ~30 kernel launches per batch instead of two graph launches
    (150 us of host time against a 9 ms batch).  The f16mx range flag is NEVER waited for inside the loop (round 6,
    VERDICT r05 item 7: the eager path used to end every batch in `flag.item()`): behind a batch's head the flag word
    travels to a pinned ring (`ops.flag_to_host`), the batch's input stays referenced, and finished batches are
    polled at the next call — a flagged one is recomputed in bf16x3 on its lane into its `dest`.  The host blocks only
    when RING - 2 batches are unchecked and in `wait()`."""

    RING = 16

    def __init__(self, base, head_fn, dev: torch.device):
        self.base, self.head_fn, self.device = base, head_fn, dev
        self.lanes, self.copy = _lane_streams(dev)
        self.calls = 0
        self.flag_ring = torch.zeros(self.RING, dtype=torch.int32).pin_memory()
        self.ev_ring = [torch.cuda.Event() for _ in range(self.RING)]
        self.pending = []                  # (call, lane index, device input, dest) with the flag unchecked
        self.done = [torch.cuda.Event() for _ in range(2)]
        self.used = [False, False]
        self.range_fallbacks = 0
        self.last_stream = None
        self.warm = set()                  # batch shapes whose first pass (weight packing, workspace growth) is behind us

    def _settle(self, entry, block: bool) -> bool:
        c, j, xd, dest = entry
        ev = self.ev_ring[c % self.RING]
        if block:
            ev.synchronize()
        elif not ev.query():
            return False
        if int(self.flag_ring[c % self.RING]) == 0:
            return True
        self.range_fallbacks += 1
        self.base.range_fallbacks += 1
        with torch.no_grad(), torch.cuda.stream(self.lanes[j]):
            dest.copy_(self.head_fn(self.base.features_fallback(xd)))
            self.done[j].record(self.lanes[j])
        return True

    def _poll(self) -> None:
        keep = []
        for n, entry in enumerate(self.pending):
            if not self._settle(entry, len(self.pending) - n > self.RING - 2):
                keep.append(entry)
        self.pending = keep

    def __call__(self, x: torch.Tensor, dest: torch.Tensor) -> None:
        c, j = self.calls, self.calls % 2
        self.calls += 1
        self._poll()
        main = torch.cuda.current_stream(self.device)
        lane = self.lanes[j]
        if x.is_cuda:
            xd = x
            lane.wait_stream(main)
            x.record_stream(lane)
            self.last_stream = lane
        else:
            with torch.cuda.stream(self.copy):
                xd = x.to(self.device, non_blocking=True)
                ready = torch.cuda.Event()
                ready.record(self.copy)
            lane.wait_event(ready)
            xd.record_stream(lane)
            self.last_stream = self.copy
        lane.wait_stream(main)                 # dest was allocated / last written on the caller's stream
        with torch.no_grad(), torch.cuda.stream(lane):
            feat = self.base.features_nhwc(xd, defer_flag=True)
            flag = self.base.last_range_flag()
            out = self.head_fn(feat)
            dest.copy_(out, non_blocking=True)
            self.done[j].record(lane)
            self.used[j] = True
            if flag is not None:
                ops.flag_to_host(flag, self.flag_ring, c % self.RING)
                self.ev_ring[c % self.RING].record(lane)
                self.pending.append((c, j, xd, dest))
        key = (tuple(x.shape), x.dtype)
        if key not in self.warm:
            # the first pass of a shape may pack weights (another effective precision, the PCA's streamed copy) on
            # THIS lane: the other lane must not read them before they are written
            self.warm.add(key)
            self.lanes[1 - j].wait_event(self.done[j])

    def wait(self) -> None:
        for entry in self.pending:
            self._settle(entry, True)
        self.pending = []
        main = torch.cuda.current_stream(self.device)
        for j in range(2):
            if self.used[j]:
                main.wait_event(self.done[j])


def _tensors_in(obj, out=None):
    """Every tensor reachable through nested dicts / lists / tuples."""
    out = [] if out is None else out
    if torch.is_tensor(obj):
        out.append(obj)
    elif isinstance(obj, dict):
        for v in obj.values():
            _tensors_in(v, out)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _tensors_in(v, out)
    return out


# ---------------------------------------------------------------------------------------------
def _head_fn(core, vlad: bool, pca, store_dtype):
    """What extract_cnn_feature (+ PCA.infer, + descriptor storage) computes behind the conv5_3 map
    (ibl/evaluators.py:22-34, 55-57), as a function of the NHWC feature map.  The closure holds the model
    WEAKLY: a GraphedForward keeps its head function, and the graph store is weakly keyed on the model — a
    strong reference from the value to its own key would keep captured graphs (2.5-5 GB per shape) alive
    after the model is dropped."""
    from .models import EmbedNetPCA
    core_ref = weakref.ref(core)

    def head(feat):
        core = core_ref()
        if core is None:
            raise RuntimeError("openibl_amd: the model behind this captured forward has been dropped")
        if isinstance(core, EmbedNetPCA):
            out = core.head_from_features(feat)
        elif vlad:
            _, out = core.net_vlad.aggregate_nhwc(feat, want_raw=False, want_norm=True)
        else:
            out = ops.global_maxpool_nhwc(feat)
        out = ops.l2_normalize(out)
        if pca is not None:
            out = pca.infer(out)
        return ops.store_descriptors(out, store_dtype)

    return head


def _guarded_backbone(base, x):
    """features_nhwc with the range flag settled at once (the first batch of an extraction: its output tells the
    descriptor width, so the host waits for it anyway)."""
    return base.features_nhwc(x)


def fast_path_supported(model) -> bool:
    from .models import EmbedNet, EmbedNetPCA, EmbedRegionNet
    return isinstance(unwrap_model(model), (EmbedNet, EmbedNetPCA, EmbedRegionNet))


class _PinnedStage:
    """Two pinned host buffers for batches the loader did not pin (pin_memory=False): the batch is
    copied into one of them on the host, then travels asynchronously like a pinned batch."""

    def __init__(self):
        self.bufs = [None, None]
        self.done = [None, None]
        self.i = 0

    def stage(self, x: torch.Tensor, dev):
        """-> (pinned view holding x, buffer index for mark())"""
        j = self.i
        self.i ^= 1
        if self.done[j] is not None:
            self.done[j].synchronize()        # the H2D that read this buffer last has completed
        b = self.bufs[j]
        if b is None or b.numel() < x.numel() or b.dtype != x.dtype:
            b = torch.empty(x.numel(), dtype=x.dtype).pin_memory()
            self.bufs[j] = b
        v = b[: x.numel()].view(x.shape)
        v.copy_(x)
        return v, j

    def mark(self, j: int, stream) -> None:
        ev = torch.cuda.Event()
        ev.record(stream)
        self.done[j] = ev


# model -> (state key, OrderedDict of captured forwards).  Outside the modules (a module holding CUDAGraph /
# Stream / Event objects could no longer be deep-copied or pickled), weakly keyed: the graphs go with the model.
_GRAPH_STORES: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def release_graphs(model=None) -> None:
    """Drop the captured forwards kept for `model` between extract_descriptors calls (all models: None).
    Each captured batch shape owns static inputs, two graph memory pools and its activation workspaces —
    2.5-5 GB for 32 x 480 x 640 — until it is evicted (MAX_CACHED_SHAPES), the model's state changes or
    this is called."""
    if model is None:
        _GRAPH_STORES.clear()
    else:
        _GRAPH_STORES.pop(unwrap_model(model), None)


def _graph_store(core, pca, vlad, store_dtype, dev) -> "OrderedDict":
    """The captured forwards of `core`, kept between extract_descriptors calls (a capture is a device sync,
    an eager warm-up and four graph captures: ~30 ms, 5 % of a 48-batch extraction).  They are valid for
    exactly the state they were captured in — every parameter / buffer (storage and version counter), the
    precision AND THE CACHE GENERATION of every module (`set_precision()`, `invalidate()` and
    `load_state_dict` bump it: they free the packed weights a graph points into, and `p.data.copy_()` +
    `invalidate()` changes weights without a version bump), the small-batch threshold, the PCA object and
    its parameters, the head's options, the device; anything else empties the store.  Every GraphedForward
    additionally keeps the packed tensors it captured referenced (`keep=`)."""
    def tensors_of(obj):
        return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in obj if torch.is_tensor(t))
    state = (
        tensors_of(list(core.parameters()) + list(core.buffers())),
        tuple((getattr(m, "precision", None), getattr(m, "_cache_gen", 0)) for m in core.modules()),
        getattr(getattr(core, "base_model", None), "F16MX_MIN_TILES", None),
        None if pca is None else (id(pca), getattr(pca, "precision", None),
                                  tensors_of([getattr(pca, "weight", None), getattr(pca, "bias", None)])),
        bool(vlad), str(store_dtype), str(dev),
    )
    store = _GRAPH_STORES.get(core)
    if store is None or store[0] != state:
        store = (state, OrderedDict())
        _GRAPH_STORES[core] = store
    return store[1]


def _module_caches(core, pca=None):
    """The packed / cast parameter copies of a model's modules (what captured kernels point into)."""
    def collect():
        return [getattr(m, "_cache", None) for m in core.modules()] + [getattr(pca, "_cache", None)]
    return collect


def extract_descriptors(model, data_loader, vlad=True, pca=None, gpu=None, print_freq=10, rank=0,
                        store_dtype=None, use_graphs: bool = True) -> torch.Tensor:
    """Descriptors of every item one rank's loader yields, in loader order, as one device matrix
    [n_local][d] (float32, or `store_dtype`).  The caller has put the model in eval mode and loaded
    `pca`.  Batches may be float32 [N][3][H][W] (normalised) or uint8 [N][H][W][3] (raw), pinned or
    not, of any mix of shapes."""
    core = unwrap_model(model)
    dev = torch.device("cuda", torch.cuda.current_device() if gpu is None else gpu)
    backbone = core.base_model.features_nhwc
    head = _head_fn(core, vlad, pca, store_dtype)
    try:
        n_items = len(data_loader.sampler)
    except Exception:
        n_items = None
    try:
        n_batches = len(data_loader)
    except Exception:
        n_batches = -1
    final, chunks, row = None, [], 0
    graphs = _graph_store(core, pca, vlad, store_dtype, dev) if use_graphs else OrderedDict()
    seen, evicted = {}, set()
    stage = _PinnedStage()
    main = torch.cuda.current_stream(dev)
    last_fwd = eager = None
    t_end = time.time()
    t_data = t_batch = 0.0
    with torch.no_grad():
        for i, batch in enumerate(data_loader):
            imgs = batch[0]
            if not torch.is_tensor(imgs):
                imgs = torch.as_tensor(imgs)
            if imgs.dtype != torch.uint8 and imgs.dtype != torch.float32:
                imgs = imgs.float()
            imgs = imgs.contiguous()
            t_data = time.time() - t_end
            slot = None
            if not imgs.is_cuda and not imgs.is_pinned():
                imgs, slot = stage.stage(imgs, dev)
            n = int(imgs.shape[0])
            key = (tuple(imgs.shape), imgs.dtype)
            seen[key] = seen.get(key, 0) + 1
            fwd = graphs.get(key)
            if fwd is None and use_graphs and seen[key] >= 2 and key not in evicted:
                if last_fwd is not None:
                    last_fwd.wait()
                main.synchronize()            # capture starts from an idle device
                ex = imgs.to(dev, non_blocking=False)
                fwd = GraphedForward(backbone, head, ex, pipeline=True, keep=_module_caches(core, pca))
                graphs[key] = fwd
                while len(graphs) > MAX_CACHED_SHAPES:
                    # a loader cycling through more shapes than are kept would re-capture (a device
                    # sync, an eager warm-up and four captures) for single batches: an evicted shape
                    # stays eager for the rest of this call
                    evicted.add(graphs.popitem(last=False)[0])
            if fwd is not None:
                graphs.move_to_end(key)
            if final is None and fwd is not None and n_items is not None:
                # a forward captured by an earlier call serves the first batch too: its output tells d
                final = torch.empty((n_items, int(fwd.out[0].shape[1])), dtype=fwd.out[0].dtype, device=dev)
            if final is None and fwd is None:
                # first batch: eager (packs the weights, sizes the workspaces, tells d and the dtype)
                out = head(_guarded_backbone(core.base_model, imgs.to(dev, non_blocking=True)))
                if slot is not None:
                    stage.mark(slot, main)
                d = int(out.shape[1])
                if n_items is not None:
                    final = torch.empty((n_items, d), dtype=out.dtype, device=dev)
                    final[row:row + n].copy_(out)
                else:
                    chunks.append(out)
            else:
                dst = None
                if final is not None:
                    if row + n > final.shape[0]:       # the loader yields more than its sampler said
                        # lane hand-offs into the old matrix may still be in flight (and the old block
                        # goes back to the allocator of THIS stream): wait for every lane first
                        if last_fwd is not None:
                            last_fwd.wait()
                        final = torch.cat([final, torch.empty((row + n - final.shape[0], final.shape[1]),
                                                              dtype=final.dtype, device=dev)])
                    dst = final[row:row + n]
                if fwd is not None:
                    if last_fwd is not None and last_fwd is not fwd:
                        last_fwd.wait()               # another shape's lanes: keep the order simple
                    out = fwd(imgs, dest=dst, stable_src=slot is None)   # (a staged batch's buffer is reused)
                    if slot is not None:
                        stage.mark(slot, fwd.last_stream)
                    last_fwd = fwd
                    if dst is None:
                        fwd.wait()
                        chunks.append(out.clone())
                elif dst is not None:
                    # an uncaptured shape (first sight, use_graphs=False, evicted): the same two lanes, eager
                    if last_fwd is not None and last_fwd is not eager:
                        last_fwd.wait()
                    if eager is None:
                        eager = EagerLanes(core.base_model, head, dev)
                    eager(imgs, dst)
                    if slot is not None:
                        stage.mark(slot, eager.last_stream)
                    last_fwd = eager
                else:
                    if last_fwd is not None:
                        last_fwd.wait()
                    out = head(_guarded_backbone(core.base_model, imgs.to(dev, non_blocking=True)))
                    if slot is not None:
                        stage.mark(slot, main)
                    chunks.append(out)
            row += n
            t_batch = time.time() - t_end
            t_end = time.time()
            if (i + 1) % print_freq == 0 and rank == 0:
                print("Extract Features: [{}/{}]\tTime {:.3f}\tData {:.3f}\t".format(
                    i + 1, n_batches, t_batch, t_data))
        if last_fwd is not None:
            last_fwd.wait()
    if final is not None:
        return final[:row]
    if not chunks:
        return torch.empty((0, 0), device=dev)
    return torch.cat(chunks)
