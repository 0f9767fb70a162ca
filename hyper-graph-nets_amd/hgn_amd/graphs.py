"""HIP-graph capture of the launch-bound regimes.

A 15-layer forward is ~130 kernel launches, a training step ~560; on one small graph (the reference's rollout,
``FlagModel.rollout/_step_fn`` src/model/flag.py:193-246: batch size 1, strictly sequential) the GPU finishes each kernel
in a few microseconds and the step is bound by launch latency.  Mesh topology is constant along a trajectory (cells are
static), so the whole forward -- or the whole training step -- is captured ONCE into a HIP graph over static buffers and
replayed with new features copied in: one hipGraphLaunch per step instead of hundreds of launches.

Capture goes through torch.cuda.CUDAGraph (on ROCm this is hipGraph): the kernels of libhgn_mp.so are launched on torch's
current stream, which during capture is the capturing stream, so they become graph nodes like any other launch.  The
topology cache (CSR build, which synchronises) must be warm before capture -- the warm-up calls below do that.
"""
from typing import Dict, Optional, Sequence

import torch

from . import ops
from .util import EdgeSet, MultiGraph


def _static_copy(graph: MultiGraph) -> MultiGraph:
    nodes = [x.detach().clone() for x in graph.node_features]
    sets = [EdgeSet(e.name, e.features.detach().clone(), e.senders, e.receivers) for e in graph.edge_sets]
    return MultiGraph(nodes, sets)


def _copy_in(static: MultiGraph, node_features: Sequence[torch.Tensor], edge_features: Dict[str, torch.Tensor]):
    for dst, src in zip(static.node_features, node_features):
        dst.copy_(src, non_blocking=True)
    for e in static.edge_sets:
        if e.name in edge_features:
            e.features.copy_(edge_features[e.name], non_blocking=True)


class GraphedForward:
    """model(graph) for a fixed topology, replayed from a HIP graph (inference / rollout).
    The packed operand images of the weights (ops.packs_of: one small launch per MLP) are NOT part of the captured graph: between two
    rollout steps the weights do not change, and at one graph per step those launches were a sixth of the replay.  Every call
    compares the parameters' version counters and the pack epoch with what the images were made from and re-packs eagerly, into
    the same buffers, when they have moved (an optimiser step between two rollouts)."""

    def __init__(self, model: torch.nn.Module, example: MultiGraph, warmup: int = 2):
        self.model = model
        self.static = _static_copy(example)
        self.warmup = max(1, warmup)
        self.captures = 0
        self._capture()

    def _capture(self):
        model = self.model
        ctx = self.ctx = getattr(model, '_hgn_ctx', None) or ops.default_context()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        rec = []
        with torch.cuda.stream(side), torch.no_grad():
            for i in range(self.warmup):
                if i == self.warmup - 1:
                    ctx.pack_recorder = rec
                try:
                    model(self.static)
                finally:
                    ctx.pack_recorder = None
        torch.cuda.current_stream().wait_stream(side)
        seen, self._packs = set(), []
        for w, t in rec:
            if (id(w.w1), t) not in seen:
                seen.add((id(w.w1), t))
                self._packs.append((w, t))
        self._sig = ops.pack_signature(self._packs, ctx)
        self._where = ops.storage_signature(self._packs)
        self.graph = torch.cuda.CUDAGraph()
        ctx.pack_in_capture = False
        try:
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self.out = model(self.static)
        finally:
            ctx.pack_in_capture = True
        self.captures += 1

    def __call__(self, node_features: Sequence[torch.Tensor], edge_features: Dict[str, torch.Tensor]) -> torch.Tensor:
        if ops.storage_signature(self._packs) != self._where:
            # parameter storage was re-homed since the capture (FlatParams, model.to(): the old addresses are baked into the
            # graph and may be freed memory by now): capture again on the same static inputs
            self._capture()
        sig = ops.pack_signature(self._packs, self.ctx)
        if sig != self._sig:                        # the weights were updated since the images were made: same buffers, new contents
            for w, t in self._packs:
                ops.packs_of(w, t, ctx=self.ctx)
            self._sig = ops.pack_signature(self._packs, self.ctx)
        _copy_in(self.static, node_features, edge_features)
        self.graph.replay()
        return self.out


class _PlanGuard:
    """What a captured training step baked in of its context's ops.PackPlan: the device address of the descriptor table (argument of
    the one pack launch) and the packed-image buffers (arguments of every split-product kernel).  Holds references to both -- a
    replay never reads freed memory, whatever happened to the plan since -- and tells when the plan's CONTENT moved (a parameter
    was re-homed, the precision changed): the captured step then holds stale addresses and must be captured again.  A bumped
    ops storage epoch with unchanged addresses (some OTHER model's .to() / FlatParams) refreshes nothing and is not stale."""

    def __init__(self, ctx):
        self.ctx, self.plan = ctx, ctx.pack_plan
        if self.plan is not None:
            self.plan.prepare(ctx)
            self.generation = self.plan.generation
            self.keep = (self.plan.table, list(self.plan.bufs))

    def stale(self) -> bool:
        plan = self.ctx.pack_plan
        if plan is not self.plan:
            return True
        if plan is None:
            return False
        plan.prepare(self.ctx)                                # eager: rewrites the table in place if a descriptor changed
        return plan.generation != self.generation


class GraphedTrainStep:
    """parallel.DataParallelTrainer.step (forward + loss + backward + fused Adam) for a fixed topology, single process.
    The trainer must have been built with ``device_step=True`` (the Adam step counter lives on the device)."""

    def __init__(self, trainer, example: MultiGraph, target: torch.Tensor, mask: torch.Tensor, warmup: int = 3):
        if trainer.world != 1:
            raise RuntimeError('graph capture of the training step is single-process; the collective stays eager')
        if trainer.t_dev is None:
            raise RuntimeError('build the trainer with device_step=True')
        self.trainer = trainer
        self.static = _static_copy(example)
        self.target = target.detach().clone()
        self.mask = mask.detach().clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                trainer.step(self.static, self.target, self.mask)
        torch.cuda.current_stream().wait_stream(side)
        self.captures = 0
        self._capture()

    def _capture(self) -> None:
        trainer = self.trainer
        self._plan = _PlanGuard(trainer.ctx)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = trainer.step(self.static, self.target, self.mask)
        self.captures += 1

    def __call__(self, node_features=None, edge_features: Optional[Dict[str, torch.Tensor]] = None, target=None) -> torch.Tensor:
        if node_features is not None:
            _copy_in(self.static, node_features, edge_features or {})
        if target is not None:
            self.target.copy_(target, non_blocking=True)
        if self._plan.stale():
            self._capture()
        self.graph.replay()
        self.trainer.ctx.invalidate_packs()      # the replayed Adam kernel rewrote the parameters: an eager forward must re-pack them
        return self.loss


class GraphedShardStep:
    """Data-parallel step with the collectives OUTSIDE the graph: forward + local masked squared-error sum + backward of
    this rank's shard are captured once and replayed; the single all-reduce ([flat gradient | NORMAL-node count]), the scaling
    to the GLOBAL mean (flag.py:150-152 semantics over the whole batch) and the fused Adam stay eager.  Works for any world
    size (world 1: no collective), keeps the measured step independent of host-side launch jitter, and needs no support for
    capturing RCCL calls."""

    def __init__(self, trainer, example: MultiGraph, target: torch.Tensor, mask: torch.Tensor, warmup: int = 2):
        self.trainer = trainer
        trainer.overlap = False          # the backward pass is ONE replayed graph: nothing to overlap with, one collective behind it
        self.static = _static_copy(example)
        self.target = target.detach().clone()
        self.maskf = mask.detach().to(torch.float32).unsqueeze(1).clone()
        self.n_local = mask.sum().to(torch.float32).reshape(1)
        self.width = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        self.captures = 0
        self._capture()

    def _capture(self) -> None:
        self._plan = _PlanGuard(self.trainer.ctx)             # (the descriptor table is written eagerly, never inside a capture)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: calls made by other threads of the process (the collective backend's watchdog) must not abort the capture
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            self.sq_sum = self._fwd_bwd()
        self.captures += 1

    def _fwd_bwd(self) -> torch.Tensor:
        from . import ops
        tr = self.trainer
        ops.discard_stale_wgrad(tr.ctx)        # tasks a failed backward pass left queued belong to no step
        tr.fp.zero_grad()
        tr._pending, tr._done_upto = [], None
        if tr.side is not None:
            tr.side.wait_stream(torch.cuda.current_stream())
            tr.ctx.wgrad_stream = tr.side
        ops.begin_step_packs(tr.ctx)
        out = tr.model(self.static)
        self.width = out.shape[1]
        s = ((out - self.target) * self.maskf).square().sum()
        s.backward()
        ops.end_step_packs(tr.ctx)
        if tr.side is not None:
            tr.ctx.wgrad_stream = None
            torch.cuda.current_stream().wait_stream(tr.side)
        return s.detach()

    def __call__(self, node_features=None, edge_features: Optional[Dict[str, torch.Tensor]] = None, target=None,
                 mask=None) -> torch.Tensor:
        tr = self.trainer
        if node_features is not None:
            _copy_in(self.static, node_features, edge_features or {})
        if target is not None:
            self.target.copy_(target, non_blocking=True)
        if mask is not None:                                  # device-side updates of the buffers the graph reads
            self.maskf.copy_(mask.to(torch.float32).unsqueeze(1))
            self.n_local.copy_(mask.sum().to(torch.float32).reshape(1))
        if self._plan.stale():                                # an address the graph baked in has moved: capture again
            self._capture()
        self.graph.replay()                                   # zero_grad + forward + local squared-error sum + backward
        tr.fp.count.copy_(self.n_local)
        return tr.reduce_and_update(self.sq_sum, self.width)  # one all-reduce (world > 1), global-mean scaling, Adam


class GraphedStepCache:
    """The captured training step in the reference's REAL loop (MeshSimulator.py:130-152): every iteration hands over a freshly
    built batch -- new feature tensors AND new index tensors -- but all batches of a trajectory are unions of one mesh.  The
    topology cache (topology.edge_topology: producer key or content fingerprint) maps the fresh index tensors to the
    EdgeTopology objects built for the first batch; this class maps that tuple of topologies to a GraphedShardStep captured
    once (its warm-up runs forward + backward only and never touches the parameters), copies the new features / targets /
    mask into its static buffers and replays.  First sight of a topology: capture (tens of milliseconds); afterwards a step
    costs one fingerprint launch + 16-byte read-back per edge set, the input copies and the replay."""

    def __init__(self, trainer, max_entries: int = 8):
        import collections
        self.trainer = trainer
        self.entries = collections.OrderedDict()
        self.max_entries = max_entries
        self.captures = 0

    def step(self, graph: MultiGraph, target: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        from . import topology
        n_tot = sum(x.shape[0] for x in graph.node_features)
        dev = graph.node_features[0].device
        topos = [topology.edge_topology(e.senders, e.receivers, n_tot, dev) for e in graph.edge_sets]
        key = tuple((e.name, id(t), tuple(e.features.shape)) for e, t in zip(graph.edge_sets, topos)) + \
            tuple(tuple(x.shape) for x in graph.node_features)
        hit = self.entries.get(key)
        if hit is None:
            gs = GraphedShardStep(self.trainer, graph, target, mask, warmup=1)
            self.entries[key] = (gs, topos)                   # the topologies stay alive with the graph that uses them
            self.captures += 1
            while len(self.entries) > self.max_entries:
                self.entries.popitem(last=False)
            return gs()
        self.entries.move_to_end(key)
        return hit[0](graph.node_features, {e.name: e.features for e in graph.edge_sets}, target, mask)


class GraphedForwardCache:
    """GraphedForward for the reference's rollout loops (FlagModel.rollout / _step_fn, flag.py:192-246): every step hands over a freshly
    built graph -- new feature tensors, often new index tensors -- of ONE mesh.  The topology cache maps the index tensors to the
    EdgeTopology objects of the first step; this class maps that tuple to a GraphedForward captured once, copies the features
    into its static buffers and replays.  A step with a topology not seen before (plate: the world edges move with the obstacle)
    runs eagerly; the SECOND sight of a topology captures it (a capture costs tens of milliseconds: a trajectory whose topology never
    repeats stays eager), later ones replay.  At most `max_entries` graphs are kept (least recently used first out)."""

    def __init__(self, model: torch.nn.Module, max_entries: int = 4):
        import collections
        self.model = model
        self.entries = collections.OrderedDict()
        self.seen = collections.OrderedDict()
        self.max_entries = max_entries
        self.captures = 0

    def __call__(self, graph: MultiGraph) -> torch.Tensor:
        from . import topology
        n_tot = sum(x.shape[0] for x in graph.node_features)
        dev = graph.node_features[0].device
        topos = [topology.edge_topology(e.senders, e.receivers, n_tot, dev) for e in graph.edge_sets]
        key = tuple((e.name, id(t), tuple(e.features.shape)) for e, t in zip(graph.edge_sets, topos)) + \
            tuple(tuple(x.shape) for x in graph.node_features)
        hit = self.entries.get(key)
        if hit is None and key not in self.seen:
            self.seen[key] = topos
            while len(self.seen) > 4 * self.max_entries:
                self.seen.popitem(last=False)
            with torch.no_grad():
                return self.model(graph)
        if hit is None:
            gf = GraphedForward(self.model, MultiGraph(list(graph.node_features), list(graph.edge_sets)))
            self.entries[key] = (gf, topos)                   # the topologies stay alive with the graph that uses them
            self.captures += 1
            while len(self.entries) > self.max_entries:
                self.entries.popitem(last=False)
            hit = self.entries[key]
        else:
            self.entries.move_to_end(key)
        # (a clone: the caller keeps the result across steps, the static output buffer is overwritten by the next replay)
        return hit[0](graph.node_features, {e.name: e.features for e in graph.edge_sets}).clone()
