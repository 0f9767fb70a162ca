"""The reference's GNN core (src/migration/*.py) re-built on the HIP kernels, with the same class names, constructor
signatures, module tree and ``state_dict`` keys, so that the reference's system models (src/model/flag.py:55-63,
plate.py:59-67, cylinder.py:55-63) construct and call it unchanged.

The module tree only *holds* parameters (real nn.LazyLinear / nn.LayerNorm objects, so lazy materialisation,
default init, ``.to()``, ``parameters()`` before the first forward (MeshSimulator.py:110) and pickling behave as in
the reference).  ``forward`` never calls those modules: it hands their tensors to the fused kernels.

Inside the processor the graph is kept in an internal layout (``_Latent``): edge latents of every edge set are
stored in receiver-sorted (CSR) order for the whole stack of blocks; they are permuted once in the encoder (a gather
folded into the encoder MLP's load) and never permuted back, because MeshGraphNet decodes node rows only
(meshgraphnet.py:50).  Block modules called stand-alone with a public MultiGraph convert in and out.
"""
import collections
import functools
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Type

import torch
from torch import nn, Tensor

from . import ops, topology
from ._lib import HgnError
from .util import EdgeSet, MultiGraph, device

PNA = ('sum', 'mean', 'max', 'min')                       # graphnet.py:52-64


# ----------------------------------------------------------------------------------------------------------------
# parameter holders
# ----------------------------------------------------------------------------------------------------------------
class LazyMLP(nn.Module):
    """meshgraphnet.py:93-108 -- Linear/ReLU stack whose input width is discovered at first use."""

    def __init__(self, output_sizes: List[int]):
        super().__init__()
        num_layers = len(output_sizes)
        layers = OrderedDict()
        for index, output_size in enumerate(output_sizes):
            layers['linear_' + str(index)] = nn.LazyLinear(output_size)
            if index < (num_layers - 1):
                layers['relu_' + str(index)] = nn.ReLU()
        self.layers = nn.Sequential(layers)

    def forward(self, input: Tensor) -> Tensor:
        return fused_apply(self, [input.to(device)])


def _split(module: nn.Module) -> Tuple[LazyMLP, Optional[nn.LayerNorm]]:
    if isinstance(module, LazyMLP):
        return module, None
    return module[0], module[1]


def _linears(mlp: LazyMLP) -> List[nn.Module]:
    return [m for n, m in mlp.layers.named_children() if n.startswith('linear_')]


def materialize(module: nn.Module, in_features: int):
    """Give the lazy layers their shapes exactly as a first reference forward would (same init, same RNG order)."""
    mlp, _ = _split(module)
    width = in_features
    for lin in _linears(mlp):
        if isinstance(lin.weight, nn.parameter.UninitializedParameter):
            dev = lin.weight.device
            with torch.no_grad():
                lin(torch.empty(0, width, device=dev))           # runs LazyLinear's own materialisation hook
        elif lin.weight.shape[1] != width:
            raise HgnError(f'MLP expects {lin.weight.shape[1]} input features, got {width}')
        width = lin.weight.shape[0]


def weights_of(module: nn.Module, in_features: int) -> ops.MLPWeights:
    materialize(module, in_features)
    mlp, ln = _split(module)
    lins = _linears(mlp)
    if len(lins) != 3:
        raise HgnError('the HIP path implements the reference setting num_layers=2 (three Linear layers per MLP)')
    w = ops.MLPWeights(lins[0].weight, lins[0].bias, lins[1].weight, lins[1].bias, lins[2].weight, lins[2].bias,
                       ln.weight if ln is not None else None, ln.bias if ln is not None else None)
    return w


def fused_apply(module: nn.Module, srcs: Sequence[Tensor], idxs=None, residual: int = -1, post=None, cols=None, width=None, share=False):
    """`cols` / `width`: first input column of every source and the full input width, when column ranges are left out (ops.fused_mlp)."""
    width = sum(s.shape[1] for s in srcs) if width is None else width
    return ops.fused_mlp(srcs, weights_of(module, width), idxs, residual, post, cols, share)


# ----------------------------------------------------------------------------------------------------------------
# internal graph layout
# ----------------------------------------------------------------------------------------------------------------
class _SplitRows(torch.autograd.Function):
    """(a[:n], a[n:]) as views, with ONE backward that concatenates the two gradients.  Plain slicing costs, per use and per
    layer, a zero-filled full-size gradient plus a copy (and an add where mesh-row and hyper-row MLPs consume the same
    aggregate): 2.8 ms of a 59 ms hyper/pna step."""

    @staticmethod
    def forward(ctx, a, n):
        ctx.set_materialize_grads(False)
        ctx.n, ctx.rows, ctx.tail = n, a.shape[0], a.shape[1:]
        ctx.opts = dict(dtype=a.dtype, device=a.device)
        return a[:n], a[n:]

    @staticmethod
    def backward(ctx, g0, g1):
        if g0 is None and g1 is None:
            return None, None
        if g0 is not None and g1 is not None:
            return torch.cat((g0, g1), dim=0), None
        out = torch.empty((ctx.rows,) + tuple(ctx.tail), **ctx.opts)         # one side unused: zero only that side
        if g0 is not None:
            out[:ctx.n].copy_(g0); out[ctx.n:].zero_()
        else:
            out[ctx.n:].copy_(g1); out[:ctx.n].zero_()
        return out, None


class _Agg:
    """Aggregate of one edge set over its receivers: rows of node part `part` (0 mesh, 1 hyper) only, or of all node rows (None)."""
    __slots__ = ('t', 'part')

    def __init__(self, t: Tensor, part: Optional[int]):
        self.t, self.part = t, part


class _Latent:
    __slots__ = ('nodes', 'edges', 'topo', 'splits', 'pre', '_cat')

    def __init__(self, nodes: List[Tensor], edges: 'OrderedDict[str, Tensor]', topo: Dict[str, topology.EdgeTopology]):
        self.nodes, self.edges, self.topo = nodes, edges, topo
        self.splits = {}
        self.pre = {}       # edge-set name -> (P, zero-filled aggregate buffer) formed by the node kernel of the block before (inference)
        self._cat = None

    def split_rows(self, a: Tensor):
        """(mesh rows, hyper rows) of an [N_tot, .] tensor; one autograd node per tensor however often it is consumed."""
        hit = self.splits.get(id(a))
        if hit is None or hit[0] is not a:
            hit = (a, _SplitRows.apply(a, self.n_mesh))
            self.splits[id(a)] = hit
        return hit[1]

    def h_all(self) -> Tensor:
        """All node rows as ONE tensor -- only for edge sets whose senders or receivers lie in both parts (topology.parts is None):
        the sets of the reference's hierarchies read one part per side and never come here."""
        if len(self.nodes) == 1:
            return self.nodes[0]
        if self._cat is None or any(a is not b for a, b in zip(self._cat[0], self.nodes)):
            self._cat = (tuple(self.nodes), torch.cat(tuple(self.nodes), dim=0))
        return self._cat[1]

    @property
    def n_mesh(self):
        return self.nodes[0].shape[0]

    def parts_known(self) -> bool:
        """Two node parts, and every edge set reads one part per side: nobody calls h_all(), so the part tensors are consumed by the
        blocks' own autograd nodes (edge blocks, node updates) only -- what ops.share_grad needs to be vouched for."""
        return len(self.nodes) == 2 and all(t.parts(self.n_mesh) is not None for t in self.topo.values())


def _num_rows(node_features) -> int:
    return sum(x.shape[0] for x in node_features)


def to_latent(graph: MultiGraph) -> _Latent:
    """Public MultiGraph (edge rows in user order) -> internal layout (receiver-sorted rows)."""
    nodes = [x.to(device) for x in graph.node_features]
    n_tot = _num_rows(nodes)
    edges, topo = OrderedDict(), {}
    for es in graph.edge_sets:
        t = topology.edge_topology(es.senders, es.receivers, n_tot, nodes[0].device)
        topo[es.name] = t
        edges[es.name] = es.features.to(device).index_select(0, t.r.perm.long())
    return _Latent(nodes, edges, topo)


def to_public(lat: _Latent, like: MultiGraph) -> MultiGraph:
    by_name = {e.name: e for e in like.edge_sets}
    sets = []
    for name, feat in lat.edges.items():
        t = lat.topo[name]
        sets.append(by_name[name]._replace(features=feat.index_select(0, t.inverse_perm())))
    return MultiGraph(lat.nodes, sets)


# ----------------------------------------------------------------------------------------------------------------
# blocks
# ----------------------------------------------------------------------------------------------------------------
class GraphNet(nn.Module):
    """Multi-Edge Interaction Network with residual connections (graphnet.py:11-124)."""

    def __init__(self, model_fn: Callable, output_size: int, message_passing_aggregator: str, edge_sets: List[str]):
        super().__init__()
        self.node_model_cross = model_fn(output_size)
        self.edge_models = nn.ModuleDict({name: model_fn(output_size) for name in edge_sets})
        self.message_passing_aggregator = message_passing_aggregator
        # The reference iterates Python *sets* of names in HyperGraphNet/MultiScaleGraphNet (hypergraphnet.py:31,44), so
        # the column order of those node-MLP inputs depends on PYTHONHASHSEED; here it is this fixed list.
        self.set_order = ['mesh_edges', 'world_edges', 'inter_cluster', 'inter_cluster_world']

    # -- stage primitives on the internal layout -----------------------------------------------------------------
    def _ops(self) -> Tuple[str, ...]:
        return PNA if self.message_passing_aggregator == 'pna' else (self.message_passing_aggregator,)

    def _edge(self, lat: _Latent, feats: Tensor, name: str) -> Tuple[Tensor, _Agg]:
        """-> (updated edge latents, their aggregates over receivers [rows, k*128]); one autograd node for both.
        Mesh rows and hyper rows stay two tensors: a set whose senders lie in one part and whose receivers lie in one part (all sets
        of the reference's hierarchies: mesh->mesh, mesh->hyper, hyper->hyper, hyper->mesh) hands over just those, and its aggregate
        has the receiver part's rows (graphnet.py:25-26 indexes ONE concatenated tensor; same values, no concatenation)."""
        t = lat.topo[name]
        w = weights_of(self.edge_models[name], 3 * ops.LAT)
        pp = t.parts(lat.n_mesh) if len(lat.nodes) == 2 else None
        if pp is not None:
            ps, pr = pp
            offs = (0, lat.n_mesh)
            e2, agg = ops.edge_block(lat.nodes[ps], feats, t, w, self._ops(), parts=(offs[ps], offs[pr]),
                                     h_r=None if pr == ps else lat.nodes[pr])
            return e2, _Agg(agg, pr)
        e2, agg = ops.edge_block(lat.h_all(), feats, t, w, self._ops(), pre=lat.pre.get(name))
        return e2, _Agg(agg, None)

    def _node(self, lat: _Latent, aggs: Sequence[_Agg], model: nn.Module, which: int):
        """nodes[which] += LN(MLP([h ; agg_1 ; agg_2 ...][rows of `which`]))   (graphnet.py:47-48,107-108,123-124).
        The concatenation is never materialised: every aggregate is its own K-segment of the first Linear -- and an aggregate over
        edges that all arrive in the OTHER part is zero for these rows: its K-segment is left out (ops.fused_mlp: cols)."""
        n_mesh = lat.n_mesh
        srcs, cols, col = [lat.nodes[which]], [0], lat.nodes[which].shape[1]
        for a in aggs:
            if a.part is None:
                # (no slice when there are no hyper rows: its backward would zero-fill and copy a full [N, k*128] gradient)
                srcs.append(a.t if (which == 0 and a.t.shape[0] == n_mesh) else lat.split_rows(a.t)[which])
                cols.append(col)
            elif a.part == which:
                srcs.append(a.t)
                cols.append(col)
            col += a.t.shape[1]
        # (the part's latents are read by this update and by the block's edge blocks -- our own autograd nodes -- only: the update's
        #  gradient tensor doubles as their accumulation target, ops.share_grad.  One part: the plain blocks; two parts: every
        #  schedule, as long as no edge set needs the concatenated rows -- torch.cat would be a consumer that reports on its own)
        share = (len(lat.nodes) == 1 and type(self) in (GraphNet, MultiGraphNet, RepeatedGraphNet)) or lat.parts_known()
        lat.nodes[which] = fused_apply(model, srcs, residual=0, cols=cols if len(srcs) <= len(aggs) else None, width=col, share=share)

    # -- GraphNet.forward (graphnet.py:72-84) --------------------------------------------------------------------
    def _forward_latent(self, lat: _Latent, nxt: Optional['GraphNet'] = None) -> _Latent:
        new_edges, aggs = OrderedDict(), OrderedDict()
        for name, feats in lat.edges.items():
            if name not in self.edge_models:
                raise KeyError(name)                                           # graphnet.py:32
            new_edges[name], aggs[name] = self._edge(lat, feats, name)
        out = _Latent(list(lat.nodes), new_edges, lat.topo)
        if nxt is not None:                                                     # (plain blocks only: Processor.forward)
            self._update_nodes(out, aggs, nxt)
        else:
            self._update_nodes(out, aggs)                                       # subclasses override this two-argument form
        return out

    def _update_nodes(self, lat: _Latent, aggs, nxt: Optional['GraphNet'] = None):
        # A plain stack of blocks with ONE edge set (Processor.forward): the node kernel also forms the NEXT block's
        # node-level pre-projection P = [h W1s^T | h W1r^T] and the zero fill of its aggregate buffer while the rows are in registers
        post = None
        if nxt is not None and len(lat.edges) == 1:
            name = next(iter(lat.edges))
            if name in nxt.edge_models:
                pk = ops.packs_of(weights_of(nxt.edge_models[name], 3 * ops.LAT))
                if pk is not None:
                    post = (pk, nxt.message_passing_aggregator == 'sum')
        if post is None:
            self._node(lat, list(aggs.values()), self.node_model_cross, 0)   # graph order (graphnet.py:43)
            return
        srcs = [lat.nodes[0]] + [a.t for a in aggs.values()]
        lat.nodes[0], got = fused_apply(self.node_model_cross, srcs, residual=0, post=post, share=len(lat.nodes) == 1)
        if got is not None:
            lat.pre = {next(iter(lat.edges)): got}

    def forward(self, graph, mask=None):
        if isinstance(graph, _Latent):
            nxt = self.__dict__.get('_next_block')            # set by Processor.forward around this call (plain blocks only)
            return self._forward_latent(graph, nxt) if nxt is not None else self._forward_latent(graph)
        return to_public(self._forward_latent(to_latent(graph)), graph)

    # -- helpers shared by the hierarchical blocks (graphnet.py:86-124) -----------------------------------------
    def _edges_stage(self, lat: _Latent, src_edges, name: str, new_edges, aggs):
        if name not in self.edge_models:                                      # graphnet.py:87-88
            return
        if name not in src_edges:
            raise IndexError(f'edge set {name!r} is registered but missing from the graph')   # graphnet.py:90
        new_edges[name], aggs[name] = self._edge(lat, src_edges[name], name)

    def _pick(self, aggs, pair):
        return [aggs[n] for n in self.set_order if n in pair and n in self.edge_models]


class MultiGraphNet(GraphNet):
    """multigraphnet.py:10-18 -- identical to GraphNet."""


class RepeatedGraphNet(GraphNet):
    """repeatedgraphnet.py:11-22 -- GraphNet.forward applied `repetitions` times with shared weights."""

    def __init__(self, model_fn, output_size, message_passing_aggregator, edge_sets, repetitions=2):
        super().__init__(model_fn, output_size, message_passing_aggregator, edge_sets)
        self.repetitions = repetitions

    def _forward_latent(self, lat):
        for _ in range(self.repetitions):
            lat = GraphNet._forward_latent(self, lat)
        return lat


class HeteroGraphNet(GraphNet):
    """heterographnet.py:10-33 -- one aggregation over all sets feeding a mesh-row MLP and a hyper-row MLP."""

    def __init__(self, model_fn, output_size, message_passing_aggregator, edge_sets):
        super().__init__(model_fn, output_size, message_passing_aggregator, edge_sets)
        self.hyper_node_model_cross = model_fn(output_size)

    def _update_nodes(self, lat, aggs):
        al = list(aggs.values())
        self._node(lat, al, self.node_model_cross, 0)         # rows of one kind never feed the other kind's MLP input,
        self._node(lat, al, self.hyper_node_model_cross, 1)   # so the two updates commute (heterographnet.py:29-32)


class HyperGraphNet(GraphNet):
    """hypergraphnet.py:11-54 -- eight sequential stages, each seeing the node rows updated by the previous ones."""

    def __init__(self, model_fn, output_size, message_passing_aggregator, edge_sets):
        super().__init__(model_fn, output_size, message_passing_aggregator, edge_sets)
        self.hyper_node_model_up = model_fn(output_size)
        self.hyper_node_model_cross = model_fn(output_size)
        self.node_model_down = model_fn(output_size)

    def _cross_models(self):
        return [self.hyper_node_model_cross]

    def _forward_latent(self, lat):
        src = lat.edges
        lat = _Latent(list(lat.nodes), src, lat.topo)
        new, aggs = OrderedDict(), {}
        E = functools.partial(self._edges_stage, lat, src)
        E('mesh_edges', new, aggs); E('world_edges', new, aggs)
        self._node(lat, self._pick(aggs, ('mesh_edges', 'world_edges')), self.node_model_cross, 0)
        E('intra_cluster_to_cluster', new, aggs)
        self._node(lat, [aggs['intra_cluster_to_cluster']], self.hyper_node_model_up, 1)
        for model in self._cross_models():
            E('inter_cluster', new, aggs); E('inter_cluster_world', new, aggs)
            self._node(lat, self._pick(aggs, ('inter_cluster', 'inter_cluster_world')), model, 1)
        E('intra_cluster_to_mesh', new, aggs)
        self._node(lat, [aggs['intra_cluster_to_mesh']], self.node_model_down, 0)
        self._tail(lat, src, new, aggs)
        return _Latent(lat.nodes, new, lat.topo)

    def _tail(self, lat, src, new, aggs):
        pass


class MultiScaleGraphNet(HyperGraphNet):
    """multiscalegraphnet.py:10-63 -- up, 3 x inter-cluster, down, then a second mesh update from the INPUT edges."""

    def __init__(self, model_fn, output_size, message_passing_aggregator, edge_sets):
        GraphNet.__init__(self, model_fn, output_size, message_passing_aggregator, edge_sets)
        self.hyper_node_model_up = model_fn(output_size)
        self.hyper_node_models_cross = nn.ModuleList([model_fn(output_size) for _ in range(3)])
        self.node_model_down = model_fn(output_size)

    def _cross_models(self):
        return list(self.hyper_node_models_cross)

    def _tail(self, lat, src, new, aggs):
        self._edges_stage(lat, src, 'mesh_edges', new, aggs)
        self._edges_stage(lat, src, 'world_edges', new, aggs)
        self._node(lat, self._pick(aggs, ('mesh_edges', 'world_edges')), self.node_model_cross, 0)


# ----------------------------------------------------------------------------------------------------------------
# encoder / processor / decoder / model
# ----------------------------------------------------------------------------------------------------------------
class Encoder(nn.Module):
    """encoder.py:9-47."""

    def __init__(self, make_mlp: Callable, latent_size: int, edge_sets: List[str], hierarchical=True):
        super().__init__()
        self._make_mlp = make_mlp
        self._latent_size = latent_size
        self.node_model = self._make_mlp(latent_size)
        self.edge_models = nn.ModuleDict({name: self._make_mlp(latent_size) for name in edge_sets})
        self.hierarchical = hierarchical
        if hierarchical:
            self.hyper_node_model = self._make_mlp(latent_size)

    def _encode(self, graph: MultiGraph) -> _Latent:
        raw = [x.to(device) for x in graph.node_features]
        nodes = [fused_apply(self.node_model, [raw[0]])]
        if len(raw) > 1:
            nodes.append(fused_apply(self.hyper_node_model if self.hierarchical else self.node_model, [raw[1]]))
        n_tot = _num_rows(raw)
        edges, topo = OrderedDict(), {}
        for es in graph.edge_sets:
            if es.name not in self.edge_models:                               # encoder.py:44-45: dropped
                continue
            t = topology.edge_topology(es.senders, es.receivers, n_tot, raw[0].device)
            topo[es.name] = t
            # the permutation into receiver-sorted order is the row-gather index of the encoder MLP's first load
            edges[es.name] = fused_apply(self.edge_models[es.name], [es.features.to(device)], idxs=[t.r.perm])
        return _Latent(nodes, edges, topo)

    def forward(self, graph: MultiGraph) -> MultiGraph:
        lat = self._encode(graph)
        return to_public(lat, graph)


class Processor(nn.Module):
    """processor.py:10-28."""

    def __init__(self, make_mlp: Callable, output_size: int, message_passing_steps: int, message_passing_aggregator: str,
                 edge_sets: List[str], graphnet_block: Type[GraphNet]):
        super().__init__()
        blocks = []
        for _ in range(message_passing_steps):
            blocks.append(graphnet_block(model_fn=make_mlp, output_size=output_size,
                                         message_passing_aggregator=message_passing_aggregator, edge_sets=edge_sets))
        self.graphnet_blocks = nn.Sequential(*blocks)

    def forward(self, latent_graph):
        blocks = list(self.graphnet_blocks)
        # plain GraphNet blocks on one node type: block i hands block i + 1 its pre-projection (GraphNet._update_nodes) -- in inference and,
        # since round 5, in training (the edge block's backward does not care who multiplied h by W1s / W1r in the forward)
        if (isinstance(latent_graph, _Latent) and len(latent_graph.nodes) == 1
                and all(type(b) in (GraphNet, MultiGraphNet) for b in blocks)):
            lat = latent_graph
            for i, b in enumerate(blocks):
                b.__dict__['_next_block'] = blocks[i + 1] if i + 1 < len(blocks) else None      # (not a registered submodule)
                try:
                    lat = b(lat)             # through Module.__call__: forward hooks (the trainer's bucket boundaries, parallel.py) must fire
                finally:
                    b.__dict__['_next_block'] = None
            return lat
        return self.graphnet_blocks(latent_graph)


class Decoder(nn.Module):
    """decoder.py:8-16."""

    def __init__(self, make_mlp: Callable, output_size: int):
        super().__init__()
        self.model = make_mlp(output_size)

    def forward(self, graph) -> Tensor:
        return fused_apply(self.model, [graph.node_features])


class MeshGraphNet(nn.Module):
    """Encode-Process-Decode GraphNet model (meshgraphnet.py:21-89)."""

    def __init__(self, output_size: int, latent_size: int, num_layers: int, message_passing_aggregator: str,
                 message_passing_steps: int, architecture: str, edge_sets: List[str]):
        super().__init__()
        self._latent_size = latent_size
        self._output_size = output_size
        self._num_layers = num_layers
        self._message_passing_steps = message_passing_steps
        self._message_passing_aggregator = message_passing_aggregator
        graphnet_block, hierarchical = self.get_architecture(architecture)
        self.encoder = Encoder(make_mlp=self._make_mlp, latent_size=self._latent_size, hierarchical=hierarchical,
                               edge_sets=edge_sets)
        self.processor = Processor(make_mlp=self._make_mlp, output_size=self._latent_size,
                                   message_passing_steps=self._message_passing_steps,
                                   message_passing_aggregator=self._message_passing_aggregator, edge_sets=edge_sets,
                                   graphnet_block=graphnet_block)
        self.decoder = Decoder(make_mlp=functools.partial(self._make_mlp, layer_norm=False), output_size=self._output_size)
        # Launch context of THIS model (ops.Context): precision mode and kernel-selection flags, the deferred weight-gradient queue,
        # pack epoch, workspaces.  Two models -- or two threads -- in one process share none of it; fields left unset follow the
        # process defaults (hgn_amd.set_matmul_precision).
        self._hgn_ctx = ops.Context()

    def set_matmul_precision(self, mode) -> None:
        """This model's own precision: 'fp32' (six split-bf16 products, fp32 accurate), 'bf16', 'fp16', or None = follow the
        process default.  Other models of the process are not affected."""
        self._hgn_ctx.set_matmul_precision(mode)

    def forward(self, graph: MultiGraph) -> Tensor:
        if self._latent_size != ops.LAT or self._num_layers != 2:
            raise HgnError('the HIP path implements latent_size=128, num_layers=2 (hard-coded by the reference models: '
                           'src/model/flag.py:57-58, plate.py:61-62, cylinder.py:57-58)')
        with ops.using(self.__dict__.get('_hgn_ctx') or ops.default_context()):
            lat = self.encoder._encode(graph)
            lat = self.processor(lat)
            return self.decoder(MultiGraph(lat.nodes[0], None))

    def _make_mlp(self, output_size: int, layer_norm=True) -> nn.Module:
        """`num_layers` hidden Linear+ReLU stages of the latent width, one output Linear; the LayerNorm'd variant is a Sequential
        whose positions 0 / 1 are what the state_dict keys name (`...0.layers.linear_k`, `...1.weight`; meshgraphnet.py:53-60)."""
        hidden = (self._latent_size,) * self._num_layers
        mlp = LazyMLP([*hidden, output_size])
        return nn.Sequential(mlp, nn.LayerNorm(output_size)) if layer_norm else mlp

    def _apply(self, fn, *args, **kwargs):
        # .to() / .float() / .cuda() give every parameter new storage: captured forward graphs hold the old addresses
        ops.storage_moved()
        return super()._apply(fn, *args, **kwargs)

    @staticmethod
    def get_architecture(architecture: str) -> Tuple[Type[GraphNet], bool]:
        table = {'hyper': (HyperGraphNet, True), 'multiscale': (MultiScaleGraphNet, True), 'hetero': (HeteroGraphNet, True),
                 'multi': (MultiGraphNet, False), 'repeated': (RepeatedGraphNet, False)}
        return table.get(architecture, (GraphNet, False))
