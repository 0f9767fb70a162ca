"""CPU oracle for the message-passing hot path of CemOezcan/hyper-graph-nets.

TEST INFRASTRUCTURE ONLY.  Nothing under ``hyper-graph-nets_amd/`` may import this file; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do (as the checker / the timed
baseline, never as the product).

It is a plain-PyTorch (CPU, fp32 or fp64) *restatement* of the reference algorithm, written functionally
over a ``state_dict`` that uses the reference's own key names, so that weights exported by the reference
(tests/golden/gen_golden.py) can be fed in unchanged.  Every function cites the reference lines it follows
(paths relative to /root/reference).  It keeps the reference's op sequence (two gathers + cat + three
Linear + LayerNorm + residual; id broadcast + one scatter pass per aggregate) so that timing it on host
cores measures "the reference's CPU path", not an optimised CPU variant.

Pinning: checked against golden vectors produced by running the *reference itself* in the build container
(tests/golden/*.pt, generator tests/golden/gen_golden.py).  The reference's segment reductions come from the
third-party wheel ``torch-scatter==2.0.9`` which is absent from the image; its published semantics are
restated in :func:`segment_reduce` (zero-initialised output, empty segment -> 0, mean = sum/max(cnt,1),
max/min route gradients to the first arg).  For that primitive alone parity is "unpinned" (the reference
holds no test for it); everything above it is pinned through the goldens.
"""
from __future__ import annotations

import collections
import re
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

EdgeSet = collections.namedtuple('EdgeSet', ['name', 'features', 'senders', 'receivers'])   # src/util.py:11
MultiGraph = collections.namedtuple('MultiGraph', ['node_features', 'edge_sets'])           # src/util.py:12

PNA_OPS = ('sum', 'mean', 'max', 'min')                                                    # graphnet.py:52-64


# --------------------------------------------------------------------------------------------------------
# a2: unsorted_segment_operation (src/util.py:92-134) on top of restated torch_scatter semantics
# --------------------------------------------------------------------------------------------------------
class _FirstArgReduce(torch.autograd.Function):
    """scatter_max / scatter_min with torch_scatter's CPU tie rule (strict compare -> the first element wins; empty -> 0,
    arg = E).  Built on a STABLE SORT by segment id and a scan inside the sorted runs -- deliberately not the construction of
    the import stand-in (tools/oracle_shims/torch_scatter: scatter_reduce + candidate indices) nor of the brute-force loops
    (oracle/scatter_loops.py); tests/test_oracle_golden.py checks the three against each other and against golden G1."""

    @staticmethod
    def forward(ctx, src, index, num_segments, is_max):
        E = src.shape[0]
        D = 1
        for n in src.shape[1:]:
            D *= int(n)
        flat = src.reshape(E, D)
        ids = index.reshape(E, D)
        out = flat.new_zeros(num_segments, D)
        arg = torch.full((num_segments, D), E, dtype=torch.long)
        if E > 0 and D > 0:
            key = flat if is_max else -flat
            # every column independently (2-D sorts along dim 0): rank the elements by (segment asc, value desc, original
            # position asc) with three stable sorts; the head of each run of equal segment ids is that segment's FIRST maximum
            order = torch.sort(ids, dim=0, stable=True).indices                  # original order kept inside a segment
            sseg, sval = ids.gather(0, order), key.gather(0, order)
            o2 = torch.sort(sval, dim=0, stable=True, descending=True).indices
            o3 = torch.sort(sseg.gather(0, o2), dim=0, stable=True).indices
            lead = o2.gather(0, o3)
            seg_sorted = sseg.gather(0, lead)
            head = torch.ones(E, D, dtype=torch.bool)
            head[1:] = seg_sorted[1:] != seg_sorted[:-1]
            winners = order.gather(0, lead)[head]                                # original element ids of the winners
            cols = torch.arange(D).expand(E, D)[head]
            rows = seg_sorted[head]
            out[rows, cols] = flat[winners, cols]
            arg[rows, cols] = winners
        shape = (num_segments,) + tuple(src.shape[1:])
        arg = arg.reshape(shape)
        ctx.save_for_backward(arg)
        ctx.E = E
        ctx.mark_non_differentiable(arg)
        return out.reshape(shape), arg

    @staticmethod
    def backward(ctx, gval, _):
        (arg,) = ctx.saved_tensors
        E = ctx.E
        D = 1
        for n in gval.shape[1:]:
            D *= int(n)
        g = gval.new_zeros(E, D)
        a2, gv = arg.reshape(-1, D), gval.reshape(-1, D)
        live = a2 < E
        cols = torch.arange(D).expand_as(a2)
        g[a2[live], cols[live]] = gv[live]
        g = g.reshape((E,) + tuple(gval.shape[1:]))
        return g, None, None, None


def segment_reduce(data: torch.Tensor, segment_ids: torch.Tensor, num_segments: int, operation: str,
                   return_arg: bool = False):
    """src/util.py:92-134.  ``data`` is [E, ...], ``segment_ids`` is [E] (or already data-shaped)."""
    assert all(i in data.shape for i in segment_ids.shape)                    # util.py:101
    segment_ids = segment_ids.long()
    if segment_ids.dim() == 1:                                                # util.py:107-110 (id broadcast)
        inner = int(torch.prod(torch.tensor(data.shape[1:])).item()) if data.dim() > 1 else 1
        segment_ids = segment_ids.repeat_interleave(inner).view(segment_ids.shape[0], *data.shape[1:])
    assert data.shape == segment_ids.shape                                    # util.py:112
    src = data if data.dtype == torch.float64 else data.float()               # util.py:117 (.float())
    arg = None
    if operation == 'sum':
        out = src.new_zeros((num_segments,) + tuple(src.shape[1:])).scatter_add(0, segment_ids, src)
    elif operation == 'mean':
        tot = src.new_zeros((num_segments,) + tuple(src.shape[1:])).scatter_add(0, segment_ids, src)
        cnt = src.new_zeros((num_segments,) + tuple(src.shape[1:])).scatter_add(0, segment_ids,
                                                                                 torch.ones_like(src))
        out = tot / cnt.clamp(min=1)
    elif operation == 'max':
        out, arg = _FirstArgReduce.apply(src, segment_ids, num_segments, True)
    elif operation == 'min':
        out, arg = _FirstArgReduce.apply(src, segment_ids, num_segments, False)
    elif operation == 'std':                                                  # util.py:129-130 (torch_scatter.scatter_std, unbiased)
        zeros = src.new_zeros((num_segments,) + tuple(src.shape[1:]))
        cnt = zeros.scatter_add(0, segment_ids, torch.ones_like(src)).clamp(min=1)
        mean = zeros.scatter_add(0, segment_ids, src) / cnt
        dev = src - mean.gather(0, segment_ids)
        out = (zeros.scatter_add(0, segment_ids, dev * dev) / ((cnt - 1).clamp(min=1) + 1e-6)).sqrt()
    else:
        raise Exception('Invalid operation type!')                            # util.py:132
    out = out.type(data.dtype)                                                # util.py:133
    return (out, arg) if return_arg else out


# --------------------------------------------------------------------------------------------------------
# a5: LazyMLP (+LayerNorm)  meshgraphnet.py:53-60,93-108
# --------------------------------------------------------------------------------------------------------
def mlp(sd: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, layer_norm: bool = True) -> torch.Tensor:
    """``prefix`` names the module built by ``MeshGraphNet._make_mlp`` (a Sequential(LazyMLP, LayerNorm) when
    layer_norm else the bare LazyMLP)."""
    lin = (prefix + '.0.layers.') if layer_norm else (prefix + '.layers.')
    i = 0
    while (lin + f'linear_{i}.weight') in sd:
        x = F.linear(x, sd[lin + f'linear_{i}.weight'], sd[lin + f'linear_{i}.bias'])
        if (lin + f'linear_{i + 1}.weight') in sd:
            x = torch.relu(x)
        i += 1
    if layer_norm:
        x = F.layer_norm(x, (x.shape[-1],), sd[prefix + '.1.weight'], sd[prefix + '.1.bias'], 1e-5)
    return x


# --------------------------------------------------------------------------------------------------------
# a1 / a3: edge and node updates  graphnet.py:22-70,94-124
# --------------------------------------------------------------------------------------------------------
def update_edge_features(sd, mlp_prefix: str, node_features: Sequence[torch.Tensor], edge_set: EdgeSet):
    nodes = torch.cat(tuple(node_features), dim=0)                            # graphnet.py:24
    s = torch.index_select(nodes, 0, edge_set.senders.long())                 # graphnet.py:28
    r = torch.index_select(nodes, 0, edge_set.receivers.long())               # graphnet.py:29
    x = torch.cat([s, r, edge_set.features], dim=-1)                          # graphnet.py:30
    return edge_set.features + mlp(sd, mlp_prefix, x)                         # graphnet.py:32


def aggregation(edge_sets: Sequence[EdgeSet], features: List[torch.Tensor], num_nodes: int, aggregator: str):
    for es in edge_sets:                                                      # graphnet.py:50-70
        ops = PNA_OPS if aggregator == 'pna' else (aggregator,)
        for op in ops:
            features.append(segment_reduce(es.features, es.receivers, num_nodes, op))
    return torch.cat(features, dim=-1)


def _node_input(node_features, edge_sets, registered, aggregator):
    nodes = torch.cat(tuple(node_features), dim=0)
    used = [es for es in edge_sets if es.name in registered]                  # graphnet.py:43
    return aggregation(used, [nodes], nodes.shape[0], aggregator)


# --------------------------------------------------------------------------------------------------------
# a4: block schedules
# --------------------------------------------------------------------------------------------------------
def _registered(sd, block_prefix):
    p = block_prefix + '.edge_models.'
    names = []
    for k in sd:
        if k.startswith(p):
            n = k[len(p):].split('.')[0]
            if n not in names:
                names.append(n)
    return names


def graphnet_block(sd, bp: str, graph: MultiGraph, aggregator: str) -> MultiGraph:
    """GraphNet.forward graphnet.py:72-84 (also MultiGraphNet multigraphnet.py:16-18)."""
    reg = _registered(sd, bp)
    new_sets = [es._replace(features=update_edge_features(sd, f'{bp}.edge_models.{es.name}',
                                                          graph.node_features, es))
                for es in graph.edge_sets]
    nf = list(graph.node_features)
    n_mesh = nf[0].shape[0]
    x = _node_input(nf, new_sets, reg, aggregator)
    nf[0] = nf[0] + mlp(sd, f'{bp}.node_model_cross', x[:n_mesh])           # graphnet.py:47-48
    return MultiGraph(nf, new_sets)


def hetero_block(sd, bp, graph, aggregator):
    """HeteroGraphNet: GraphNet.forward with the node update of heterographnet.py:17-33."""
    reg = _registered(sd, bp)
    new_sets = [es._replace(features=update_edge_features(sd, f'{bp}.edge_models.{es.name}',
                                                          graph.node_features, es))
                for es in graph.edge_sets]
    nf = list(graph.node_features)
    n_mesh = nf[0].shape[0]
    x = _node_input(nf, new_sets, reg, aggregator)
    up_mesh = mlp(sd, f'{bp}.node_model_cross', x[:n_mesh])
    up_hyper = mlp(sd, f'{bp}.hyper_node_model_cross', x[n_mesh:])
    nf[0] = nf[0] + up_mesh
    nf[1] = nf[1] + up_hyper
    return MultiGraph(nf, new_sets)


def repeated_block(sd, bp, graph, aggregator, repetitions=2):
    """repeatedgraphnet.py:18-22."""
    for _ in range(repetitions):
        graph = graphnet_block(sd, bp, graph, aggregator)
    return graph


class _Stages:
    """Shared stage helpers of HyperGraphNet / MultiScaleGraphNet (graphnet.py:86-124)."""

    def __init__(self, sd, bp, graph, aggregator, set_order):
        self.sd, self.bp, self.agg = sd, bp, aggregator
        self.reg = _registered(sd, bp)
        self.nf = list(graph.node_features)
        self.in_sets = {es.name: es for es in graph.edge_sets}
        self.new = collections.OrderedDict()
        self.set_order = set_order

    def edges(self, name):                                                    # perform_edge_updates :86-92
        if name not in self.reg:
            return
        es = self.in_sets[name]
        self.new[name] = es._replace(features=update_edge_features(
            self.sd, f'{self.bp}.edge_models.{name}', self.nf, es))

    def pick(self, pair):
        """{'a','b'}.intersection(registered) -- the reference iterates a Python set (hypergraphnet.py:31);
        ``set_order`` fixes the order the goldens were generated under (PYTHONHASHSEED=0)."""
        names = [n for n in pair if n in self.reg]
        if self.set_order is not None:
            names.sort(key=self.set_order.index)
        return [self.new[n] for n in names]

    def nodes(self, sets, model, rows):
        n_mesh = self.nf[0].shape[0]
        x = _node_input(self.nf, sets, self.reg, self.agg)
        if rows == 'mesh':
            self.nf[0] = self.nf[0] + mlp(self.sd, f'{self.bp}.{model}', x[:n_mesh])
        else:
            self.nf[1] = self.nf[1] + mlp(self.sd, f'{self.bp}.{model}', x[n_mesh:])


def hyper_block(sd, bp, graph, aggregator, set_order=None):
    """HyperGraphNet.forward hypergraphnet.py:21-54 (eight sequential stages)."""
    st = _Stages(sd, bp, graph, aggregator, set_order)
    st.edges('mesh_edges'); st.edges('world_edges')
    st.nodes(st.pick(('mesh_edges', 'world_edges')), 'node_model_cross', 'mesh')
    st.edges('intra_cluster_to_cluster')
    st.nodes([st.new['intra_cluster_to_cluster']], 'hyper_node_model_up', 'hyper')
    st.edges('inter_cluster'); st.edges('inter_cluster_world')
    st.nodes(st.pick(('inter_cluster', 'inter_cluster_world')), 'hyper_node_model_cross', 'hyper')
    st.edges('intra_cluster_to_mesh')
    st.nodes([st.new['intra_cluster_to_mesh']], 'node_model_down', 'mesh')
    return MultiGraph(st.nf, list(st.new.values()))


def multiscale_block(sd, bp, graph, aggregator, set_order=None):
    """MultiScaleGraphNet.forward multiscalegraphnet.py:20-63."""
    st = _Stages(sd, bp, graph, aggregator, set_order)
    st.edges('mesh_edges'); st.edges('world_edges')
    st.nodes(st.pick(('mesh_edges', 'world_edges')), 'node_model_cross', 'mesh')
    st.edges('intra_cluster_to_cluster')
    st.nodes([st.new['intra_cluster_to_cluster']], 'hyper_node_model_up', 'hyper')
    for i in range(3):
        st.edges('inter_cluster'); st.edges('inter_cluster_world')
        st.nodes(st.pick(('inter_cluster', 'inter_cluster_world')), f'hyper_node_models_cross.{i}', 'hyper')
    st.edges('intra_cluster_to_mesh')
    st.nodes([st.new['intra_cluster_to_mesh']], 'node_model_down', 'mesh')
    # stage 5 re-reads the *input* graph's mesh/world edge sets (perform_edge_updates filters graph.edge_sets)
    st.edges('mesh_edges'); st.edges('world_edges')
    st.nodes(st.pick(('mesh_edges', 'world_edges')), 'node_model_cross', 'mesh')
    return MultiGraph(st.nf, list(st.new.values()))


BLOCKS = {'hyper': (hyper_block, True), 'multiscale': (multiscale_block, True), 'hetero': (hetero_block, True),
          'multi': (graphnet_block, False), 'repeated': (repeated_block, False)}   # meshgraphnet.py:62-89


# --------------------------------------------------------------------------------------------------------
# a5: encoder / processor / decoder / MeshGraphNet
# --------------------------------------------------------------------------------------------------------
def encoder(sd, graph: MultiGraph, hierarchical: bool) -> MultiGraph:
    """encoder.py:24-47."""
    lat = [mlp(sd, 'encoder.node_model', graph.node_features[0])]
    if len(graph.node_features) > 1:
        lat.append(mlp(sd, 'encoder.hyper_node_model' if hierarchical else 'encoder.node_model',
                       graph.node_features[1]))
    sets = []
    for es in graph.edge_sets:
        pfx = f'encoder.edge_models.{es.name}'
        if (pfx + '.0.layers.linear_0.weight') not in sd:                      # KeyError -> dropped :44-45
            continue
        sets.append(es._replace(features=mlp(sd, pfx, es.features)))
    return MultiGraph(lat, sets)


def mesh_graph_net(sd: Dict[str, torch.Tensor], graph: MultiGraph, architecture: str, aggregator: str,
                   set_order: Optional[Sequence[str]] = None) -> torch.Tensor:
    """MeshGraphNet.forward meshgraphnet.py:46-51.  ``message_passing_steps`` is read off the state_dict."""
    block, hierarchical = BLOCKS.get(architecture, (graphnet_block, False))
    g = encoder(sd, graph, hierarchical)
    steps = 0
    while any(k.startswith(f'processor.graphnet_blocks.{steps}.') for k in sd):
        steps += 1
    for l in range(steps):
        bp = f'processor.graphnet_blocks.{l}'
        if block in (hyper_block, multiscale_block):
            g = block(sd, bp, g, aggregator, set_order)
        else:
            g = block(sd, bp, g, aggregator)
    return mlp(sd, 'decoder.model', g.node_features[0], layer_norm=False)      # decoder.py:15-16


# --------------------------------------------------------------------------------------------------------
# a7: Normalizer (normalizer.py:9-75), functional-state restatement
# --------------------------------------------------------------------------------------------------------
class Normalizer:
    def __init__(self, size: int, max_accumulations: int = 10 ** 6, std_epsilon: float = 1e-8,
                 dtype=torch.float32):
        self.max_acc = max_accumulations
        self.eps = torch.tensor([std_epsilon], dtype=dtype)
        self.acc_count = torch.zeros(1, dtype=dtype)
        self.num_acc = torch.zeros(1, dtype=dtype)
        self.acc_sum = torch.zeros(size, dtype=dtype)
        self.acc_sum_sq = torch.zeros(size, dtype=dtype)

    def mean(self):
        return self.acc_sum / torch.clamp(self.acc_count, min=1.0)            # normalizer.py:64-66

    def std(self):
        safe = torch.clamp(self.acc_count, min=1.0)
        s = torch.sqrt(torch.abs(self.acc_sum_sq / safe - self.mean() ** 2))  # normalizer.py:68-71
        return torch.maximum(s, self.eps)

    def __call__(self, x, accumulate=True):
        if accumulate and float(self.num_acc) < self.max_acc:                 # normalizer.py:43-46
            self.acc_sum = self.acc_sum + x.sum(0)
            self.acc_sum_sq = self.acc_sum_sq + (x ** 2).sum(0)
            self.acc_count = self.acc_count + float(x.shape[0])
            self.num_acc = self.num_acc + 1.0
        return (x - self.mean()) / self.std()

    def inverse(self, y):
        return y * self.std() + self.mean()                                   # normalizer.py:48-50


# --------------------------------------------------------------------------------------------------------
# f1 ("next" row): MeshSimulator._get_batched index mapping (MeshSimulator.py:159-234)
# --------------------------------------------------------------------------------------------------------
def batch_graphs(graphs: Sequence[MultiGraph], reference_compat: bool = False) -> MultiGraph:
    """Disjoint union of graphs that share one topology signature.

    ``reference_compat=True`` reproduces the reference's index arithmetic *including* its mis-mapping of
    hyper-node ids for batch sizes >= 2: an id ``x`` is treated as a hyper id only if
    ``x >= batch_size * num_nodes`` (MeshSimulator.py:196,206-208).  The default maps hyper ids correctly.
    """
    B = len(graphs)
    names = [e.name for e in graphs[0].edge_sets]
    feats = {n: [] for n in names}
    snd = {n: [] for n in names}
    rcv = {n: [] for n in names}
    for i, g in enumerate(graphs):
        n_mesh = g.node_features[0].shape[0]
        n_hyp = g.node_features[1].shape[0] if len(g.node_features) > 1 else 0
        thresh = B * n_mesh if reference_compat else n_mesh
        for e in g.edge_sets:
            def remap(idx):
                idx = idx.long()
                if reference_compat:
                    hyp = idx + (B - 1) * n_mesh + i * n_hyp
                else:
                    hyp = (idx - n_mesh) + B * n_mesh + i * n_hyp
                return torch.where(idx < thresh, idx + i * n_mesh, hyp)
            feats[e.name].append(e.features)
            snd[e.name].append(remap(e.senders))
            rcv[e.name].append(remap(e.receivers))
    nodes = [torch.cat(x, 0) for x in zip(*[g.node_features for g in graphs])]
    return MultiGraph(nodes, [EdgeSet(n, torch.cat(feats[n], 0), torch.cat(snd[n], 0), torch.cat(rcv[n], 0))
                              for n in names])


# --------------------------------------------------------------------------------------------------------
# loss used by bench / training parity: FlagModel.training_step flag.py:146-154
# --------------------------------------------------------------------------------------------------------
def masked_mse(pred: torch.Tensor, target: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    return F.mse_loss(target[mask], pred[mask])


def _name_hash(s: str) -> int:
    h = 0
    for c in s:
        h = (h * 131 + ord(c)) % (1 << 31)
    return h


def init_state_dict_like(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, dtype=torch.float32):
    """Deterministic weights for a given {reference key name: shape} map.  Linear weights/biases are
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the nn.Linear default family); LayerNorm affine parameters are
    perturbed away from (1, 0) so that their gradients are exercised.  Each tensor has its own generator
    seeded from (seed, name), so the result does not depend on dict order."""
    sd = {}
    for name, shape in shapes.items():
        g = torch.Generator().manual_seed((seed * 1000003 + _name_hash(name)) % (1 << 31))
        is_ln = re.search(r'\.1\.(weight|bias)$', name) is not None
        if is_ln:
            t = 0.1 * torch.randn(shape, generator=g)
            if name.endswith('weight'):
                t = t + 1.0
        else:
            if name.endswith('weight'):
                fan_in = shape[1]
            else:
                wshape = shapes[name[:-4] + 'weight']
                fan_in = wshape[1]
            t = (torch.rand(shape, generator=g) * 2 - 1) / (fan_in ** 0.5)
        sd[name] = t.to(dtype)
    return sd


def param_shapes(architecture: str, aggregator: str, steps: int, edge_sets: Sequence[str], node_in: int,
                 edge_in: Dict[str, int], hyper_in: int = 0, out_size: int = 3, latent: int = 128,
                 n_sets_node: Optional[Dict[str, int]] = None) -> Dict[str, Tuple[int, ...]]:
    """{reference state_dict key: shape} for a MeshGraphNet (meshgraphnet.py:24-44).
    ``n_sets_node[model]`` = number of edge sets whose aggregates feed that node model (defaults: the sets
    the reference block would aggregate when every registered set is present in the graph)."""
    sh: Dict[str, Tuple[int, ...]] = collections.OrderedDict()
    k = 4 if aggregator == 'pna' else 1

    def make(prefix, fan_in, out, ln=True):
        base = prefix + ('.0.layers.' if ln else '.layers.')
        for i, (a, b) in enumerate(((fan_in, latent), (latent, latent), (latent, out))):
            sh[f'{base}linear_{i}.weight'] = (b, a)
            sh[f'{base}linear_{i}.bias'] = (b,)
        if ln:
            sh[prefix + '.1.weight'] = (out,)
            sh[prefix + '.1.bias'] = (out,)

    hierarchical = BLOCKS.get(architecture, (None, False))[1]
    make('encoder.node_model', node_in, latent)
    for n in edge_sets:
        make(f'encoder.edge_models.{n}', edge_in[n], latent)
    if hierarchical:
        make('encoder.hyper_node_model', hyper_in, latent)
    S = len(edge_sets)
    nd = n_sets_node or {}
    for l in range(steps):
        bp = f'processor.graphnet_blocks.{l}'
        for n in edge_sets:
            make(f'{bp}.edge_models.{n}', 3 * latent, latent)
        if architecture in ('hyper', 'multiscale'):
            n_cross = nd.get('node_model_cross', sum(n in edge_sets for n in ('mesh_edges', 'world_edges')))
            n_hcross = nd.get('hyper_node_model_cross',
                              sum(n in edge_sets for n in ('inter_cluster', 'inter_cluster_world')))
            make(f'{bp}.node_model_cross', latent * (1 + k * n_cross), latent)
            make(f'{bp}.hyper_node_model_up', latent * (1 + k), latent)
            if architecture == 'hyper':
                make(f'{bp}.hyper_node_model_cross', latent * (1 + k * n_hcross), latent)
            else:
                for i in range(3):
                    make(f'{bp}.hyper_node_models_cross.{i}', latent * (1 + k * n_hcross), latent)
            make(f'{bp}.node_model_down', latent * (1 + k), latent)
        else:
            make(f'{bp}.node_model_cross', latent * (1 + k * nd.get('node_model_cross', S)), latent)
            if architecture == 'hetero':
                make(f'{bp}.hyper_node_model_cross', latent * (1 + k * nd.get('hyper_node_model_cross', S)),
                     latent)
    make('decoder.model', latent, out_size, ln=False)
    return sh
