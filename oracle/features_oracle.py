"""CPU oracle for the rows in front of the message-passing path: frame -> graph features
and the remote-graph assembly over a given clustering (row f3).

TEST INFRASTRUCTURE ONLY (same rule as mgn_oracle.py: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it).  Plain PyTorch on CPU, dtype-generic (fp32 restates the
reference's arithmetic, fp64 gives the reference value to compare fp32 implementations against).  Each function
cites the reference lines it follows (paths relative to /root/reference/src).

Pinning: tests/golden/feat_*.pt, produced by running the reference's own FlagModel / CylinderModel /
RemoteMessagePassing / util.triangles_to_edges in the build container (generator tests/golden/gen_golden_features.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from .mgn_oracle import EdgeSet, MultiGraph, Normalizer, segment_reduce


# --------------------------------------------------------------------------------------------------------
# util.py:50-89
# --------------------------------------------------------------------------------------------------------
def triangles_to_edges(cells: torch.Tensor, deform: bool = False):
    n = 4 if deform else 3
    pairs = [(i, (i + 1) % n) for i in range(n)]                      # (0,1),(1,2),(2,0) | (0,1),(1,2),(2,3),(3,0)
    e = torch.cat([torch.stack((cells[:, a], cells[:, b]), 1) for a, b in pairs], 0)
    lo, hi = e.min(1).values, e.max(1).values
    keys = sorted(set(zip(hi.tolist(), lo.tolist())))                 # torch.unique(dim=0): lexicographic rows
    s = torch.tensor([k[0] for k in keys], dtype=torch.int64)
    r = torch.tensor([k[1] for k in keys], dtype=torch.int64)
    return torch.cat([s, r]), torch.cat([r, s])                       # 'two_way_connectivity' (util.py:68)


def rel_features(world: torch.Tensor, mesh: Optional[torch.Tensor], s: torch.Tensor, r: torch.Tensor):
    """flag.py:80-93 / abstract_connector.py:90-95 / cylinder.py:83-87."""
    rw = world[s] - world[r]
    cols = [rw, torch.sqrt(rw.pow(2).sum(-1, keepdim=True))]
    if mesh is not None:
        rm = mesh[s] - mesh[r]
        cols += [rm, torch.sqrt(rm.pow(2).sum(-1, keepdim=True))]
    return torch.cat(cols, -1)


# --------------------------------------------------------------------------------------------------------
# model/flag.py
# --------------------------------------------------------------------------------------------------------
class FlagFeatures:
    """The normalisers FlagModel owns (flag.py:26-32) and its frame -> graph functions."""

    def __init__(self, dtype=torch.float32):
        self.dtype = dtype
        self.output = Normalizer(3, dtype=dtype)
        self.node = Normalizer(5, dtype=dtype)
        self.node_dynamic = Normalizer(1, dtype=dtype)
        self.mesh_edge = Normalizer(7, dtype=dtype)
        self.intra_edge = Normalizer(7, dtype=dtype)
        self.inter_edge = Normalizer(7, dtype=dtype)
        self.hyper_node = Normalizer(3, dtype=dtype)

    def build_graph(self, inputs: Dict[str, torch.Tensor], is_training: bool) -> dict:
        """flag.py:65-128."""
        dt = self.dtype
        world, prev, mesh = inputs['world_pos'].to(dt), inputs['prev|world_pos'].to(dt), inputs['mesh_pos'].to(dt)
        node_type = inputs['node_type']
        velocity = world - prev
        cls = (node_type[:, 0] != 0).long()
        one_hot = torch.nn.functional.one_hot(cls, 2).to(dt)         # flag.py:72-73 (HANDLE nodes always present)
        node_features = torch.cat((velocity, one_hot), -1)
        s, r = triangles_to_edges(inputs['cells'])
        edge_features = rel_features(world, mesh, s, r)
        length = torch.sqrt((world[s] - world[r]).pow(2).sum(-1))
        N = node_type.shape[0]
        mx = segment_reduce(length, r, N, 'max')                     # flag.py:101-113 (1-D data)
        mn = segment_reduce(length, r, N, 'min')
        edges_n = self.mesh_edge(edge_features, is_training)         # call order as in flag.py:95-115
        node_dynamic = self.node_dynamic(mx - mn)                    # accumulate defaults to True (flag.py:115)
        nodes_n = self.node(node_features, is_training)
        return {'node_features': [nodes_n], 'edge_sets': [EdgeSet('mesh_edges', edges_n, s, r)],
                'target_feature': world, 'mesh_features': mesh, 'node_dynamic': node_dynamic,
                'unnormalized_edges': EdgeSet('mesh_edges', edge_features, s, r)}

    def get_target(self, frame, is_training=True):
        """flag.py:182-190."""
        dt = self.dtype
        cur, prev, tgt = frame['world_pos'].to(dt), frame['prev|world_pos'].to(dt), frame['target|world_pos'].to(dt)
        return self.output(tgt - 2 * cur + prev, is_training)

    def update(self, inputs, net_out):
        """flag.py:169-180."""
        acc = self.output.inverse(net_out.to(self.dtype))
        return 2 * inputs['world_pos'].to(self.dtype) + acc - inputs['prev|world_pos'].to(self.dtype)


# --------------------------------------------------------------------------------------------------------
# model/cylinder.py
# --------------------------------------------------------------------------------------------------------
class CylinderFeatures:
    def __init__(self, dtype=torch.float32):
        self.dtype = dtype
        self.output = Normalizer(3, dtype=dtype)
        self.node = Normalizer(6, dtype=dtype)
        self.mesh_edge = Normalizer(3, dtype=dtype)

    def build_graph(self, inputs, is_training: bool) -> dict:
        """cylinder.py:65-106."""
        dt = self.dtype
        velocity, mesh = inputs['velocity'].to(dt), inputs['mesh_pos'].to(dt)
        t = inputs['node_type'][:, 0].long().clone()
        t[t == 4] = 1
        t[t == 5] = 2
        t[t == 6] = 3                                                # cylinder.py:71-74
        node_features = torch.cat((velocity, torch.nn.functional.one_hot(t, 4).to(dt)), -1)
        s, r = triangles_to_edges(inputs['cells'])
        edge_features = rel_features(mesh, None, s, r)
        edges_n = self.mesh_edge(edge_features, is_training)
        nodes_n = self.node(node_features, is_training)
        return {'node_features': [nodes_n], 'edge_sets': [EdgeSet('mesh_edges', edges_n, s, r)],
                'target_feature': velocity, 'mesh_features': mesh,
                'unnormalized_edges': EdgeSet('mesh_edges', edge_features, s, r)}

    def get_target(self, frame, is_training=True):
        """cylinder.py:167-173."""
        dt = self.dtype
        dv = frame['target|velocity'].to(dt) - frame['velocity'].to(dt)
        return self.output(torch.cat((dv, frame['pressure'].to(dt)), 1), is_training)

    def update(self, inputs, net_out):
        """cylinder.py:155-165."""
        o = self.output.inverse(net_out.to(self.dtype))
        return inputs['velocity'].to(self.dtype) + o[:, :2], o[:, 2:]


# --------------------------------------------------------------------------------------------------------
# model/plate.py
# --------------------------------------------------------------------------------------------------------
class PlateFeatures:
    RADIUS = 0.03                                                     # plate.py:85

    def __init__(self, dtype=torch.float32):
        self.dtype = dtype
        self.output = Normalizer(3, dtype=dtype)
        self.node = Normalizer(6, dtype=dtype)
        self.mesh_edge = Normalizer(8, dtype=dtype)
        self.world_edge = Normalizer(4, dtype=dtype)
        self.intra_edge = Normalizer(8, dtype=dtype)
        self.inter_edge = Normalizer(8, dtype=dtype)
        self.hyper_node = Normalizer(3, dtype=dtype)

    def build_graph(self, inputs, is_training: bool) -> dict:
        """plate.py:69-200.  Distances by the direct difference formula (the reference's cdist uses the matrix-product
        form above 25 points; both agree unless a pair sits within fp32 noise of the radius)."""
        dt = self.dtype
        world, mesh, target = inputs['world_pos'].to(dt), inputs['mesh_pos'].to(dt), inputs['target|world_pos'].to(dt)
        raw = inputs['node_type'][:, 0].long()
        t = raw.clone()
        t[t == 3] = 2                                                 # plate.py:78
        one_hot = torch.nn.functional.one_hot(t, 3).to(dt)
        s, r = triangles_to_edges(inputs['cells'], deform=True)
        dist = torch.sqrt((world[:, None, :] - world[None, :, :]).pow(2).sum(-1))
        conn = dist < self.RADIUS                                     # plate.py:86-88
        conn.fill_diagonal_(False)
        conn[s, r] = False                                            # :91
        conn[raw != 1, :] = False                                     # :96-97 only OBSTACLE senders
        conn[:, raw != 0] = False                                     # :105-106 only NORMAL receivers
        ws, wr = torch.nonzero(conn, as_tuple=True)
        world_feat = rel_features(world, None, ws, wr)                # :137-140
        mesh_feat = rel_features(world, mesh, s, r)                   # :165-173
        world_n = self.world_edge(world_feat, is_training)
        mesh_n = self.mesh_edge(mesh_feat, is_training)
        vel = torch.zeros(world.shape[0], 3, dtype=dt)
        obst = raw == 1
        vel[obst] = target[obst] - world[obst]                        # :190-194
        node_features = torch.cat((one_hot, vel), -1)
        nodes_n = self.node(node_features, is_training)
        return {'node_features': [nodes_n],
                'edge_sets': [EdgeSet('mesh_edges', mesh_n, s, r), EdgeSet('world_edges', world_n, ws, wr)],
                'target_feature': world, 'mesh_features': mesh, 'obstacle_nodes': obst,
                'unnormalized_edges': EdgeSet('mesh_edges', mesh_feat, s, r)}

    def get_target(self, frame, is_training=True):
        """plate.py:259-264."""
        return self.output(frame['target|world_pos'].to(self.dtype) - frame['world_pos'].to(self.dtype), is_training)

    def update(self, inputs, net_out):
        """plate.py:246-257."""
        v = self.output.inverse(net_out.to(self.dtype))
        return inputs['world_pos'].to(self.dtype) + v


# --------------------------------------------------------------------------------------------------------
# rmp/: neighbouring clusters and the hierarchical connector
# --------------------------------------------------------------------------------------------------------
def neighboring_clusters(senders: torch.Tensor, receivers: torch.Tensor, labels: Sequence[int]) -> List[tuple]:
    """abstract_clustering_algorithm.py:124-145: the set of unordered label pairs joined by a mesh edge.  The
    reference's list order comes from Python set iteration; here the pairs are returned sorted (order of the edges
    inside an edge set does not change the model's result beyond fp summation order)."""
    lab = torch.as_tensor(list(labels))
    a, b = lab[senders], lab[receivers]
    keep = a != b
    return sorted({(min(int(x), int(y)), max(int(x), int(y))) for x, y in zip(a[keep].tolist(), b[keep].tolist())})


def hierarchical_connect(graph: dict, clusters: Sequence[torch.Tensor], neighbors: Sequence, intra: Normalizer,
                         inter: Normalizer, hyper: Normalizer, is_training: bool, hyper_node_features: bool = True,
                         fully_connect: bool = False, noise: Optional[torch.Tensor] = None) -> MultiGraph:
    """hierarchical_connector.py:27-143 on a graph from ``build_graph`` (node_features = graph['node_features'][0],
    remote_message_passing.py:67).  ``noise`` (if given) is the sample the reference draws at :48-51."""
    world, mesh = graph['target_feature'], graph['mesh_features']
    cf = torch.cat((world, mesh), 1)                                             # :29
    nf = graph['node_features'][0]
    N, K = nf.shape[0], len(clusters)
    means = torch.stack([cf[c].mean(0) for c in clusters])                        # :39-43
    if noise is not None:
        means = means + noise
    nf_means = torch.stack([nf[c].mean(0) for c in clusters])                     # :44-45,53
    if hyper_node_features:                                                       # :54-71
        spread_mesh = torch.stack([torch.sqrt((means[i][-3:] - cf[c][:, -3:]).pow(2).sum(1)).max()
                                   for i, c in enumerate(clusters)])
        spread_world = torch.stack([torch.sqrt((means[i][:3] - cf[c][:, :3]).pow(2).sum(1)).max()
                                    for i, c in enumerate(clusters)])
        sizes = torch.tensor([len(c) for c in clusters]).to(cf.dtype)
        aug = hyper(torch.stack([sizes, spread_mesh, spread_world], -1), is_training)
        nf_means = torch.cat([nf_means, aug], -1)
    tf = torch.cat([cf, means], 0)                                                # _get_subgraph: cat of the list

    def sub(s, r):                                                                # abstract_connector.py:84-98
        s2, r2 = torch.cat((s, r)), torch.cat((r, s))
        d = tf[s2] - tf[r2]
        w, m = d[:, :3], d[:, 3:]
        return s2, r2, torch.cat((w, torch.sqrt(w.pow(2).sum(-1, keepdim=True)), m,
                                  torch.sqrt(m.pow(2).sum(-1, keepdim=True))), -1)
    to_mesh, to_cluster = [], []
    for k, c in enumerate(clusters):                                              # :85-100
        h = torch.full((len(c),), N + k, dtype=torch.int64)
        s, r, f = sub(h, c.long())
        n = len(c)
        to_mesh.append((s[:n], r[:n], f[:n]))
        to_cluster.append((s[n:], r[n:], f[n:]))

    def cat(parts, name):
        feats = intra(torch.cat([p[2] for p in parts]), is_training)
        return EdgeSet(name, feats, torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]))
    e_cluster = cat(to_cluster, 'intra_cluster_to_cluster')                       # :102-113 (normalised first)
    e_mesh = cat(to_mesh, 'intra_cluster_to_mesh')                                # :115-125
    if fully_connect or K < 4:                                                    # :128-129, :207-212
        idx = torch.combinations(torch.arange(N, N + K), with_replacement=True)
        idx = idx[idx[:, 0] != idx[:, 1]]
        s, r = idx[:, 0], idx[:, 1]
    else:                                                                         # :132, :201-205
        nb = torch.tensor([list(p) for p in neighbors], dtype=torch.int64).reshape(-1, 2) + N
        s, r = nb[:, 0], nb[:, 1]
    s2, r2, f = sub(s, r)
    e_inter = EdgeSet('inter_cluster', inter(f, is_training), s2, r2)             # :134-138
    return MultiGraph([nf, nf_means], list(graph['edge_sets']) + [e_cluster, e_mesh, e_inter])


def multigraph_connect(graph: dict, clusters, neighbors, intra: Normalizer, inter: Normalizer, hyper: Normalizer,
                       is_training: bool, hyper_node_features: bool = False) -> MultiGraph:
    """multigraph_connector.py:23-86: hierarchical expansion, then one-hot tags and everything merged into 'mesh_edges'."""
    g = hierarchical_connect(graph, clusters, neighbors, intra, inter, hyper, is_training, hyper_node_features)
    nf, hnf = g.node_features

    def tag(x, k, n):
        t = torch.zeros(x.shape[0], n, dtype=x.dtype)
        t[:, k] = 1
        return torch.cat((x, t), 1)
    by = {e.name: e for e in g.edge_sets}
    parts = [by[n] for n in ('mesh_edges', 'inter_cluster', 'intra_cluster_to_cluster', 'intra_cluster_to_mesh')]
    merged = EdgeSet('mesh_edges', torch.cat([tag(e.features, k, 4) for k, e in enumerate(parts)]),
                     torch.cat([e.senders for e in parts]), torch.cat([e.receivers for e in parts]))
    return MultiGraph([tag(nf, 0, 2), tag(hnf, 1, 2)], [merged, by['world_edges']])


# --------------------------------------------------------------------------------------------------------
# rollout inference: model/flag.py:194-260, cylinder.py:175-245, plate.py:264-347.
# Pinned by tests/golden/rollout_*.pt (generator tests/golden/gen_golden_rollout.py runs the reference's own rollout).
# `net(graph: MultiGraph) -> [N, out]` is the learned model (mgn_oracle.mesh_graph_net with a state_dict bound);
# `expand(graph_dict, step, is_training) -> MultiGraph` stands for expand_graph (remote message passing over a GIVEN
# clustering: the labels come from scikit-learn in the reference and are an input here); None -> the mesh graph as is.
# --------------------------------------------------------------------------------------------------------
def _as_multigraph(g: dict) -> MultiGraph:
    return MultiGraph(list(g['node_features']), list(g['edge_sets']))


def _first_frame(trajectory: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {k: torch.squeeze(v, 0)[0] for k, v in trajectory.items()}               # flag.py:198


def _per_step_mse(truth: torch.Tensor, pred: torch.Tensor) -> torch.Tensor:
    return ((truth - pred) ** 2).mean(-1).mean(-1)                                   # flag.py:221-223


def flag_rollout(ff: FlagFeatures, net, trajectory, num_steps, expand=None):
    """flag.py:194-246: the recorded trajectory holds the state BEFORE each step; HANDLE nodes keep their position."""
    num_steps = trajectory['cells'].shape[0] if num_steps is None else num_steps
    start = _first_frame(trajectory)
    mask = (start['node_type'][:, 0] == 0).unsqueeze(1).expand(-1, 3)                # NodeType.NORMAL
    prev, cur, visited = start['prev|world_pos'].to(ff.dtype), start['world_pos'].to(ff.dtype), []
    for step in range(num_steps):
        frame = {**start, 'prev|world_pos': prev, 'world_pos': cur}
        g = ff.build_graph(frame, False)
        graph = expand(g, step, False) if expand is not None else _as_multigraph(g)
        pred = ff.update(frame, net(graph))
        visited.append(cur)
        prev, cur = cur, torch.where(mask, pred, cur)
    predictions = torch.stack(visited)
    return predictions, _per_step_mse(trajectory['world_pos'][:num_steps].to(ff.dtype), predictions)


def cylinder_rollout(cf: CylinderFeatures, net, trajectory, num_steps=None, expand=None):
    """cylinder.py:175-232: `num_steps` is overwritten by the trajectory length (:178); NORMAL and OUTFLOW nodes move; the
    recorded trajectory holds the state AFTER each step."""
    start = _first_frame(trajectory)
    num_steps = trajectory['cells'].shape[0]
    t = start['node_type'][:, 0]
    mask = ((t == 0) | (t == 5)).unsqueeze(1).expand(-1, 2)
    velocity, pressure = start['velocity'].to(cf.dtype), start['pressure'].to(cf.dtype)
    vel_traj, pr_traj = [], []
    for step in range(num_steps):
        frame = {**start, 'velocity': velocity, 'pressure': pressure}
        g = cf.build_graph(frame, False)
        graph = expand(g, step, False) if expand is not None else _as_multigraph(g)
        pred, pressure = cf.update(frame, net(graph))
        velocity = torch.where(mask, pred, velocity)
        vel_traj.append(velocity)
        pr_traj.append(pressure)
    predictions = torch.stack(vel_traj)
    return predictions, torch.stack(pr_traj), _per_step_mse(trajectory['velocity'][:num_steps].to(cf.dtype), predictions)


def plate_rollout(pf: PlateFeatures, net, trajectory, num_steps, expand=None):
    """plate.py:264-334: NORMAL nodes are integrated, every other node is set to the scripted target position of the step."""
    num_steps = trajectory['cells'].shape[0] if num_steps is None else num_steps
    start = _first_frame(trajectory)
    mask = (start['node_type'][:, 0] == 0).unsqueeze(1).expand(-1, 3)
    cur = start['world_pos'].to(pf.dtype)
    targets = trajectory['target|world_pos'].to(pf.dtype)
    visited = []
    for step in range(num_steps):
        frame = {**start, 'world_pos': cur, 'target|world_pos': targets[step]}
        g = pf.build_graph(frame, False)
        graph = expand(g, step, False) if expand is not None else _as_multigraph(g)
        pred = pf.update(frame, net(graph))
        cur = torch.where(mask, pred, targets[step])
        visited.append(cur)
    predictions = torch.stack(visited)
    return predictions, _per_step_mse(trajectory['world_pos'][:num_steps].to(pf.dtype), predictions)


def n_step_computation(rollout_fn, trajectory, n_step: int, num_timesteps=None):
    """flag.py:248-260 (identical in cylinder.py / plate.py): every window of n_step + 1 frames is rolled out from its first
    frame; `rollout_fn(window, n_step + 1)` returns (..., per-step mse) with the mse LAST."""
    frames = trajectory['cells'].shape[0] if num_timesteps is None else num_timesteps
    means, lasts = [], []
    for start in range(frames - n_step):
        window = {k: v[start:start + n_step + 1] for k, v in trajectory.items()}
        mse = rollout_fn(window, n_step + 1)[-1]
        means.append(mse.mean())
        lasts.append(mse[-1])
    return torch.stack(means).mean(), torch.stack(lasts).mean()
