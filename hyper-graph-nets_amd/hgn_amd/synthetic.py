"""Synthetic mesh graphs of the reference's dataset shapes (datasets are not shipped with the reference:
``data/*/input`` is empty).  Pure host-side tensor construction, no kernels.

``grid_graph`` builds a triangulated nx x ny grid the way ``FlagModel.build_graph`` sees a flag_simple frame
(/root/reference src/model/flag.py:65-128): 5-dim node features (velocity(3) + one-hot type(2)), two-way mesh
edges from the triangle rule of src/util.py:50-70, 7-dim edge features (relative world pos(3) + norm +
relative mesh pos(2) + norm).  Optional extras mirror the edge sets the remote-message-passing and
graph-balancer layers emit (hierarchical_connector.py:85-137, abstract_graph_balancer.py:48-63).
"""
import collections

import torch

EdgeSet = collections.namedtuple('EdgeSet', ['name', 'features', 'senders', 'receivers'])
MultiGraph = collections.namedtuple('MultiGraph', ['node_features', 'edge_sets'])


def grid_triangles(nx: int, ny: int) -> torch.Tensor:
    ii, jj = torch.meshgrid(torch.arange(nx - 1), torch.arange(ny - 1), indexing='ij')
    a = (ii * ny + jj).flatten()
    b = a + 1
    c = a + ny
    d = c + 1
    return torch.cat([torch.stack([a, b, d], 1), torch.stack([a, d, c], 1)], 0)


def two_way_edges(faces: torch.Tensor):
    """Unique undirected edges of the triangles, then both directions (src/util.py:50-70 semantics)."""
    e = torch.cat([faces[:, 0:2], faces[:, 1:3], torch.stack([faces[:, 2], faces[:, 0]], 1)], 0)
    lo, hi = e.min(1).values, e.max(1).values
    uniq = torch.unique(torch.stack([hi, lo], 1), dim=0)
    s, r = uniq[:, 0].long(), uniq[:, 1].long()
    return torch.cat([s, r]), torch.cat([r, s])


def _rel_features(pos_w, pos_m, s, r):
    rw = pos_w[s] - pos_w[r]
    rm = pos_m[s] - pos_m[r]
    return torch.cat([rw, rw.norm(dim=-1, keepdim=True), rm, rm.norm(dim=-1, keepdim=True)], -1)


def grid_graph(seed: int = 0, nx: int = 40, ny: int = 40, clusters: int = 0, balance: int = 0, world: int = 0,
               normalise: bool = True, dtype=torch.float32) -> MultiGraph:
    g = torch.Generator().manual_seed(seed)
    N = nx * ny
    xs = torch.linspace(0, 3, nx)
    ys = torch.linspace(0, 2, ny)
    mesh_pos = torch.stack(torch.meshgrid(xs, ys, indexing='ij'), -1).reshape(N, 2)
    world_pos = torch.cat([mesh_pos, 0.1 * torch.randn(N, 1, generator=g)], -1)
    prev = world_pos + 0.01 * torch.randn(N, 3, generator=g)
    node_type = torch.zeros(N, dtype=torch.long)
    node_type[:3] = 1                                    # HANDLE nodes -> one-hot column 1 (flag.py:72-73)
    node_feat = torch.cat([world_pos - prev, torch.nn.functional.one_hot(node_type, 2).float()], -1)
    s, r = two_way_edges(grid_triangles(nx, ny))
    mesh_feat = _rel_features(world_pos, mesh_pos, s, r)

    def norm(x):
        if not normalise:
            return x
        return (x - x.mean(0)) / x.std(0).clamp(min=1e-8)

    sets = [EdgeSet('mesh_edges', norm(mesh_feat).to(dtype), s, r)]
    nodes = [norm(node_feat).to(dtype)]
    if balance:
        bs = torch.randint(0, N, (balance,), generator=g)
        br = (bs + 1 + torch.randint(0, N - 1, (balance,), generator=g)) % N
        s2, r2 = torch.cat([bs, br]), torch.cat([br, bs])
        sets.append(EdgeSet('balance', norm(_rel_features(world_pos, mesh_pos, s2, r2)).to(dtype), s2, r2))
    if world:
        ws = torch.randint(0, N, (world,), generator=g)
        wr = (ws + 1 + torch.randint(0, N - 1, (world,), generator=g)) % N
        s2, r2 = torch.cat([ws, wr]), torch.cat([wr, ws])
        rw = world_pos[s2] - world_pos[r2]
        sets.append(EdgeSet('world_edges', norm(torch.cat([rw, rw.norm(dim=-1, keepdim=True)], -1)).to(dtype),
                            s2, r2))
    if clusters:
        K = clusters
        # contiguous blocks along x, like a spectral clustering of a strip would give
        lab = torch.div(torch.arange(N) // ny * K, nx, rounding_mode='floor').clamp(max=K - 1)
        pos5 = torch.cat([world_pos, mesh_pos], -1)
        cm = torch.stack([pos5[lab == c].mean(0) for c in range(K)])
        nf = torch.stack([node_feat[lab == c].mean(0) for c in range(K)])
        extra = torch.stack([torch.stack([(lab == c).sum().float(),
                                          (pos5[lab == c][:, 3:] - cm[c, 3:]).norm(dim=-1).max(),
                                          (pos5[lab == c][:, :3] - cm[c, :3]).norm(dim=-1).max()])
                             for c in range(K)])
        nodes.append(norm(torch.cat([nf, extra], -1)).to(dtype))
        ids = torch.arange(N)
        hyp = N + lab

        def rel5(a, b):
            d = a - b
            return torch.cat([d[:, :3], d[:, :3].norm(dim=-1, keepdim=True), d[:, 3:],
                              d[:, 3:].norm(dim=-1, keepdim=True)], -1)
        down = EdgeSet('intra_cluster_to_mesh', norm(rel5(cm[lab], pos5)).to(dtype), hyp, ids)
        up = EdgeSet('intra_cluster_to_cluster', norm(rel5(pos5, cm[lab])).to(dtype), ids, hyp)
        a = torch.arange(K - 1)
        cs, cr = torch.cat([a, a + 1]), torch.cat([a + 1, a])
        if K > 3:                                        # one skip connection so degrees differ
            cs, cr = torch.cat([cs, torch.tensor([0, K - 1])]), torch.cat([cr, torch.tensor([K - 1, 0])])
        inter = EdgeSet('inter_cluster', norm(rel5(cm[cs], cm[cr])).to(dtype), N + cs, N + cr)
        sets += [down, up, inter]
    return MultiGraph(nodes, sets)


def batch(graphs) -> MultiGraph:
    """Disjoint union with the *correct* hyper-node id mapping: mesh rows of all graphs
    first, then hyper rows of all graphs."""
    B = len(graphs)
    n_mesh = [g.node_features[0].shape[0] for g in graphs]
    has_h = len(graphs[0].node_features) > 1
    n_hyp = [g.node_features[1].shape[0] if has_h else 0 for g in graphs]
    mesh_off = [sum(n_mesh[:i]) for i in range(B)]
    hyp_off = [sum(n_mesh) + sum(n_hyp[:i]) for i in range(B)]
    names = [e.name for e in graphs[0].edge_sets]
    out = []
    for k, name in enumerate(names):
        fs, ss, rs = [], [], []
        for i, g in enumerate(graphs):
            e = g.edge_sets[k]

            def remap(idx):
                return torch.where(idx < n_mesh[i], idx + mesh_off[i], idx - n_mesh[i] + hyp_off[i])
            fs.append(e.features); ss.append(remap(e.senders)); rs.append(remap(e.receivers))
        out.append(EdgeSet(name, torch.cat(fs), torch.cat(ss), torch.cat(rs)))
    nodes = [torch.cat([g.node_features[j] for g in graphs]) for j in range(len(graphs[0].node_features))]
    return MultiGraph(nodes, out)


def flag_frame(seed: int = 0, nx: int = 40, ny: int = 40, dtype=torch.float32) -> dict:
    """One flag_simple-shape data frame as the reference's dataset hands it to ``FlagModel.build_graph``
    (src/model/flag.py:65-71): world/prev/target positions [N,3], mesh_pos [N,2], node_type [N,1], cells [F,3]."""
    g = torch.Generator().manual_seed(seed)
    N = nx * ny
    xs = torch.linspace(0, 3, nx)
    ys = torch.linspace(0, 2, ny)
    mesh_pos = torch.stack(torch.meshgrid(xs, ys, indexing='ij'), -1).reshape(N, 2).to(dtype)
    world_pos = torch.cat([mesh_pos, 0.1 * torch.randn(N, 1, generator=g, dtype=dtype)], -1)
    prev = world_pos + 0.01 * torch.randn(N, 3, generator=g, dtype=dtype)
    target = world_pos + 0.01 * torch.randn(N, 3, generator=g, dtype=dtype)
    node_type = torch.zeros(N, 1, dtype=torch.int32)
    node_type[:3] = 3                                    # HANDLE (src/util.py:27-35)
    return {'world_pos': world_pos, 'prev|world_pos': prev, 'target|world_pos': target, 'mesh_pos': mesh_pos,
            'node_type': node_type, 'cells': grid_triangles(nx, ny)}


def cylinder_frame(seed: int = 0, nx: int = 30, ny: int = 20, dtype=torch.float32) -> dict:
    """One cylinder_flow-shape frame (src/model/cylinder.py:65-87): velocity [N,2], mesh_pos [N,2], node types
    NORMAL / INFLOW(4) / OUTFLOW(5) / WALL_BOUNDARY(6)."""
    g = torch.Generator().manual_seed(seed)
    N = nx * ny
    xs = torch.linspace(0, 1.6, nx)
    ys = torch.linspace(0, 0.4, ny)
    mesh_pos = torch.stack(torch.meshgrid(xs, ys, indexing='ij'), -1).reshape(N, 2).to(dtype)
    velocity = torch.randn(N, 2, generator=g, dtype=dtype)
    target = velocity + 0.01 * torch.randn(N, 2, generator=g, dtype=dtype)
    node_type = torch.zeros(nx, ny, dtype=torch.int32)
    node_type[0, :] = 4
    node_type[-1, :] = 5
    node_type[:, 0] = 6
    node_type[:, -1] = 6
    return {'velocity': velocity, 'target|velocity': target, 'mesh_pos': mesh_pos,
            'node_type': node_type.reshape(N, 1), 'cells': grid_triangles(nx, ny),
            'pressure': torch.randn(N, 1, generator=g, dtype=dtype)}


def cylinder_remote_sets(nodes: torch.Tensor, pos: torch.Tensor, mesh_set: EdgeSet, K: int, n_balance: int, seed: int) -> MultiGraph:
    """BASELINE.json configs[4] shape on top of a cylinder_flow frame's mesh graph: a `balance` edge set (what the graph balancer
    appends, abstract_graph_balancer.py:48-63) and the three remote sets + hyper-node rows of the `hyper` connector
    (hierarchical_connector.py:85-137), K clusters as contiguous strips along x.  The reference cannot produce this combination
    (cylinder + any clustering or balancer fails in its normalisers), so the sets are built here from
    the frame's 2-D positions with the cylinder feature rule (relative mesh position + norm, cylinder.py:85-87).
    ``nodes`` / ``mesh_set``: the normalised node features [N,6] and mesh edge set of CylinderModel.build_graph (host)."""
    N = pos.shape[0]
    gen = torch.Generator().manual_seed(seed)

    def norm(x):
        return (x - x.mean(0)) / x.std(0).clamp(min=1e-8)

    def rel(a, b):
        d = a - b
        return torch.cat([d, d.norm(dim=-1, keepdim=True)], -1)
    lab = (pos[:, 0] / (pos[:, 0].max() + 1e-6) * K).long().clamp(max=K - 1)
    cm_pos = torch.stack([pos[lab == c].mean(0) for c in range(K)])
    hyper = torch.stack([torch.cat([nodes[lab == c].mean(0), torch.tensor([float((lab == c).sum())]),
                                    (pos[lab == c] - cm_pos[c]).norm(dim=-1).max().reshape(1)]) for c in range(K)])
    ids, hyp = torch.arange(N), N + lab
    a = torch.arange(K - 1)
    cs, cr = torch.cat([a, a + 1, torch.tensor([0, K - 1])]), torch.cat([a + 1, a, torch.tensor([K - 1, 0])])
    bs = torch.randint(0, N, (n_balance,), generator=gen)
    br = (bs + 1 + torch.randint(0, N - 1, (n_balance,), generator=gen)) % N
    b_s, b_r = torch.cat([bs, br]), torch.cat([br, bs])
    sets = [mesh_set,
            EdgeSet('balance', norm(rel(pos[b_s], pos[b_r])), b_s, b_r),
            EdgeSet('intra_cluster_to_mesh', norm(rel(cm_pos[lab], pos)), hyp, ids),
            EdgeSet('intra_cluster_to_cluster', norm(rel(pos, cm_pos[lab])), ids, hyp),
            EdgeSet('inter_cluster', norm(rel(cm_pos[cs], cm_pos[cr])), N + cs, N + cr)]
    return MultiGraph([nodes, norm(hyper)], sets)


def _grid_tets(nx: int, ny: int, nz: int, offset: int = 0) -> torch.Tensor:
    """4-vertex cells of an nx x ny x nz node grid: five tetrahedra per cube."""
    def nid(i, j, k):
        return offset + (i * ny + j) * nz + k
    cells = []
    for i in range(nx - 1):
        for j in range(ny - 1):
            for k in range(nz - 1):
                v = [nid(i + a, j + b, k + c) for a in (0, 1) for b in (0, 1) for c in (0, 1)]   # v[4a+2b+c]
                cells += [[v[0], v[1], v[2], v[4]], [v[3], v[1], v[2], v[7]], [v[5], v[1], v[4], v[7]],
                          [v[6], v[2], v[4], v[7]], [v[1], v[2], v[4], v[7]]]
    return torch.tensor(cells, dtype=torch.int64)


def plate_frame(seed: int = 0, nx: int = 7, ny: int = 6, nz: int = 3, obstacle=(4, 4, 2), obstacle_first: bool = True,
                spacing: float = 0.02, dtype=torch.float32) -> dict:
    """One deforming_plate-shape frame (src/model/plate.py:69-83): a plate of NORMAL (0) / HANDLE (3) nodes and a
    contiguous block of OBSTACLE (1) nodes hovering 0.8 grid spacings above it (so obstacle -> plate pairs fall inside the
    0.03 world-edge radius), 4-vertex cells, world / target positions [N,3], mesh_pos [N,3]."""
    g = torch.Generator().manual_seed(seed)

    def grid(n, origin):
        ax = [torch.arange(m, dtype=dtype) * spacing for m in n]
        p = torch.stack(torch.meshgrid(*ax, indexing='ij'), -1).reshape(-1, 3)
        return p + torch.tensor(origin, dtype=dtype)
    plate = grid((nx, ny, nz), (0.0, 0.0, 0.0))
    ox, oy, oz = obstacle
    obst = grid(obstacle, (0.4 * spacing, 0.3 * spacing, (nz - 1) * spacing + 0.8 * spacing))
    n_p, n_o = plate.shape[0], obst.shape[0]
    t_plate = torch.zeros(n_p, dtype=torch.int32)
    t_plate[:ny * nz] = 3                                       # the i = 0 face is clamped (HANDLE)
    t_obst = torch.ones(n_o, dtype=torch.int32)
    if obstacle_first:
        mesh_pos = torch.cat([obst, plate])
        node_type = torch.cat([t_obst, t_plate])
        cells = torch.cat([_grid_tets(ox, oy, oz, 0), _grid_tets(nx, ny, nz, n_o)])
    else:
        mesh_pos = torch.cat([plate, obst])
        node_type = torch.cat([t_plate, t_obst])
        cells = torch.cat([_grid_tets(nx, ny, nz, 0), _grid_tets(ox, oy, oz, n_p)])
    world = mesh_pos + 0.002 * torch.randn(mesh_pos.shape, generator=g, dtype=dtype)
    target = world + 0.001 * torch.randn(mesh_pos.shape, generator=g, dtype=dtype)
    target[node_type == 1] = world[node_type == 1] + torch.tensor([0.0, 0.0, -0.002], dtype=dtype)
    return {'world_pos': world, 'target|world_pos': target, 'mesh_pos': mesh_pos, 'node_type': node_type.reshape(-1, 1),
            'cells': cells, 'stress': torch.zeros(mesh_pos.shape[0], 1, dtype=dtype)}
