"""Shared by the CPU test that pins the rollout oracle to the reference's own trajectories (tests/golden/rollout_*.pt,
generator tests/golden/gen_golden_rollout.py) and by the GPU test that compares the HIP rollout with both."""
import os

import torch

from oracle import features_oracle as FO
from oracle import mgn_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ['flag_none', 'flag_hyper_k4', 'cylinder_none', 'plate_none', 'plate_hetero_k4']
ARCH = {'none': 'none', 'hyper': 'hyper', 'hetero': 'hetero'}


def load(name):
    return torch.load(os.path.join(GOLDEN, f'rollout_{name}.pt'), weights_only=False)


def weights(fx, dtype=torch.float32):
    """The fixture stores the seed of the build-owned deterministic init, not the values."""
    return {k: v.to(dtype) for k, v in O.init_state_dict_like(fx['weight_shapes'], fx['weight_seed']).items()}


def frame_at(traj, i):
    return {k: v[i] for k, v in traj.items()}


def oracle_rollout(name, fx, dtype, clusters=None, neighbors=None):
    """The reference's call sequence restated on the oracle: warm-up frames in training mode (normaliser statistics), then
    rollout(T) and n_step_computation(n_step).  Returns (predictions dict, mse, (n-step mean, n-step last), features object)."""
    kind = name.split('_')[0]
    feats = {'flag': FO.FlagFeatures, 'cylinder': FO.CylinderFeatures, 'plate': FO.PlateFeatures}[kind](dtype=dtype)
    cfg = fx['config']
    connector = cfg['rmp']['connector']
    sd = weights(fx, dtype)
    arch, agg = ARCH[connector], cfg['aggregation']
    expand = None
    if connector != 'none':
        clusters = fx['clusters'] if clusters is None else clusters
        neighbors = fx['neighbors'] if neighbors is None else neighbors
        nb = [tuple(torch.as_tensor(t).tolist()) for t in neighbors]

        def expand(g, step, training):
            return FO.hierarchical_connect(g, clusters, nb, feats.intra_edge, feats.inter_edge, feats.hyper_node, training)
    for i, fr in enumerate(fx['warm']):
        g = feats.build_graph(fr, True)
        feats.get_target(fr, True)
        if expand is not None:
            expand(g, i, True)

    def net(graph):
        return O.mesh_graph_net(sd, graph, arch, agg)
    traj, T, n = fx['trajectory'], fx['T'], fx['n_step']
    with torch.no_grad():
        if kind == 'flag':
            roll = lambda tr, steps: FO.flag_rollout(feats, net, tr, steps, expand)            # noqa: E731
            pred, mse = roll(traj, T)
            preds = {'pred_pos': pred}
        elif kind == 'cylinder':
            roll = lambda tr, steps: FO.cylinder_rollout(feats, net, tr, steps, expand)        # noqa: E731
            vel, pr, mse = roll(traj, T)
            preds = {'pred_velocity': vel, 'pred_pressure': pr}
        else:
            roll = lambda tr, steps: FO.plate_rollout(feats, net, tr, steps, expand)           # noqa: E731
            pred, mse = roll(traj, T)
            preds = {'pred_pos': pred}
        nstep = FO.n_step_computation(roll, traj, n)
    return preds, mse, nstep, feats


def step_scale(fx, key):
    """Size of one step's change of state, the scale a rollout error is measured against (positions are O(1), one step moves
    them by O(1e-2): an error relative to max|position| would hide everything the network does)."""
    p = fx['rollout'][key].double()
    if key == 'pred_pressure' or p.shape[0] < 2:          # a direct network output, not an integrated state
        return float(p.abs().max())
    return float((p[1:] - p[:-1]).abs().max())


def per_step_err(got, want, scale):
    """[T]: max-abs error of every recorded step over the step scale."""
    d = (got.detach().cpu().double() - want.double()).abs().flatten(1).max(1).values
    return d / scale
