"""Rollout inference on the HIP path against trajectories the REFERENCE's own
FlagModel / CylinderModel / PlateModel produced (tests/golden/rollout_*.pt; flag.py:194-260, cylinder.py:175-245,
plate.py:264-347) and against the fp64 oracle on the same inputs.

Metrics and tolerances.
 * norm-wise (BASELINE.json's 1e-5): max|a - b| / max|b| per recorded step <= 1e-5 -- positions are O(1) and one step moves them
   by O(1e-2), so this bound is met with two orders to spare and says little;
 * step-relative (the strict one): max|a - b| / (largest change of state between two recorded steps).  Against the fixture
   <= 2e-5 at the first predicted step, growing by at most 2x per further step (the state feeds back into the next
   frame's features); against the fp64 oracle <= max(1e-5, 3x the distance of the reference's own fp32 trajectory from fp64) --
   the reference's fp32 sits 0.8e-5 .. 3.9e-5 of a step from fp64 after 1 .. 3 steps (rounding of O(1) positions);
 * per-step MSE and the two n-step figures: rtol 1e-5.
Replayed (HIP graph captured at the second sight of a topology) and eager launches must agree bit for bit."""
import random

import numpy as np
import pytest
import torch

from tests import rollout_cases as RC

pytestmark = pytest.mark.gpu


def cuda(d):
    return {k: v.cuda() for k, v in d.items()}


def hip_system_model(name, fx):
    from hgn_amd import system_model
    kind = name.split('_')[0]
    cls = {'flag': system_model.FlagModel, 'cylinder': system_model.CylinderModel, 'plate': system_model.PlateModel}[kind]
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    model = cls(fx['config'])                                  # the reference's own `model` section of the YAML
    model.evaluate()
    for i, fr in enumerate(fx['warm']):                        # same call sequence as the generator
        g = model.build_graph(cuda(fr), True)
        model.get_target(cuda(fr), True)
        g = model.expand_graph(g, i, len(fx['warm']), True)
    with torch.no_grad():
        model.learned_model(g)                                 # materialise the lazy layers
    sd = RC.weights(fx)
    assert {k: tuple(v.shape) for k, v in model.learned_model.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
    model.learned_model.load_state_dict({k: v.cuda() for k, v in sd.items()}, strict=True)
    return model


def hip_predictions(ops):
    return {k: v for k, v in ops.items() if k in ('pred_pos', 'pred_velocity', 'pred_pressure')}


@pytest.mark.parametrize('name', RC.CASES)
def test_hip_rollout_matches_the_reference_trajectory(name):
    fx = RC.load(name)
    model = hip_system_model(name, fx)
    # normaliser statistics in front of the rollout equal the reference's
    for key, st in fx['normalizers'].items():
        nz = getattr(model, key)
        torch.testing.assert_close(nz._acc_sum.cpu(), st['acc_sum'], rtol=1e-5, atol=1e-4)
        assert torch.equal(nz._acc_count.cpu(), st['acc_count'])
    traj = cuda(fx['trajectory'])
    T = fx['T']
    ops, mse = model.rollout(traj, T)
    if fx['config']['rmp']['connector'] != 'none':            # the clustering the reference's scikit-learn call produced
        got = sorted(tuple(sorted(torch.as_tensor(c).tolist())) for c in model._remote_graph._clusters)
        want = sorted(tuple(sorted(torch.as_tensor(c).tolist())) for c in fx['clusters'])
        assert got == want
    p64, mse64, nstep64, _ = RC.oracle_rollout(name, fx, torch.float64)
    report = {}
    for key, got in hip_predictions(ops).items():
        want = fx['rollout'][key]
        assert got.shape == want.shape and mse.shape[0] == fx['rollout_steps']
        normwise = (got.cpu().double() - want.double()).abs().flatten(1).max(1).values / want.double().abs().max()
        assert float(normwise.max()) <= 1e-5, (name, key, normwise.tolist())
        scale = RC.step_scale(fx, key)
        e_ref = RC.per_step_err(got, want, scale)
        e_64 = RC.per_step_err(got, p64[key], scale)
        noise = RC.per_step_err(want, p64[key], scale)        # the reference's own fp32 against fp64
        first = 1 if name.startswith('flag') else 0            # flag records the input state first (exact)
        bound = torch.tensor([2e-5 * 2.0 ** max(t - first, 0) for t in range(e_ref.shape[0])], dtype=torch.float64)
        assert bool((e_ref <= bound).all()), (name, key, e_ref.tolist())
        assert bool((e_64 <= torch.clamp(3.0 * noise, min=1e-5)).all()), (name, key, e_64.tolist(), noise.tolist())
        report[key] = (e_ref.tolist(), e_64.tolist(), noise.tolist())
    torch.testing.assert_close(mse.cpu(), fx['mse'], rtol=1e-5, atol=1e-9)
    # replay: the steps above ran eager (step 0), captured (second sight of the topology) and replayed; eager == replayed
    model.replay_rollout = False
    eager_ops, eager_mse = model.rollout(traj, T)
    for key, got in hip_predictions(ops).items():
        assert torch.equal(eager_ops[key], got), (name, key)
    assert torch.equal(eager_mse, mse)
    # n_step_computation (flag.py:248-260): sliding windows, each rolled out from its first frame
    model.replay_rollout = True
    a, b = model.n_step_computation(traj, fx['n_step'])
    torch.testing.assert_close(a.cpu(), fx['n_step_result'][0], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(b.cpu(), fx['n_step_result'][1], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(a.cpu().double(), nstep64[0], rtol=1e-5, atol=1e-9)
    print(f'rollout[{name}] step-relative error vs reference / vs fp64 / reference-vs-fp64:', report)


def test_hip_rollout_is_independent_of_the_num_steps_hint_only_where_the_reference_is():
    """cylinder.py:178 overwrites num_steps with the trajectory length; flag and plate honour it."""
    fx = RC.load('cylinder_none')
    model = hip_system_model('cylinder_none', fx)
    ops, mse = model.rollout(cuda(fx['trajectory']), 2)
    assert mse.shape[0] == fx['trajectory']['cells'].shape[0]
    fx = RC.load('flag_none')
    model = hip_system_model('flag_none', fx)
    ops, mse = model.rollout(cuda(fx['trajectory']), 2)
    assert mse.shape[0] == 2 and torch.equal(ops['pred_pos'].cpu(), ops['pred_pos'].cpu())
    torch.testing.assert_close(ops['pred_pos'].cpu(), fx['rollout']['pred_pos'][:2], rtol=0, atol=1e-6)
