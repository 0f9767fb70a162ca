"""Pins the rollout restatement (oracle/features_oracle.py: flag_rollout / cylinder_rollout / plate_rollout /
n_step_computation) to trajectories the reference's own FlagModel / CylinderModel / PlateModel produced
(tests/golden/rollout_*.pt).  CPU only.

Tolerance: the oracle in fp32 restates the reference's fp32 arithmetic up to summation order, so its trajectory agrees with
the fixture to 2e-5 of one step's displacement at the first predicted step, with a growth allowance of 2x per further step
(the state feeds back); in fp64 it gives the exact value both fp32 evaluations are measured against on the GPU."""
import pytest
import torch

from tests import rollout_cases as RC


@pytest.mark.parametrize('name', RC.CASES)
def test_oracle_rollout_reproduces_the_reference_trajectory(name):
    fx = RC.load(name)
    preds, mse, nstep, feats = RC.oracle_rollout(name, fx, torch.float32)
    assert mse.shape[0] == fx['rollout_steps']
    for key, got in preds.items():
        want = fx['rollout'][key]
        assert got.shape == want.shape
        err = RC.per_step_err(got, want, RC.step_scale(fx, key))
        bound = torch.tensor([2e-5 * 2.0 ** max(t - 1, 0) for t in range(err.shape[0])], dtype=torch.float64)
        assert bool((err <= bound).all()), (name, key, err.tolist())
    torch.testing.assert_close(mse, fx['mse'], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(nstep[0], fx['n_step_result'][0], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(nstep[1], fx['n_step_result'][1], rtol=1e-5, atol=1e-9)
    # the rollout ran in evaluation mode: no normaliser accumulated, except flag's node-dynamic one (flag.py:115)
    for key, st in fx['normalizers_after'].items():
        same = torch.equal(st['acc_count'], fx['normalizers'][key]['acc_count'])
        assert same or key == '_node_dynamic_normalizer', key


def test_rollout_first_recorded_state_follows_each_models_convention():
    """flag records the state BEFORE a step (pred[0] is the input frame), cylinder and plate the state AFTER it."""
    fx = RC.load('flag_none')
    assert torch.equal(fx['rollout']['pred_pos'][0], fx['trajectory']['world_pos'][0])
    fx = RC.load('plate_none')
    assert not torch.equal(fx['rollout']['pred_pos'][0], fx['trajectory']['world_pos'][0])
    moved = fx['trajectory']['node_type'][0][:, 0] != 0
    assert torch.equal(fx['rollout']['pred_pos'][0][moved], fx['trajectory']['target|world_pos'][0][moved])
    fx = RC.load('cylinder_none')
    assert fx['rollout_steps'] == fx['trajectory']['cells'].shape[0]          # num_steps argument ignored (cylinder.py:178)
