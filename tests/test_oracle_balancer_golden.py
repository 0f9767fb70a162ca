"""Pins oracle/balancer_oracle.py to outputs of the reference's own graph-balancer code (tests/golden/balancer.pt, generator
tests/golden/gen_golden_balancer.py: numba-CUDA kernel bodies emulated thread by thread, SDRF loop, random balancing, FlagModel
with a balancer).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import balancer_oracle as BO
from oracle import features_oracle as FO

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def fx():
    return torch.load(os.path.join(GOLDEN, 'balancer.pt'), weights_only=False)


def test_forman_curvature_bit_exact(fx):
    for c in fx['curvature']:
        A = BO.dense_adjacency(c['edge_index'][0], c['edge_index'][1])
        assert np.array_equal(A, c['A'].numpy())
        assert np.array_equal(BO.forman_curvature(A), c['C'].numpy()), c['name']


def test_post_delta_bit_exact(fx):
    for c, d in zip(fx['curvature'], fx['post_delta']):
        D = BO.post_delta(c['A'].numpy(), d['x'], d['y'], d['x_neighbors'], d['y_neighbors'])
        assert np.array_equal(D, d['D'].numpy()), d['name']


def test_sdrf_trajectories(fx):
    graphs = {c['name']: c['edge_index'] for c in fx['curvature']}
    for s in fx['sdrf']:
        ei = graphs[s['name']]
        np.random.seed(s['seed'])
        added, removed = BO.sdrf(ei[0], ei[1], int(ei.max()) + 1, loops=s['loops'], remove_edges=s['remove_edges'], tau=s['tau'])
        assert added == {k: [int(v) for v in vs] for k, vs in s['added'].items()}
        assert removed == {k: [int(v) for v in vs] for k, vs in s['removed'].items()}


def test_flag_model_with_balancer(fx):
    for case in fx['flag']:
        cfg = case['config']['graph_balancer']
        ff = FO.FlagFeatures()
        np.random.seed(case['np_seed'])
        added = mask = None
        for i, fr in enumerate(case['frames']):
            g = ff.build_graph(fr, True)
            if added is None:
                s, r = g['edge_sets'][0].senders, g['edge_sets'][0].receivers
                if case['algorithm'] == 'random':
                    added, removed = BO.random_balance(fr['world_pos'].shape[0], cfg['random']['edge_amount'], cfg['remove_edges'])
                else:
                    added, removed = BO.sdrf(s, r, fr['world_pos'].shape[0], loops=cfg['ricci']['loops'],
                                             remove_edges=cfg['remove_edges'], tau=cfg['ricci']['tau'])
                mask = BO.determine_mask(s, r, removed) if removed is not None else None
                assert [int(v) for v in added['senders']] == [int(v) for v in case['added']['senders']]
                assert torch.equal(mask, case['mask'])
            sets = BO.apply_balancer(g, added, mask, ff.mesh_edge, True)
            ref = case['edge_sets'][i]
            assert [e.name for e in sets] == [e['name'] for e in ref]
            for a, b in zip(sets, ref):
                assert torch.equal(a.senders, b['senders']) and torch.equal(a.receivers, b['receivers'])
                torch.testing.assert_close(a.features, b['features'], rtol=1e-4, atol=1e-4)
