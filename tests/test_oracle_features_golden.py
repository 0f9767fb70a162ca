"""Pins oracle/features_oracle.py (frame -> graph features, remote-graph assembly) to outputs of the reference's own
FlagModel / CylinderModel / RemoteMessagePassing / util.triangles_to_edges (tests/golden/feat_*.pt, generator
tests/golden/gen_golden_features.py).  CPU only."""
import os

import pytest
import torch

from oracle import features_oracle as FO
from oracle import mgn_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = dict(rtol=2e-5, atol=2e-6)       # fp32 restatement vs the reference's fp32: summation order only


def load(name):
    return torch.load(os.path.join(GOLDEN, f'feat_{name}.pt'), weights_only=False)


def directed_set(s, r):
    return sorted(zip(s.tolist(), r.tolist()))


def test_cells_to_edges_bit_exact():
    fx = load('cells_to_edges')
    for nm, deform in (('tri', False), ('quad', True)):
        s, r = FO.triangles_to_edges(fx[nm]['cells'], deform)
        assert torch.equal(s, fx[nm]['senders']) and torch.equal(r, fx[nm]['receivers'])


@pytest.mark.parametrize('name', ['flag_none', 'flag_hyper_k5', 'flag_hyper_k3_full', 'flag_hyper_k4_sampled'])
def test_flag_build_graph_and_expand(name):
    fx = load(name)
    cfg = fx['config']
    ff = FO.FlagFeatures()
    for i, fr in enumerate(fx['frames']):
        training = i < 2
        g = ff.build_graph(fr, training)
        ref = fx['graphs'][i]
        e, re_ = g['edge_sets'][0], ref['edge_sets'][0]
        assert torch.equal(e.senders, re_['senders']) and torch.equal(e.receivers, re_['receivers'])
        torch.testing.assert_close(g['unnormalized_edges'].features, ref['unnormalized_edges']['features'], **TOL)
        torch.testing.assert_close(e.features, re_['features'], **TOL)
        torch.testing.assert_close(g['node_features'][0], ref['node_features'][0], **TOL)
        torch.testing.assert_close(g['node_dynamic'], ref['node_dynamic'], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(ff.get_target(fr, training), fx['targets'][i], rtol=1e-4, atol=1e-4)
        if fx['expanded']:
            ex = fx['expanded'][i]
            s, r = e.senders, e.receivers
            nb = FO.neighboring_clusters(s, r, ex['labels'])
            assert nb == sorted({(min(a, b), max(a, b)) for a, b in (tuple(t.tolist()) for t in ex['neighbors'])})
            rmp = cfg['rmp']
            # the reference's neighbour ORDER (Python set iteration) is an input here, so that rows line up
            mg = FO.hierarchical_connect(g, ex['clusters'], [tuple(t.tolist()) for t in ex['neighbors']],
                                         ff.intra_edge, ff.inter_edge, ff.hyper_node, training,
                                         hyper_node_features=rmp['hyper_node_features'],
                                         fully_connect=rmp['fully_connect'])
            assert [x.name for x in mg.edge_sets] == [x['name'] for x in ex['edge_sets']]
            for a, b in zip(mg.edge_sets, ex['edge_sets']):
                assert torch.equal(a.senders, b['senders']) and torch.equal(a.receivers, b['receivers']), a.name
                torch.testing.assert_close(a.features, b['features'], rtol=1e-4, atol=1e-4)
            for a, b in zip(mg.node_features, ex['node_features']):
                torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ff.update(fx['frames'][0], fx['net_out']), fx['update'], **TOL)
    for key, nz in (('_output_normalizer', ff.output), ('_node_normalizer', ff.node),
                    ('_node_dynamic_normalizer', ff.node_dynamic), ('_mesh_edge_normalizer', ff.mesh_edge),
                    ('_intra_edge_normalizer', ff.intra_edge), ('_inter_edge_normalizer', ff.inter_edge),
                    ('_hyper_node_normalizer', ff.hyper_node)):
        st = fx['normalizers'][key]
        torch.testing.assert_close(nz.acc_sum, st['acc_sum'], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(nz.acc_sum_sq, st['acc_sum_squared'], rtol=1e-5, atol=1e-5)
        assert torch.equal(nz.acc_count, st['acc_count']) and torch.equal(nz.num_acc, st['num_accumulations'])


def test_cylinder_build_graph():
    fx = load('cylinder')
    cf = FO.CylinderFeatures()
    for i, fr in enumerate(fx['frames']):
        g = cf.build_graph(fr, i < 1)
        ref = fx['graphs'][i]
        e, re_ = g['edge_sets'][0], ref['edge_sets'][0]
        assert torch.equal(e.senders, re_['senders']) and torch.equal(e.receivers, re_['receivers'])
        torch.testing.assert_close(e.features, re_['features'], **TOL)
        torch.testing.assert_close(g['unnormalized_edges'].features, ref['unnormalized_edges']['features'], **TOL)
        torch.testing.assert_close(g['node_features'][0], ref['node_features'][0], **TOL)
        torch.testing.assert_close(cf.get_target(fr, i < 1), fx['targets'][i], rtol=1e-4, atol=1e-4)
    v, p = cf.update(fx['frames'][0], fx['net_out'])
    torch.testing.assert_close(v, fx['update'][0], **TOL)
    torch.testing.assert_close(p, fx['update'][1], **TOL)


@pytest.mark.parametrize('name', ['plate_hetero_k4_first', 'plate_none_last', 'plate_hetero_k4_last'])
def test_plate_build_graph_and_expand(name):
    fx = load(name)
    pf = FO.PlateFeatures()
    for i, fr in enumerate(fx['frames']):
        training = i < 1
        g = pf.build_graph(fr, training)
        ref = fx['graphs'][i]
        assert [e.name for e in g['edge_sets']] == [e['name'] for e in ref['edge_sets']]
        for e, re_ in zip(g['edge_sets'], ref['edge_sets']):
            assert torch.equal(e.senders, re_['senders']) and torch.equal(e.receivers, re_['receivers']), e.name
            torch.testing.assert_close(e.features, re_['features'], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(g['unnormalized_edges'].features, ref['unnormalized_edges']['features'], **TOL)
        torch.testing.assert_close(g['node_features'][0], ref['node_features'][0], rtol=1e-4, atol=1e-4)
        assert torch.equal(g['obstacle_nodes'], ref['obstacle_nodes'])
        torch.testing.assert_close(pf.get_target(fr, training), fx['targets'][i], rtol=1e-4, atol=1e-4)
        if fx['expanded']:
            ex = fx['expanded'][i]
            mg = FO.hierarchical_connect(g, ex['clusters'], [tuple(t.tolist()) for t in ex['neighbors']], pf.intra_edge,
                                         pf.inter_edge, pf.hyper_node, training)
            assert [x.name for x in mg.edge_sets] == [x['name'] for x in ex['edge_sets']]
            for a, b in zip(mg.edge_sets, ex['edge_sets']):
                assert torch.equal(a.senders, b['senders']) and torch.equal(a.receivers, b['receivers']), a.name
                torch.testing.assert_close(a.features, b['features'], rtol=1e-4, atol=1e-4)
            for a, b in zip(mg.node_features, ex['node_features']):
                torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(pf.update(fx['frames'][0], fx['net_out']), fx['update'], **TOL)


def test_plate_multigraph_connector():
    fx = load('plate_multi_k4_first')
    pf = FO.PlateFeatures()
    for i, fr in enumerate(fx['frames']):
        g = pf.build_graph(fr, i < 1)
        ex = fx['expanded'][i]
        mg = FO.multigraph_connect(g, ex['clusters'], [tuple(t.tolist()) for t in ex['neighbors']], pf.intra_edge,
                                   pf.inter_edge, pf.hyper_node, i < 1)
        assert [x.name for x in mg.edge_sets] == [x['name'] for x in ex['edge_sets']] == ['mesh_edges', 'world_edges']
        for a, b in zip(mg.edge_sets, ex['edge_sets']):
            assert torch.equal(a.senders, b['senders']) and torch.equal(a.receivers, b['receivers'])
            torch.testing.assert_close(a.features, b['features'], rtol=1e-4, atol=1e-4)
        for a, b in zip(mg.node_features, ex['node_features']):
            torch.testing.assert_close(a, b.float(), rtol=1e-4, atol=1e-4)
