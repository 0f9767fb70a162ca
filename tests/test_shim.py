"""The `src` path shim (hyper-graph-nets_amd/shim): every hot-path module path the reference imports resolves to the MI355X
implementation, and everything else falls through to the reference's own `src` package (north_star: "drops into main.py
unchanged").  Import lists follow /root/reference src/model/flag.py:4-11,47, src/model/get_model.py:4-6,
src/algorithms/MeshSimulator.py:23-24, src/rmp/get_rmp.py:4-16, src/graph_balancer/get_graph_balancer.py:4-7."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'hyper-graph-nets_amd')
SHIM = os.path.join(PKG, 'shim')

PATHS = {
    'src.migration.meshgraphnet': ['MeshGraphNet', 'LazyMLP'], 'src.migration.normalizer': ['Normalizer'],
    'src.migration.graphnet': ['GraphNet'], 'src.migration.hypergraphnet': ['HyperGraphNet'],
    'src.migration.heterographnet': ['HeteroGraphNet'], 'src.migration.multiscalegraphnet': ['MultiScaleGraphNet'],
    'src.migration.multigraphnet': ['MultiGraphNet'], 'src.migration.repeatedgraphnet': ['RepeatedGraphNet'],
    'src.migration.encoder': ['Encoder'], 'src.migration.processor': ['Processor'], 'src.migration.decoder': ['Decoder'],
    'src.util': ['EdgeSet', 'MultiGraph', 'MultiGraphWithPos', 'NodeType', 'device', 'detach', 'read_yaml', 'triangles_to_edges',
                 'unsorted_segment_operation'],
    'src.model.abstract_system_model': ['AbstractSystemModel'], 'src.model.flag': ['FlagModel'],
    'src.model.cylinder': ['CylinderModel'], 'src.model.plate': ['PlateModel'], 'src.model.get_model': ['get_model'],
    'src.rmp.get_rmp': ['get_rmp', 'get_clustering_algorithm', 'get_connector'],
    'src.rmp.remote_message_passing': ['RemoteMessagePassing'], 'src.rmp.hierarchical_connector': ['HierarchicalConnector'],
    'src.rmp.multigraph_connector': ['MultigraphConnector'], 'src.rmp.abstract_connector': ['AbstractConnector'],
    'src.rmp.abstract_clustering_algorithm': ['AbstractClusteringAlgorithm'], 'src.rmp.k_means_clustering': ['KMeansClustering'],
    'src.rmp.spectral_clustering': ['SpectralClustering'], 'src.rmp.gaussian_mixture': ['GaussianMixtureClustering'],
    'src.rmp.random_clustering': ['RandomClustering'], 'src.rmp.hdbscan': ['HDBSCAN'],
    'src.graph_balancer.get_graph_balancer': ['get_balancer', 'get_balancer_algorithm'],
    'src.graph_balancer.graph_balancer': ['GraphBalancer'], 'src.graph_balancer.abstract_graph_balancer': ['AbstractGraphBalancer'],
    'src.graph_balancer.random_balancing': ['RandomGraphBalancer'], 'src.graph_balancer.ricci': ['Ricci'],
}


def _run(code, extra_path=()):
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join([SHIM, PKG, *extra_path])
    env['PYTHONDONTWRITEBYTECODE'] = '1'
    return subprocess.run([sys.executable, '-c', textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=300,
                          cwd=ROOT)


def test_reference_module_paths_resolve_to_the_hip_implementation():
    code = f'''
        import importlib
        paths = {PATHS!r}
        for mod, names in paths.items():
            m = importlib.import_module(mod)
            for n in names:
                obj = getattr(m, n)
                owner = getattr(obj, '__module__', 'hgn_amd.util')
                assert owner.startswith('hgn_amd') or n in ('device', 'EdgeSet', 'MultiGraph', 'MultiGraphWithPos'), (mod, n, owner)
        # the statements of the reference's system models, verbatim (flag.py:4-11,47)
        import src.rmp.get_rmp as rmp
        from src import util
        from src.migration.meshgraphnet import MeshGraphNet
        from src.migration.normalizer import Normalizer
        from src.model.abstract_system_model import AbstractSystemModel
        from src.util import EdgeSet, MultiGraphWithPos, NodeType, device, MultiGraph
        import src.graph_balancer.get_graph_balancer as graph_balancer
        import hgn_amd
        assert MeshGraphNet is hgn_amd.MeshGraphNet and Normalizer is hgn_amd.Normalizer
        assert rmp.get_rmp is hgn_amd.rmp.get_rmp and graph_balancer.get_balancer is hgn_amd.graph_balancer.get_balancer
        assert util.unsorted_segment_operation is hgn_amd.util.unsorted_segment_operation
        print('ok', len(paths))
    '''
    r = _run(code)
    assert r.returncode == 0 and r.stdout.startswith('ok'), r.stderr[-3000:]


def test_modules_outside_the_hot_path_fall_through_to_the_reference_tree(tmp_path):
    """src.algorithms / src.tasks / src.data are not provided by the shim: they must come from the reference checkout that
    follows on PYTHONPATH (stand-in tree here: the reference is absent on the GPU box), while shimmed paths still win."""
    ref = tmp_path / 'reference'
    for d in ('src/algorithms', 'src/migration', 'src/rmp'):
        (ref / d).mkdir(parents=True)
        (ref / d / '__init__.py').write_text('')
    (ref / 'src' / '__init__.py').write_text('')
    (ref / 'src' / 'algorithms' / 'MeshSimulator.py').write_text('from src.model.get_model import get_model\nWHO = "reference trainer"\n')
    (ref / 'src' / 'migration' / 'meshgraphnet.py').write_text('WHO = "reference model (must be shadowed)"\n')
    (ref / 'src' / 'rmp' / 'user_extension.py').write_text('WHO = "a module of the checkout the shim does not provide"\n')
    code = '''
        from src.algorithms.MeshSimulator import WHO, get_model
        import src.migration.meshgraphnet as m
        import src.rmp.user_extension as h
        import hgn_amd.system_model
        assert WHO == "reference trainer" and get_model is hgn_amd.system_model.get_model
        assert not hasattr(m, 'WHO') and m.MeshGraphNet.__module__ == 'hgn_amd.modules'
        assert h.WHO == "a module of the checkout the shim does not provide"
        print('ok')
    '''
    r = _run(code, extra_path=[str(ref)])
    assert r.returncode == 0 and r.stdout.startswith('ok'), r.stderr[-3000:]


# ------------------------------------------------------------------------------------------------------------------------
# the launcher (hgn_amd/run_main.py): the arrangement a user actually has -- cwd = the reference checkout, whose own REGULAR
# package `src` sits at sys.path[0] ahead of PYTHONPATH
# ------------------------------------------------------------------------------------------------------------------------
REFERENCE = '/root/reference'
ORACLE_SHIMS = os.path.join(ROOT, 'tools', 'oracle_shims')      # import-only stand-ins for wandb / tfrecord / ... (trainer side)


def _launch(args, cwd, extra_path=(), module='hgn_amd.run_main'):
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join([PKG, *extra_path])      # the shim directory is NOT listed: the launcher places it
    env['PYTHONDONTWRITEBYTECODE'] = '1'
    env.pop('HGN_REFERENCE', None)
    return subprocess.run([sys.executable, '-m', module, *args], env=env, capture_output=True, text=True, timeout=600, cwd=cwd)


def _standin_reference(tmp_path):
    """A tree shaped like the reference checkout: regular package `src` (with __init__.py) holding BOTH hot-path modules that
    must be shadowed and trainer-side modules that must be found, a top-level `util` package, and a main.py with the
    reference's import block (main.py:3-17) and __main__ body."""
    ref = tmp_path / 'reference'
    for d in ('src/algorithms', 'src/migration', 'src/model', 'src/rmp', 'src/tasks', 'util', 'configs'):
        (ref / d).mkdir(parents=True)
        (ref / d / '__init__.py').write_text('')
    (ref / 'src' / '__init__.py').write_text('WHO = "reference src package"\n')
    (ref / 'src' / 'util.py').write_text('import torch_scatter_must_not_be_imported\n')
    (ref / 'src' / 'model' / 'flag.py').write_text('class FlagModel: pass\n')
    (ref / 'src' / 'model' / 'get_model.py').write_text('def get_model(*a): raise RuntimeError("reference get_model")\n')
    (ref / 'src' / 'migration' / 'meshgraphnet.py').write_text('class MeshGraphNet: pass\n')
    (ref / 'src' / 'algorithms' / 'MeshSimulator.py').write_text(
        'from src.model.flag import FlagModel\nfrom src.model.get_model import get_model\n'
        'from src.util import detach, EdgeSet, MultiGraph\nclass MeshSimulator: pass\n')
    (ref / 'src' / 'tasks' / 'get_task.py').write_text('from src.algorithms.MeshSimulator import MeshSimulator\n'
                                                        'def get_task(params): return params\n')
    (ref / 'util' / 'Functions.py').write_text('def get_from_nested_dict(*a, **k): return None\n')
    (ref / 'main.py').write_text(textwrap.dedent("""
        import json, os, sys
        from src.tasks.get_task import get_task
        from src.util import device, read_yaml
        from src.algorithms.MeshSimulator import MeshSimulator
        from util.Functions import get_from_nested_dict
        if __name__ == '__main__':
            import src, src.model.flag, src.migration.meshgraphnet as mgn
            print(json.dumps({'argv': sys.argv[1:], 'cwd': os.getcwd(), 'src': src.__file__,
                              'FlagModel': src.model.flag.FlagModel.__module__, 'MeshGraphNet': mgn.MeshGraphNet.__module__,
                              'MeshSimulator': sys.modules['src.algorithms.MeshSimulator'].__file__,
                              'has_who': hasattr(src, 'WHO')}))
    """))
    return ref


def test_launcher_puts_the_shim_ahead_of_a_regular_src_package_in_the_working_directory(tmp_path):
    """The arrangement in which the PYTHONPATH recipe fails: cwd (= sys.path[0] under `python main.py` / `python -m`) holds a
    REGULAR package `src`.  Through the launcher the hot-path modules come from hgn_amd, the trainer side from the checkout, and
    main.py runs as __main__ with its argv and the checkout as cwd."""
    import json
    ref = _standin_reference(tmp_path)
    r = _launch(['flag'], cwd=str(ref))
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep['argv'] == ['flag'] and os.path.samefile(rep['cwd'], ref)
    assert rep['src'].startswith(SHIM) and not rep['has_who']
    assert rep['FlagModel'] == 'hgn_amd.system_model' and rep['MeshGraphNet'] == 'hgn_amd.modules'
    assert rep['MeshSimulator'].startswith(str(ref))
    # --reference from somewhere else, and the import-only probe
    r = _launch(['--reference', str(ref), '--probe', 'flag'], cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep['reference_hot'] == [] and rep['FlagModel'] == 'hgn_amd.system_model' and rep['get_model'] == 'hgn_amd.system_model'
    assert 'src.algorithms.MeshSimulator' in rep['reference'] and 'src.util' in rep['shim']
    # the same arrangement WITHOUT the launcher binds `src` to the checkout (why the launcher exists): the stand-in's src/util.py
    # is what gets imported
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([SHIM, PKG]), PYTHONDONTWRITEBYTECODE='1')
    bare = subprocess.run([sys.executable, 'main.py', 'flag'], env=env, capture_output=True, text=True, timeout=300, cwd=str(ref))
    assert bare.returncode != 0 and 'torch_scatter_must_not_be_imported' in bare.stderr


def test_install_shim_refuses_when_src_is_already_bound_to_the_checkout(tmp_path):
    ref = _standin_reference(tmp_path)
    code = f'''
        import sys
        sys.path.insert(0, {str(ref)!r})
        import src
        from hgn_amd import run_main
        try:
            run_main.install_shim({str(ref)!r})
        except RuntimeError as e:
            assert 'already imported' in str(e)
            print('ok')
    '''
    r = _run(code)
    assert r.returncode == 0 and r.stdout.startswith('ok'), r.stderr[-3000:]
    r = _launch(['--reference', str(tmp_path / 'nowhere'), 'flag'], cwd=str(tmp_path))
    assert r.returncode != 0 and 'not a checkout of the reference' in r.stderr


@pytest.mark.skipif(not (os.path.isfile(os.path.join(REFERENCE, 'main.py')) and os.path.isdir(ORACLE_SHIMS)),
                    reason='needs the reference checkout (build container only) and the trainer-side import stand-ins')
def test_launcher_against_the_real_reference_checkout():
    """Where the reference lives: `python -m hgn_amd.run_main --probe flag` with cwd = /root/reference executes main.py's own
    import block (main.py:3-17: src.tasks.get_task, src.util, src.tasks.MeshTask, src.algorithms.MeshSimulator, util.Functions)
    and reports the origin of every src.* module.  No module of src/{util,model,migration,rmp,graph_balancer} may come from the
    checkout; the trainer / tasks / data side must.  torch_scatter (imported by the reference's src/util.py:5, not installed)
    is never asked for."""
    import json
    r = _launch(['--probe', 'flag'], cwd=REFERENCE, extra_path=[ORACLE_SHIMS])
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep['reference_hot'] == [], rep['reference_hot']
    assert rep['FlagModel'] == 'hgn_amd.system_model' and rep['get_model'] == 'hgn_amd.system_model'
    assert rep['src.util'].startswith(SHIM)
    assert rep['MeshSimulator'] == os.path.join(REFERENCE, 'src', 'algorithms', 'MeshSimulator.py')
    for m in ('src.algorithms.MeshSimulator', 'src.tasks.get_task', 'src.tasks.MeshTask', 'src.data.data_loader'):
        assert m in rep['reference'], m
    for m in ('src', 'src.util', 'src.model.flag', 'src.model.get_model'):
        assert m in rep['shim'], m
    assert not rep['torch_scatter_imported']
    # and the full run gets as far as the (absent) dataset, i.e. through get_task -> MeshTask -> get_data on the shimmed src.util
    r = _launch(['flag'], cwd=REFERENCE, extra_path=[ORACLE_SHIMS])
    assert r.returncode != 0 and 'tfrecord stub: datasets are not available' in r.stderr, r.stderr[-2000:]
    assert 'Device used for this run' in r.stdout
