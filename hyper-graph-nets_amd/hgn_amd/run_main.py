"""Launcher: run the reference's own ``main.py`` (main.py:22-44), unchanged, on the MI355X implementation.

    cd <reference checkout>
    PYTHONPATH=<repo>/hyper-graph-nets_amd python -m hgn_amd.run_main flag          # what `python main.py flag` was
    python -m hgn_amd.run_main --probe flag                                         # import only; report who serves src.*

Why a launcher and not a PYTHONPATH recipe: ``python main.py`` puts the script's directory at ``sys.path[0]`` -- ahead of
PYTHONPATH -- and the reference's ``src/__init__.py`` makes ``src`` a regular package, so ``import src`` binds to the reference
tree and the path shim (hyper-graph-nets_amd/shim/src, a package named ``src`` that re-exports the hot-path modules of
flag.py:4-11, get_model.py:4-6, MeshSimulator.py:23-24) is never consulted.  Here the order is fixed from inside the
interpreter, before the first ``import src``:

    sys.path = [<shim>, <hyper-graph-nets_amd>, <reference checkout>, ...what was there]

``src`` then binds to the shim package, whose ``__path__`` extends over the reference's ``src`` directory, so ``src.algorithms``,
``src.tasks``, ``src.data`` (trainer, tasks, TFRecord input: not on the hot path) still come from the checkout, as does the
top-level ``util`` package (main.py:17).  ``main.py`` is then executed with ``runpy`` as ``__main__`` with the checkout as working
directory (configs are opened by relative path, src/util.py:39).  Nothing here touches the GPU; no process is replaced.
"""
import json
import os
import runpy
import sys

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))        # .../hyper-graph-nets_amd
SHIM_DIR = os.path.join(_PKG_DIR, 'shim')

# the subtrees of the reference's `src` that the shim replaces
HOT_SUBTREES = ('util', 'model', 'migration', 'rmp', 'graph_balancer')


def _same(a: str, b: str) -> bool:
    try:
        return os.path.samefile(a, b)
    except OSError:
        return os.path.abspath(a) == os.path.abspath(b)


def install_shim(reference_dir: str) -> None:
    """Order sys.path as [shim, package dir, reference checkout, rest] and bind ``src`` to the shim package.  Raises when a
    ``src`` imported earlier in this process is the reference's: modules already bound to it cannot be re-routed."""
    reference_dir = os.path.abspath(reference_dir)
    if not os.path.isfile(os.path.join(reference_dir, 'main.py')) or not os.path.isdir(os.path.join(reference_dir, 'src')):
        raise FileNotFoundError(f'{reference_dir} is not a checkout of the reference (main.py and src/ expected)')
    rest = []
    for p in sys.path:
        q = p or os.getcwd()
        if any(_same(q, d) for d in (SHIM_DIR, _PKG_DIR, reference_dir)):
            continue
        rest.append(p)
    sys.path[:] = [SHIM_DIR, _PKG_DIR, reference_dir] + rest
    # child interpreters (multiprocessing 'spawn' hands sys.path over by itself; anything else reads the environment)
    env = [SHIM_DIR, _PKG_DIR, reference_dir] + [p for p in os.environ.get('PYTHONPATH', '').split(os.pathsep) if p]
    os.environ['PYTHONPATH'] = os.pathsep.join(dict.fromkeys(env))
    old = sys.modules.get('src')
    if old is not None and not _same(os.path.dirname(getattr(old, '__file__', '') or '.'), os.path.join(SHIM_DIR, 'src')):
        raise RuntimeError(f"'src' is already imported from {getattr(old, '__file__', '?')}: call install_shim() before the "
                           "first `import src`")
    import src                                                               # noqa: F401  (binds the name now, shim first)
    if not _same(os.path.dirname(src.__file__), os.path.join(SHIM_DIR, 'src')):
        raise RuntimeError(f'`src` resolved to {src.__file__}, not to the shim')
    ref_src = os.path.join(reference_dir, 'src')
    if not any(_same(p, ref_src) for p in src.__path__):
        raise RuntimeError(f'the shim package does not extend over {ref_src}: src.algorithms / src.tasks would not import')


def provenance(reference_dir: str) -> dict:
    """Where every imported ``src.*`` module came from: {'shim': [...], 'reference': [...], 'reference_hot': [...]}.
    ``reference_hot`` lists modules of the replaced subtrees that were loaded from the CHECKOUT -- must be empty."""
    ref_src = os.path.join(os.path.abspath(reference_dir), 'src') + os.sep
    shim_src = os.path.join(SHIM_DIR, 'src') + os.sep
    out = {'shim': [], 'reference': [], 'reference_hot': [], 'other': []}
    for name, m in sorted(sys.modules.items()):
        if name != 'src' and not name.startswith('src.'):
            continue
        f = os.path.abspath(getattr(m, '__file__', None) or '')
        if f.startswith(shim_src):
            out['shim'].append(name)
        elif f.startswith(ref_src):
            out['reference'].append(name)
            parts = name.split('.')
            if len(parts) > 1 and parts[1] in HOT_SUBTREES:
                out['reference_hot'].append(name)
        else:
            out['other'].append(name)
    return out


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    reference_dir = os.environ.get('HGN_REFERENCE') or os.getcwd()
    probe = False
    while argv and argv[0].startswith('--'):
        opt = argv.pop(0)
        if opt == '--reference':
            reference_dir = argv.pop(0)
        elif opt.startswith('--reference='):
            reference_dir = opt.split('=', 1)[1]
        elif opt == '--probe':
            probe = True
        elif opt == '--':
            break
        else:
            print(f'usage: python -m hgn_amd.run_main [--reference DIR] [--probe] [config name]   (unknown option {opt})',
                  file=sys.stderr)
            return 2
    install_shim(reference_dir)
    reference_dir = os.path.abspath(reference_dir)
    os.chdir(reference_dir)
    script = os.path.join(reference_dir, 'main.py')
    sys.argv = [script] + argv
    if probe:
        # main.py's import block (and everything it pulls in) without its `if __name__ == '__main__'` body
        runpy.run_path(script, run_name='__hgn_probe__')
        import src.model.flag                                                 # noqa: F401
        import src.model.get_model                                            # noqa: F401
        report = provenance(reference_dir)
        report['get_model'] = sys.modules['src.model.get_model'].get_model.__module__
        report['FlagModel'] = sys.modules['src.model.flag'].FlagModel.__module__
        report['src.util'] = sys.modules['src.util'].__file__
        report['MeshSimulator'] = sys.modules['src.algorithms.MeshSimulator'].__file__
        report['torch_scatter_imported'] = 'torch_scatter' in sys.modules
        print(json.dumps(report))
        return 1 if report['reference_hot'] else 0
    runpy.run_path(script, run_name='__main__')
    return 0


if __name__ == '__main__':
    sys.exit(main())
