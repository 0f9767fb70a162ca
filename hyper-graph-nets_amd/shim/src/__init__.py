"""Path shim: a package named ``src`` that resolves the reference's hot-path module paths to the MI355X implementation.

This directory must come BEFORE the reference checkout on ``sys.path``.  ``python main.py`` puts the checkout (the script
directory) first, ahead of PYTHONPATH, so the reference's unchanged ``main.py`` is started through the launcher, which orders
the path from inside the interpreter (hgn_amd/run_main.py):

    cd <reference>; PYTHONPATH=<repo>/hyper-graph-nets_amd python -m hgn_amd.run_main flag

``import src.migration.meshgraphnet`` (flag.py:8), ``src.migration.normalizer`` (flag.py:9), ``src.util`` (flag.py:11),
``src.rmp.get_rmp`` (flag.py:4), ``src.graph_balancer.get_graph_balancer`` (flag.py:47), ``src.model.get_model``
(MeshSimulator.py:109) ... then come from here; everything this shim does NOT provide (src.algorithms, src.tasks, src.data:
trainer, tasks, TFRecord input -- out of scope) falls through to the reference's own ``src`` package,
because the package path is extended over every ``src`` directory on sys.path (shim first).
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
