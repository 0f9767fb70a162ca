import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'hyper-graph-nets_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """Parity figures recorded by tests.helpers.report -> gpurun_out/parity_report.jsonl (copied into profiles/ by hand)."""
    try:
        from tests import helpers as H
        if H._REPORT:
            import json
            d = os.path.join(ROOT, 'gpurun_out')
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, 'parity_report.jsonl'), 'w') as f:
                for r in H._REPORT:
                    f.write(json.dumps(r) + '\n')
    except Exception:
        pass
