"""Labels fixture for `clustering: hdbscan` (src/rmp/hdbscan.py -> hgn_amd.rmp.HDBSCANClustering).  The reference clusters with the
third-party `hdbscan` wheel, which this image does not have (tools/oracle_shims/hdbscan is an import-only stub that raises): the
reference itself cannot produce labels here.  What is pinned is the scikit-learn port the build uses, on a seeded point cloud with
the reference's flag.yaml arguments -- a scikit-learn upgrade that moves a label is noticed.   python tests/golden/gen_hdbscan_fixture.py"""
import json
import os
import sys

import numpy as np
import sklearn
import sklearn.cluster
from sklearn.preprocessing import StandardScaler

HERE = os.path.dirname(os.path.abspath(__file__))


def cloud(seed=7):
    rng = np.random.RandomState(seed)
    centres = np.array([[0.0, 0.0, 0.0], [3.0, 0.5, -1.0], [-2.0, 2.5, 0.5], [1.0, -3.0, 2.0]])
    pts = np.concatenate([c + 0.25 * rng.randn(40, 3) for c in centres] + [rng.uniform(-5, 5, size=(12, 3))])
    return pts.astype(np.float32)


if __name__ == '__main__':
    X = cloud()
    args = {'max_cluster_size': 50, 'min_cluster_size': 20, 'min_samples': 1}           # configs/flag.yaml:47-51
    labels = sklearn.cluster.HDBSCAN(copy=True, **args).fit(StandardScaler().fit_transform(X)).labels_
    out = {'sklearn': sklearn.__version__, 'args': args, 'seed': 7, 'labels': [int(x) for x in labels]}
    json.dump(out, open(os.path.join(HERE, 'hdbscan_labels.json'), 'w'))
    print(out['sklearn'], 'clusters', int(labels.max()) + 1, 'noise', int((labels < 0).sum()))
