"""Golden G1 entry for the 'std' operation of unsorted_segment_operation (src/util.py:129-130), produced by running the REFERENCE's own
function (through the torch_scatter import stand-in, whose scatter_std restates torch_scatter/composite/std.py of 2.0.9) next to the
brute-force loops of oracle/scatter_loops.py.  Build container only:

    cd /root/reference && PYTHONDONTWRITEBYTECODE=1 PYTHONHASHSEED=0 \
      PYTHONPATH=/root/repo/tools/oracle_shims:/root/reference \
      python /root/repo/tests/golden/gen_golden_std.py --out /root/repo/tests/golden
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import scatter_loops as SL

from src import util as ref_util                             # noqa: E402  (the reference)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    a = ap.parse_args()
    gen = torch.Generator().manual_seed(21)
    cases = {
        'wide': (torch.randn(97, 128, generator=gen), torch.randint(0, 23, (97,), generator=gen), 25),        # segments 23, 24 empty
        'one_d': (torch.randn(50, generator=gen), torch.randint(0, 7, (50,), generator=gen), 9),
        'narrow_unsorted': (torch.randn(31, 3, generator=gen) * 4 + 1, torch.randint(0, 5, (31,), generator=gen), 5),
    }
    out = {}
    for name, (dat, ids, n) in cases.items():
        # (every segment that exists here has >= 2 distinct values or is empty: a zero-variance segment's gradient is NaN)
        x = dat.clone().requires_grad_(True)
        y = ref_util.unsorted_segment_operation(x, ids, n, 'std')
        w = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
        (y * w).sum().backward()
        o, _, gx = SL.segment_op(dat, ids, n, 'std', w)
        cnt = torch.bincount(ids, minlength=n)
        multi = cnt[ids] >= 2
        assert torch.allclose(y.detach().double(), o.double(), rtol=1e-5, atol=1e-6), name
        assert torch.allclose(x.grad.double()[multi], gx[multi], rtol=1e-4, atol=1e-6), name
        out[name] = {'data': dat, 'ids': ids, 'num_segments': n, 'w': w, 'ref_y': y.detach().clone(), 'ref_gx': x.grad.clone(),
                     'bf_y': o, 'bf_gx': gx, 'count': cnt}
    torch.save(out, os.path.join(a.out, 'g1_segment_std.pt'))
    print('g1_segment_std.pt written:', {k: tuple(v['ref_y'].shape) for k, v in out.items()})


if __name__ == '__main__':
    main()
