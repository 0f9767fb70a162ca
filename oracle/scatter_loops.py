"""TEST INFRASTRUCTURE -- brute-force statement of the scatter primitive the reference gets from the un-vendored wheel
torch-scatter==2.0.9 (requirements.txt:7; call sites src/util.py:117-127), as plain Python loops over elements.

Nothing here shares code with oracle/mgn_oracle.py::segment_reduce (sort + per-segment scan) or with the import stand-in
tools/oracle_shims/torch_scatter (scatter_reduce + candidate-index trick): all three are checked against each other, and the
values stored in tests/golden/g1_segment_ops.pt under 'bf' were produced by THIS file next to the reference's own outputs.

Semantics restated (torch-scatter 2.0.9 documentation of scatter(..., reduce) and its CPU kernel's behaviour):
  * the output has `dim_size` rows and starts at 0 for sum / mean; max / min start from the lowest / highest representable
    value and rows that received no element are reset to 0 afterwards  =>  EMPTY SEGMENTS YIELD 0 FOR ALL FOUR OPS;
  * mean = sum / max(count, 1);
  * max / min visit the elements in index order e = 0, 1, ... and replace the running value only on a STRICT improvement
    (`src[e] > out` / `src[e] < out`)  =>  AMONG EQUAL VALUES THE FIRST ELEMENT WINS; the returned arg is that element's
    position, or src.size(0) for an empty segment;
  * std (torch_scatter/composite/std.py; src/util.py:129-130, unreachable from the configs): sqrt(sum (x - mean)^2 /
    (max(count - 1, 1) + 1e-6)) with mean = sum / max(count, 1); 0 for an empty segment;
  * backward: sum -> grad_out[index[e]]; mean -> grad_out[index[e]] / max(count, 1); max / min -> grad_out goes to the single
    arg element, nothing to the others.
Only tests/ (and tests/golden/gen_golden.py) import this module.
"""
import math
from typing import List, Tuple


def scatter_forward(src: List[List[float]], index: List[int], dim_size: int, op: str) -> Tuple[List[List[float]], List[List[int]]]:
    """src: E rows of D python floats; index: E segment ids.  -> (out [dim_size][D], arg [dim_size][D]; arg = E when empty)."""
    E = len(src)
    D = len(src[0]) if E else 0
    if op in ('sum', 'mean'):
        out = [[0.0] * D for _ in range(dim_size)]
        cnt = [0] * dim_size
        for e in range(E):
            n = index[e]
            cnt[n] += 1
            row = out[n]
            for d in range(D):
                row[d] += src[e][d]
        if op == 'mean':
            for n in range(dim_size):
                c = max(cnt[n], 1)
                for d in range(D):
                    out[n][d] = out[n][d] / c
        return out, [[E] * D for _ in range(dim_size)]
    if op == 'std':
        # torch_scatter/composite/std.py (2.0.9): count = max(#elements, 1); mean = sum / count; out = sum (x - mean)^2;
        # unbiased: count = max(count - 1, 1); out = sqrt(out / (count + 1e-6)).  Empty segment: sqrt(0 / (1 + 1e-6)) = 0.
        cnt = [0] * dim_size
        tot = [[0.0] * D for _ in range(dim_size)]
        for e in range(E):
            cnt[index[e]] += 1
            for d in range(D):
                tot[index[e]][d] += src[e][d]
        out = [[0.0] * D for _ in range(dim_size)]
        for e in range(E):
            n = index[e]
            c = max(cnt[n], 1)
            for d in range(D):
                dev = src[e][d] - tot[n][d] / c
                out[n][d] += dev * dev
        for n in range(dim_size):
            c = max(max(cnt[n], 1) - 1, 1)
            for d in range(D):
                out[n][d] = math.sqrt(out[n][d] / (c + 1e-6))
        return out, [[E] * D for _ in range(dim_size)]
    if op not in ('max', 'min'):
        raise Exception('Invalid operation type!')
    start = -math.inf if op == 'max' else math.inf
    out = [[start] * D for _ in range(dim_size)]
    arg = [[E] * D for _ in range(dim_size)]
    for e in range(E):
        n = index[e]
        for d in range(D):
            v = src[e][d]
            better = v > out[n][d] if op == 'max' else v < out[n][d]
            if better:                       # strict: a later equal value does not take over
                out[n][d] = v
                arg[n][d] = e
    for n in range(dim_size):
        for d in range(D):
            if arg[n][d] == E:               # nothing arrived
                out[n][d] = 0.0
    return out, arg


def scatter_backward(grad_out: List[List[float]], index: List[int], arg: List[List[int]], op: str, E: int,
                     src: List[List[float]] = None) -> List[List[float]]:
    D = len(grad_out[0]) if grad_out else 0
    g = [[0.0] * D for _ in range(E)]
    if op == 'std':
        # autograd of the composite: d std / d x_e = (x_e - mean) / (std * (count_unbiased + 1e-6)); a segment without variance
        # gives 0 / 0 = NaN (the wheel's sqrt'(0) * 0), an empty one has no elements
        fwd, _ = scatter_forward(src, index, len(grad_out), 'std')
        cnt = [0] * len(grad_out)
        tot = [[0.0] * D for _ in range(len(grad_out))]
        for e in range(E):
            cnt[index[e]] += 1
            for d in range(D):
                tot[index[e]][d] += src[e][d]
        for e in range(E):
            n = index[e]
            c = max(cnt[n], 1)
            cu = max(c - 1, 1) + 1e-6
            for d in range(D):
                den = fwd[n][d] * cu
                g[e][d] = grad_out[n][d] * (src[e][d] - tot[n][d] / c) / den if den != 0.0 else float('nan')
        return g
    if op in ('sum', 'mean'):
        cnt = [0] * len(grad_out)
        for e in range(E):
            cnt[index[e]] += 1
        for e in range(E):
            n = index[e]
            scale = 1.0 if op == 'sum' else 1.0 / max(cnt[n], 1)
            for d in range(D):
                g[e][d] = grad_out[n][d] * scale
        return g
    for n in range(len(grad_out)):
        for d in range(D):
            e = arg[n][d]
            if e < E:
                g[e][d] += grad_out[n][d]
    return g


def segment_op(data, segment_ids, num_segments: int, op: str, weight=None):
    """Tensor front end (any trailing shape incl. 1-D, like src/util.py:92-134): -> (out, arg, grad of sum(out * weight))
    computed in float64 python arithmetic; `out` is rounded to data's dtype like util.py:133."""
    import torch
    E = data.shape[0]
    D = 1
    for n in data.shape[1:]:
        D *= int(n)
    flat = data.detach().double().reshape(E, D)
    src = flat.tolist()
    idx = [int(i) for i in segment_ids.reshape(segment_ids.shape[0], -1)[:, 0].tolist()] if segment_ids.dim() > 1 else \
        [int(i) for i in segment_ids.tolist()]
    if E == 0:
        src = []
    out, arg = scatter_forward(src, idx, num_segments, op) if E else ([[0.0] * flat.shape[1] for _ in range(num_segments)],
                                                                      [[0] * flat.shape[1] for _ in range(num_segments)])
    shape = (num_segments,) + tuple(data.shape[1:])
    o = torch.tensor(out, dtype=torch.float64).reshape(shape) if num_segments else torch.zeros(shape, dtype=torch.float64)
    a = torch.tensor(arg, dtype=torch.long).reshape(shape) if num_segments else torch.zeros(shape, dtype=torch.long)
    gx = None
    if weight is not None:
        w = weight.detach().double().reshape(num_segments, -1).tolist()
        g = scatter_backward(w, idx, arg, op, E, src)
        gx = torch.tensor(g, dtype=torch.float64).reshape(data.shape) if E else torch.zeros(data.shape, dtype=torch.float64)
    return o.to(data.dtype), a, gx
