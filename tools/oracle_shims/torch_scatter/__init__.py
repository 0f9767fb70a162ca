"""Stand-in for the un-vendored third-party ``torch-scatter==2.0.9`` wheel (test tooling only).

The reference imports ``torch_scatter`` at ``src/util.py:5`` and calls ``scatter_add / scatter_max /
scatter_mean / scatter_min`` at ``src/util.py:117-127``.  The wheel is not installed in this image and
cannot be fetched, so this module restates its *published* CPU semantics so that the reference can be
imported to generate golden vectors (tests/golden/gen_golden.py):

* output is zero-initialised with ``dim_size`` rows; empty segments stay 0 for all four ops;
* ``mean`` = sum / max(count, 1);
* ``max`` / ``min`` return ``(values, arg)``; among equal values the first element in iteration order wins
  (the CPU kernel compares with a strict ``>`` / ``<``); ``arg`` of an empty segment is ``src.size(dim)``;
* backward of max/min routes the gradient to the single ``arg`` element.

This is build-owned code, not a copy of torch_scatter.  Parity claims for the scatter primitive itself are
therefore "unpinned" (no reference-side test pins it); everything layered on top is pinned by running the
reference through this stand-in.
"""
import torch


def _check(src, index, dim):
    assert dim == 0, "stand-in only implements dim=0 (the only form the reference uses)"
    assert index.shape == src.shape, "reference pre-broadcasts index to src.shape (src/util.py:107-110)"


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    _check(src, index, dim)
    res = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return res.scatter_add(0, index, src)


scatter_sum = scatter_add


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    _check(src, index, dim)
    total = scatter_add(src, index, 0, None, dim_size)
    count = scatter_add(torch.ones_like(src), index, 0, None, dim_size).clamp_(min=1)
    return total / count


class _ArgReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, dim_size, is_max):
        E = src.shape[0]
        mode = 'amax' if is_max else 'amin'
        init = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        val = init.scatter_reduce(0, index, src, reduce=mode, include_self=False)
        hit = src == val.gather(0, index)
        eid = torch.arange(E, device=src.device).view((E,) + (1,) * (src.dim() - 1)).expand_as(src)
        cand = torch.where(hit, eid, torch.full_like(eid, E))
        arg = torch.full(init.shape, E, dtype=torch.long, device=src.device)
        arg = arg.scatter_reduce(0, index, cand, reduce='amin', include_self=True)
        ctx.save_for_backward(arg)
        ctx.E = E
        ctx.mark_non_differentiable(arg)
        return val, arg

    @staticmethod
    def backward(ctx, gval, _garg):
        (arg,) = ctx.saved_tensors
        E = ctx.E
        g = torch.zeros((E + 1,) + tuple(gval.shape[1:]), dtype=gval.dtype, device=gval.device)
        g.scatter_(0, arg, gval)
        return g[:E], None, None, None


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    _check(src, index, dim)
    return _ArgReduce.apply(src, index, dim_size, True)


def scatter_min(src, index, dim=0, out=None, dim_size=None):
    _check(src, index, dim)
    return _ArgReduce.apply(src, index, dim_size, False)


def scatter_std(src, index, dim=0, out=None, dim_size=None, unbiased=True):
    """torch-scatter 2.0.9, torch_scatter/composite/std.py as published: count = clamp(#elements, 1); mean = sum / count;
    out = scatter_sum((src - mean[index])^2); unbiased: count = clamp(count - 1, 1); out = sqrt(out / (count + 1e-6)).
    A composite of differentiable ops, so autograd gives its backward (NaN where a segment's variance is 0, as in the wheel).
    Unreachable from every reference config (src/util.py:129-130); restated because unsorted_segment_operation accepts it."""
    _check(src, index, dim)
    if out is not None:
        dim_size = out.size(0)
    count = scatter_add(torch.ones_like(src), index, 0, None, dim_size).clamp(1)
    mean = scatter_add(src, index, 0, None, dim_size).div(count)
    var = src - mean.gather(0, index)
    res = scatter_add(var * var, index, 0, None, dim_size)
    if unbiased:
        count = count.sub(1).clamp_(1)
    return res.div(count + 1e-6).sqrt()
