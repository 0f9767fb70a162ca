"""Data-parallel training of batches of independent mesh graphs: one process per GPU, RCCL over xGMI.

The reference has no distributed code at all; what must hold is equivalence with the reference's
single-process step on the concatenated batch (MeshSimulator.py:141-152 with FlagModel.training_step, flag.py:146-154):

  * graphs are independent, so the batch shards at graph granularity with NO collective on the data path;
  * the loss is a mean over all NORMAL nodes of the *global* batch: each rank back-propagates its local SUM of squared
    errors, the NORMAL-node count rides in a spare slot behind the gradients, and after the all-reduce everything is scaled
    by 1 / (n_global * out_dim) -- the gradient of the global mean, with no collective between forward and backward;
  * gradients live in ONE flat fp32 buffer (parameter .grad tensors are views into it) and are summed -- together with that
    count -- by a handful of LARGE all-reduces over contiguous ranges of it (default 4 buckets, 9.3 MB in all for the 15-layer
    model: xGMI is point-to-point, ring collectives are per-link bound, so few large collectives beat many small ones), each
    launched as soon as the backward pass has left the layers it covers (last layers first) so that it runs beside the rest
    of the backward pass, and consumed by one fused Adam launch on the flat parameter buffer;
  * Normalizer statistics are sum-reduced across ranks at every accumulate (normalizer.py:53-63) so all replicas
    normalise identically.
"""
from typing import Callable, Iterable, Optional

import torch
import torch.distributed as dist
from torch import nn


def shard_indices(num_graphs: int, rank: int, world: int):
    """Graphs {g : g mod world == rank}."""
    return list(range(rank, num_graphs, world))


class FlatParams:
    """Re-homes every parameter of `module` into one contiguous buffer (and its gradient into a twin buffer).
    Parameter objects keep their identity, names and shapes; only their storage moves."""

    def __init__(self, module: nn.Module):
        params = [p for p in module.parameters() if p.requires_grad]
        lazy = [p for p in params if isinstance(p, nn.parameter.UninitializedParameter)]
        if lazy and not any(isinstance(m, nn.Linear) and not isinstance(m, nn.modules.lazy.LazyModuleMixin) for m in module.modules()):
            raise RuntimeError('materialise the lazy layers (run one forward) before flattening')
        # still lazy after a forward: modules the block schedule never calls (e.g. the `balance` edge model under a hyper block,
        # hypergraphnet.py:54 keeps only its five sets).  They have no gradient and no storage: left alone.
        params = [p for p in params if not isinstance(p, nn.parameter.UninitializedParameter)]
        self.params = params
        self.left_lazy = [n for n, p in module.named_parameters() if isinstance(p, nn.parameter.UninitializedParameter)]
        if self.left_lazy:
            import warnings
            warnings.warn('FlatParams: %d parameter(s) still lazy after the warm-up forward are NOT part of the flat buffer '
                          '(never broadcast, reduced or updated): %s' % (len(self.left_lazy), ', '.join(self.left_lazy[:8]) +
                                                                          (' ...' if len(self.left_lazy) > 8 else '')))
        self._module = module
        self._member = {id(p) for p in params}
        # every parameter starts on a 16-byte boundary: the kernels read biases / LayerNorm affine vectors as float4
        self.offsets = []
        total = 0
        for p in params:
            self.offsets.append(total)
            total += (p.numel() + 3) // 4 * 4
        dev = params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        # one spare 16-byte slot IN FRONT of the gradients: [NORMAL-node count 0 0 0 | grad (total)].  The count travels with the
        # range that holds the FIRST parameters, i.e. with the bucket that is reduced last (the backward pass reaches them last).
        self.grad_ext = torch.zeros(total + 4, dtype=torch.float32, device=dev)
        self.grad = self.grad_ext[4:]
        self.count = self.grad_ext[0:1]
        for p, off in zip(params, self.offsets):
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
            p._hgn_grad = p.grad          # the kernels accumulate straight into this view (ops._grad_targets)
        self.numel = total
        from . import ops
        ops.storage_moved()               # every parameter has a new address: captured forward graphs (graphs.GraphedForward) re-capture

    def check_outsiders(self):
        """A module that was lazy when the buffer was laid out and has been exercised since (a batch with an edge set the warm-up
        graph lacked) owns parameters that no collective and no optimiser step sees: the replicas would diverge silently."""
        if not self.left_lazy:
            return
        bad = [n for n, p in self._module.named_parameters()
               if id(p) not in self._member and p.requires_grad and not isinstance(p, nn.parameter.UninitializedParameter)]
        if bad:
            raise RuntimeError('parameters materialised after FlatParams was built are outside the flat buffer: ' + ', '.join(bad[:8]))

    def zero_grad(self):
        self.grad_ext.zero_()
        for p, off in zip(self.params, self.offsets):     # autograd may have replaced .grad (e.g. after set_to_none)
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + n].view(p.shape)


def _torch_adam(p, g, m, v, lr, b1, b2, eps, step, grad_scale=1.0):
    """Reference-semantics Adam on flat tensors (used only where the HIP kernel cannot run: CPU gloo tests)."""
    g = g * grad_scale
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    p.addcdiv_(m, (v.sqrt() / (bc2 ** 0.5)).add_(eps), value=-lr / bc1)


def _map_tensors(obj, fn):
    """Apply fn to every tensor of a nested output (tensor, list / tuple, dict, or an object with `nodes` / `edges` such as
    modules._Latent); returns a structure of the same kind."""
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(x, fn) for x in obj)
    if isinstance(obj, dict):
        return type(obj)((k, _map_tensors(v, fn)) for k, v in obj.items())
    if hasattr(obj, 'nodes') and hasattr(obj, 'edges'):
        import copy
        out = copy.copy(obj)
        out.nodes, out.edges = _map_tensors(obj.nodes, fn), _map_tensors(obj.edges, fn)
        return out
    return obj


class _BucketMark(torch.autograd.Function):
    """Identity on every tensor that crosses a bucket boundary of the model.  Its backward runs when the gradients of ALL of
    them are ready, i.e. when the backward pass has left every layer behind the boundary: the gradients of those layers'
    parameters are then complete (enqueued on the stream) and their range of the flat buffer can be reduced."""

    @staticmethod
    def forward(ctx, on_backward, *xs):
        ctx.on_backward = on_backward
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for x in xs)

    @staticmethod
    def backward(ctx, *gs):
        ctx.on_backward()
        return (None, *gs)


class _DetachedHook:
    """What a bucket-boundary hook becomes in a pickle / deep copy of the model: the copy belongs to no trainer."""

    def __call__(self, module, inputs, output):
        return None


class _BucketHook:
    """Forward hook at a bucket boundary (DataParallelTrainer._plan_buckets): threads a _BucketMark through the tensors that cross
    it.  A class instead of a closure so that a model with hooks installed still pickles (the reference checkpoints by
    `pickle.dump` of the object that holds the network, MeshSimulator.py:492-493): the hook itself is left out of the pickle."""

    def __init__(self, trainer, start: int):
        self.trainer, self.start = trainer, start

    def __reduce__(self):
        return (_DetachedHook, ())

    def _on_backward(self):
        self.trainer._reduce_from(self.start)

    def __call__(self, module, inputs, output):
        if not (self.trainer.overlap and torch.is_grad_enabled()):
            return None
        tensors = []
        _map_tensors(output, lambda x: tensors.append(x) or x)
        live = [x for x in tensors if x.requires_grad]
        if not live:
            return None
        marked = iter(_BucketMark.apply(self._on_backward, *live))
        return _map_tensors(output, lambda x: next(marked) if x.requires_grad else x)


class DataParallelTrainer:
    """fwd -> global-mean masked MSE -> bwd beside bucketed all-reduces -> fused Adam, on this rank's shard of the batch.

    ``buckets``: number of contiguous ranges the flat gradient buffer is reduced in (world > 1).  Boundaries are put between the
    message-passing blocks of a MeshGraphNet (``model.processor.graphnet_blocks``; ``bucket_after`` names other modules) so
    that the ranges hold about equal numbers of parameters; range k is all-reduced (asynchronously: RCCL's own stream, after
    the gradient kernels enqueued so far) from an autograd node at its boundary, while the backward pass goes on through the
    earlier layers.  The range of the first parameters goes last, after ``backward()`` has returned, and carries the node
    count.  The result is the same sum as one all-reduce of the whole buffer (every element is reduced exactly once)."""

    def __init__(self, model: nn.Module, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 group: Optional[dist.ProcessGroup] = None, adam_fn: Optional[Callable] = None,
                 device_step: bool = False, wgrad_stream: bool = False, buckets: int = 4, bucket_after=None,
                 force_collectives: bool = False):
        self.model = model
        from . import ops as _ops
        self.ctx = getattr(model, '_hgn_ctx', None) or _ops.default_context()      # the launch context of the model it trains (ops.Context)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # (force_collectives: a world of ONE rank still broadcasts / all-reduces -- the N > 1 code path rehearsed on one GPU over RCCL)
        self.collectives = self.world > 1 or (force_collectives and dist.is_available() and dist.is_initialized())
        self.fp = FlatParams(model)
        if self.collectives:                                         # identical replicas: rank 0's weights win
            dist.broadcast(self.fp.flat, src=0, group=group)
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.t = 0
        # device-resident step counter: needed when step() is captured into a HIP graph (graphs.GraphedTrainStep)
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=self.fp.flat.device) if device_step else None
        if adam_fn is None:
            from . import ops
            adam_fn = ops.adam_step
        self.adam_fn = adam_fn
        self.side = torch.cuda.Stream() if (wgrad_stream and self.fp.flat.is_cuda) else None
        self.overlap = True                                        # False: every range after the backward pass (graphs.GraphedShardStep)
        self._pending, self._done_upto = [], None
        self.bucket_starts = self._plan_buckets(max(1, int(buckets)), bucket_after) if self.collectives else []

    # ---- bucket plan ------------------------------------------------------------------------------------------------
    def _plan_buckets(self, n_buckets: int, bucket_after):
        """-> ascending offsets (into grad_ext) at which a new range starts; installs the boundary hooks."""
        if bucket_after is None:
            proc = getattr(getattr(self.model, 'processor', None), 'graphnet_blocks', None)
            blocks = list(proc) if proc is not None else []
            offset_of = {id(p): off for p, off in zip(self.fp.params, self.fp.offsets)}
            firsts = []                                            # (flat offset of a block's first parameter, block before it)
            for i in range(1, len(blocks)):
                ps = [p for p in blocks[i].parameters() if id(p) in offset_of]
                if ps:
                    firsts.append((min(offset_of[id(p)] for p in ps), blocks[i - 1]))
            chosen = []
            for k in range(1, n_buckets):                          # the boundary nearest to k / n_buckets of the buffer
                want = self.fp.numel * k / n_buckets
                if firsts:
                    best = min(firsts, key=lambda f: abs(f[0] - want))
                    if best not in chosen:
                        chosen.append(best)
            boundaries = sorted(chosen, key=lambda f: f[0])
        else:
            offset_of = {id(p): off for p, off in zip(self.fp.params, self.fp.offsets)}
            boundaries = []
            for mod_before, mod_after in bucket_after:            # explicit: (module whose output is the boundary, first module behind it)
                ps = [p for p in mod_after.parameters() if id(p) in offset_of]
                boundaries.append((min(offset_of[id(p)] for p in ps), mod_before))
            boundaries.sort(key=lambda f: f[0])
        starts = []
        for off, module in boundaries:
            start = off + 4                                        # (grad_ext = [count slot | grad])
            starts.append(start)
            module.register_forward_hook(self._make_hook(start))
        return starts

    def _make_hook(self, start: int):
        return _BucketHook(self, start)

    def _reduce_from(self, start: int):
        """All-reduce grad_ext[start : the range already under way), asynchronously."""
        end = self._done_upto if self._done_upto is not None else self.fp.grad_ext.numel()
        if start >= end:
            return
        if self.fp.flat.is_cuda:
            from . import ops
            ops.flush_wgrad(self.ctx)                              # queued node-level weight-gradient tasks of these layers
            if self.side is not None:
                torch.cuda.current_stream().wait_stream(self.side)
        self._pending.append(dist.all_reduce(self.fp.grad_ext[start:end], group=self.group, async_op=True))
        self._done_upto = start

    def step(self, graph, target: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        self.fp.check_outsiders()
        self.fp.zero_grad()
        self._pending, self._done_upto = [], None
        if self.fp.flat.is_cuda:
            from . import ops
            ops.discard_stale_wgrad(self.ctx)                        # tasks a failed backward pass left queued belong to no step
        if self.side is not None:
            from . import ops
            self.side.wait_stream(torch.cuda.current_stream())      # the zeroed gradient buffer is visible to the side stream
            self.ctx.wgrad_stream = self.side
        # (the count rides in the LAST range: written before the backward pass, reduced after it)
        self.fp.count.copy_(mask.sum().to(torch.float32).reshape(1))
        if self.fp.flat.is_cuda:
            ops.begin_step_packs(self.ctx)                            # one launch refreshes every packed weight image of the step
        out = self.model(graph)
        diff = (out - target) * mask.unsqueeze(1).to(out.dtype)       # masked without boolean indexing: no host sync, capturable
        sq = diff.square().sum()                                      # local SUM; scaled to the global mean after the collective
        sq.backward()
        if self.fp.flat.is_cuda:
            ops.end_step_packs(self.ctx)
        if self.side is not None:
            self.ctx.wgrad_stream = None
            torch.cuda.current_stream().wait_stream(self.side)      # join: all weight gradients are in the flat buffer
        return self.reduce_and_update(sq.detach(), out.shape[1])

    def reduce_and_update(self, sq_local: torch.Tensor, width: int) -> torch.Tensor:
        """[local NORMAL-node count | gradients of the local squared-error sum] -> all-reduce (the ranges not yet under way; with
        nothing under way -- captured backward pass -- ONE collective over the whole buffer) -> scale to the gradient of the global mean (flag.py:150-152
        over the whole batch) -> fused Adam.  Returns this rank's share of the global-mean loss."""
        if self.collectives:
            # (no boundary fired -- world of one bucket, or a captured backward pass, graphs.GraphedShardStep, where nothing can
            # be overlapped: ONE collective over the whole buffer; otherwise the range of the first parameters + the count)
            self._reduce_from(0)
            for w in self._pending:
                w.wait()
            self._pending, self._done_upto = [], None
        inv = 1.0 / (self.fp.count * width)
        self.fp.grad.mul_(inv)
        self.t += 1
        if self.t_dev is not None:
            from . import ops
            ops.adam_step_dev(self.fp.flat, self.fp.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.t_dev)
        else:
            self.adam_fn(self.fp.flat, self.fp.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.t)
        if self.fp.flat.is_cuda:
            from . import ops
            self.ctx.invalidate_packs()             # the Adam kernel rewrote the parameters behind torch's version counters
        return (sq_local * inv).squeeze(0)


def attach_normalizer_sync(normalizers: Iterable, group: Optional[dist.ProcessGroup] = None):
    """Make every Normalizer accumulate GLOBAL batch statistics: (count, sum, sum^2) are sum-all-reduced."""
    def reduce_fn(count, data_sum, sq_sum):
        packed = torch.cat([count.reshape(1), data_sum.reshape(-1), sq_sum.reshape(-1)])
        dist.all_reduce(packed, group=group)
        n = data_sum.numel()
        return packed[:1], packed[1:1 + n].view_as(data_sum), packed[1 + n:].view_as(sq_sum)
    for nz in normalizers:
        nz._reduce_fn = reduce_fn
