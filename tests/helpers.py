"""Shared helpers of the parity tests: run the HIP path and the CPU oracle on identical inputs."""
import os

import torch

from oracle import mgn_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max(|b|_inf, tiny): the '1e-5 relative fp32' metric of BASELINE.json, on the tensor's own scale."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    if a.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def elem_err(a: torch.Tensor, b: torch.Tensor, atol_frac: float = 1e-2) -> float:
    """Element-wise companion of rel_err: max_i |a_i - b_i| / (|b_i| + atol_frac * max|b|), i.e. the smallest rtol for which
    torch.allclose(a, b, rtol, atol = rtol * atol_frac * max|b|) holds.  An element 100x below the tensor's scale is still
    judged relative to 1 % of that scale; fp32 arithmetic (the reference's own included) cannot resolve less."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    if a.numel() == 0:
        return 0.0
    scale = max(float(b.abs().max()), 1e-30)
    return float(((a - b).abs() / (b.abs() + atol_frac * scale)).max())


_REPORT = []


def report(test: str, what: str, got: torch.Tensor, exact: torch.Tensor, ref32: torch.Tensor = None):
    """Record norm-wise and element-wise error of the HIP result (and, when given, of the reference's own fp32 arithmetic =
    the oracle run in fp32) against the fp64 oracle; written to gpurun_out/parity_report.jsonl at the end of the session."""
    rec = {'test': test, 'what': what, 'norm': rel_err(got, exact), 'elem': elem_err(got, exact)}
    if ref32 is not None:
        rec.update(ref_fp32_norm=rel_err(ref32, exact), ref_fp32_elem=elem_err(ref32, exact))
    _REPORT.append(rec)
    print('parity', rec)
    return rec


def worst_grad(grads, grads_exact):
    """(norm-wise, element-wise) worst case over all parameters with a non-zero exact gradient."""
    ks = [k for k in grads_exact if float(grads_exact[k].abs().max()) > 0]
    return (max(rel_err(grads[k], grads_exact[k]) for k in ks), max(elem_err(grads[k], grads_exact[k]) for k in ks))


def graph_from_fixture(fx):
    return O.MultiGraph([x.clone() for x in fx['graph']['node_features']],
                        [O.EdgeSet(n, f.clone(), s.clone(), r.clone()) for n, f, s, r in fx['graph']['edge_sets']])


def oracle_run(sd, graph, arch, agg, target, mask, set_order=None, dtype=torch.float64):
    """fp64 oracle forward + masked-MSE + all gradients (the error budget reference)."""
    sd = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
    nf = [x.detach().clone().to(dtype).requires_grad_(True) for x in graph.node_features]
    es = [O.EdgeSet(e.name, e.features.detach().clone().to(dtype).requires_grad_(True), e.senders, e.receivers)
          for e in graph.edge_sets]
    out = O.mesh_graph_net(sd, O.MultiGraph(nf, es), arch, agg, set_order=set_order)
    loss = O.masked_mse(out, target.to(dtype), mask)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}
    in_grads = {'node': [x.grad for x in nf], 'edge': {e.name: e.features.grad for e in es}}
    return out.detach(), loss.detach(), grads, in_grads


def hip_model(arch, agg, steps, edge_sets, sd, out_size=3, set_order=None):
    import hgn_amd
    model = hgn_amd.MeshGraphNet(output_size=out_size, latent_size=128, num_layers=2, message_passing_aggregator=agg,
                                 message_passing_steps=steps, architecture=arch, edge_sets=list(edge_sets)).to('cuda')
    missing = model.load_state_dict({k: v.to('cuda') for k, v in sd.items()}, strict=True)
    if set_order is not None:
        for blk in model.processor.graphnet_blocks:
            blk.set_order = list(set_order)
    return model


def hip_run(model, graph, target, mask, index_device='cuda'):
    import hgn_amd
    nf = [x.detach().clone().cuda().requires_grad_(True) for x in graph.node_features]
    es = [hgn_amd.EdgeSet(e.name, e.features.detach().clone().cuda().requires_grad_(True),
                          e.senders.to(index_device), e.receivers.to(index_device)) for e in graph.edge_sets]
    model.zero_grad(set_to_none=True)
    out = model(hgn_amd.MultiGraph(nf, es))
    loss = torch.nn.functional.mse_loss(target.cuda()[mask.cuda()], out[mask.cuda()])
    loss.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
             for k, p in model.named_parameters()}
    in_grads = {'node': [x.grad for x in nf], 'edge': {e.name: e.features.grad for e in es}}
    return out.detach(), loss.detach(), grads, in_grads


def digest(grads, seed):
    d = {}
    for k, g in grads.items():
        gen = torch.Generator().manual_seed(seed + (O._name_hash(k) % 100003))
        R = torch.randn((4,) + tuple(g.shape), generator=gen, dtype=torch.float64)
        gg = g.detach().double().cpu()
        d[k] = {'proj': (R * gg).flatten(1).sum(1), 'l2': gg.norm(), 'sum': gg.sum()}
    return d


# ------------------------------------------------------------------------------------------------------------------
# Gate transfer: the fp64 oracle evaluated with the ReLU gates the HIP forward chose.
# At 40 M ReLU inputs a handful sit within fp32 rounding of zero; a gate that falls the other way than in fp64 is not an error
# of the arithmetic (the reference's own fp32 does the same, at other positions) but it moves every gradient upstream of it by
# 1e-4..1e-3.  With the gates transferred, what is left is the arithmetic itself.
# ------------------------------------------------------------------------------------------------------------------
def hip_gates(model, log):
    """ops._GATE_LOG of one training forward -> {oracle MLP prefix: [(mask hidden 1, mask hidden 2), ...]} in oracle row order."""
    prefix_of = {}
    for k, p in model.named_parameters():
        if k.endswith('.layers.linear_0.weight'):
            pre = k[:-len('.layers.linear_0.weight')]
            prefix_of[p.data_ptr()] = pre[:-2] if pre.endswith('.0') else pre
    out = {}
    shifts = torch.arange(32, dtype=torch.int64)
    for ptr, bits, idx in log:
        b = bits.detach().cpu().to(torch.int64) & 0xffffffff                     # [M, 8] words: 4 * layer + q
        M = b.shape[0]
        pair = []
        for layer in (0, 1):
            m = torch.zeros(M, 128, dtype=torch.bool)
            for q in range(4):
                w = ((b[:, 4 * layer + q].unsqueeze(1) >> shifts) & 1).bool()      # bit 4 * blk + u <-> unit 16 * blk + 4 * q + u
                for blk in range(8):
                    m[:, 16 * blk + 4 * q: 16 * blk + 4 * q + 4] = w[:, 4 * blk: 4 * blk + 4]
            if idx is not None:                                                     # row i of the kernel = source row idx[i]
                u = torch.zeros_like(m)
                u[idx.detach().cpu().long()] = m
                m = u
            pair.append(m)
        out.setdefault(prefix_of[ptr], []).append(tuple(pair))
    return out


class GateTransfer:
    """Context manager: inside it the oracle's ReLUs use the given gates; counts where they differ from the oracle's own."""

    def __init__(self, gates):
        self.gates = {k: list(v) for k, v in gates.items()}
        self.flipped, self.total, self.max_abs_at_flip = 0, 0, 0.0

    def __enter__(self):
        self._mlp, self._relu = O.mlp, torch.relu
        state = {'pair': None, 'layer': 0}

        def mlp(sd, prefix, x, layer_norm=True):
            state['pair'] = self.gates[prefix].pop(0)
            state['layer'] = 0
            return self._mlp(sd, prefix, x, layer_norm)

        def relu(x):
            m = state['pair'][state['layer']]
            state['layer'] += 1
            assert m.shape == x.shape, (m.shape, x.shape)
            own = x.detach() > 0
            diff = own != m
            self.total += m.numel()
            n = int(diff.sum())
            if n:
                self.flipped += n
                self.max_abs_at_flip = max(self.max_abs_at_flip, float(x.detach().abs()[diff].max()))
            return x * m.to(x.dtype)
        O.mlp, torch.relu = mlp, relu
        return self

    def __exit__(self, *exc):
        O.mlp, torch.relu = self._mlp, self._relu
        return False


# ------------------------------------------------------------------------------------------------------------------
# Winner transfer: the fp64 oracle evaluated with the max / min WINNERS the HIP forward chose.
# A `pna` block takes, per receiver and feature, the largest and the smallest incoming edge value.  Where two candidates sit within
# fp32 rounding of each other the fp32 evaluation may crown the other one than fp64 does -- like a ReLU gate, a discrete decision
# that moves the gradient routing (all of d(max) goes to ONE edge) without being an error of the arithmetic.  With the HIP
# forward's winners (hgn_segment_reduce_fwd returns them: CSR positions) forced in the oracle, what is left is the arithmetic.
# ------------------------------------------------------------------------------------------------------------------
def hip_winners(model, log):
    """ops._ARG_LOG of one training forward -> {edge-set name: [{'max': arg, 'min': arg}, ...] in block order}; arg [N, 128] holds
    the ORIGINAL edge index of the winner (-1: empty segment)."""
    name_of = {}
    for k, p in model.named_parameters():
        if '.edge_models.' in k and k.startswith('processor.') and k.endswith('.layers.linear_0.weight'):
            name_of[p.data_ptr()] = k.split('.edge_models.')[1].split('.')[0]
    out = {}
    for ptr, amax, amin, perm, off, n_rows in log:      # (arg arrays cover the receiver part's rows [off, off + len) of n_rows node rows)
        p = perm.detach().cpu().long()
        rec = {}
        for op, a in (('max', amax), ('min', amin)):
            if a is not None:
                a = a.detach().cpu().long()
                part = torch.where(a >= 0, p[a.clamp(min=0)], torch.full_like(a, -1))
                rec[op] = torch.full((n_rows, part.shape[1]), -1, dtype=part.dtype)
                rec[op][off:off + part.shape[0]] = part
        out.setdefault(name_of[ptr], []).append(rec)
    return out


class WinnerTransfer:
    """Context manager: inside it the oracle's max / min aggregates take the given winners; counts where they differ from the
    oracle's own and how far the two candidates were apart there (relative to the aggregate's scale)."""

    def __init__(self, winners):
        self.winners = {k: list(v) for k, v in winners.items()}
        self.flipped, self.total, self.max_gap_at_flip = 0, 0, 0.0

    def __enter__(self):
        self._agg = O.aggregation

        def aggregation(edge_sets, features, num_nodes, aggregator):
            for es in edge_sets:
                ops = O.PNA_OPS if aggregator == 'pna' else (aggregator,)
                rec = None
                for op in ops:
                    if op not in ('max', 'min'):
                        features.append(O.segment_reduce(es.features, es.receivers, num_nodes, op))
                        continue
                    if rec is None:
                        rec = self.winners[es.name].pop(0)
                    arg = rec[op]
                    assert arg.shape == (num_nodes, es.features.shape[1]), (arg.shape, num_nodes)
                    own, own_arg = O.segment_reduce(es.features, es.receivers, num_nodes, op, return_arg=True)
                    has = arg >= 0
                    picked = es.features.gather(0, arg.clamp(min=0)) * has.to(es.features.dtype)
                    assert bool(((own_arg < es.features.shape[0]) == has).all()), 'empty segments differ'
                    diff = has & (own_arg != arg)
                    self.total += int(has.sum())
                    n = int(diff.sum())
                    if n:
                        self.flipped += n
                        gap = (picked.detach() - own.detach()).abs()[diff].max() / es.features.detach().abs().max().clamp(min=1e-30)
                        self.max_gap_at_flip = max(self.max_gap_at_flip, float(gap))
                    features.append(picked)
            return torch.cat(features, dim=-1)
        O.aggregation = aggregation
        return self

    def __exit__(self, *exc):
        O.aggregation = self._agg
        return False


def hip_run_logged(model, graph, target, mask):
    """hip_run that also returns the discrete decisions of the HIP forward: (ReLU gates, max / min winners) for the transfers."""
    from hgn_amd import ops
    ops._GATE_LOG, ops._ARG_LOG = [], []
    try:
        res = hip_run(model, graph, target, mask)
        gates, winners = hip_gates(model, ops._GATE_LOG), hip_winners(model, ops._ARG_LOG)
    finally:
        ops._GATE_LOG, ops._ARG_LOG = None, None
    return res + (gates, winners)


def oracle_run_with_hip_decisions(sd, graph, arch, agg, target, mask, gates, winners, set_order=None):
    """The fp64 oracle with the HIP forward's ReLU gates AND max / min winners: -> (out, loss, grads, GateTransfer, WinnerTransfer)."""
    with GateTransfer(gates) as gt, WinnerTransfer(winners) as wt:
        out, loss, grads, _ = oracle_run(sd, graph, arch, agg, target, mask, set_order=set_order)
    assert all(len(v) == 0 for v in gt.gates.values()) and all(len(v) == 0 for v in wt.winners.values()), 'decisions left over'
    return out, loss, grads, gt, wt


# ------------------------------------------------------------------------------------------------------------------
# Conditioning of a test instance, measured on an fp64 oracle run: distance from a ReLU kink / from a max-min tie.
# ------------------------------------------------------------------------------------------------------------------
class KinkMargin:
    """Smallest |x| over every ReLU input of an fp64 oracle run (the distance of the instance from a ReLU kink)."""

    def __enter__(self):
        self.worst = float('inf')
        self._orig = torch.relu

        def probe(x):
            if x.numel():
                self.worst = min(self.worst, float(x.detach().abs().min()))
            return self._orig(x)
        torch.relu = probe
        return self

    def __exit__(self, *exc):
        torch.relu = self._orig
        return False


class TieMargin:
    """Records, over every max/min aggregation of an fp64 oracle run, the smallest lead of a segment's winner over its
    runner-up (relative to the largest magnitude of that aggregation's input)."""

    def __enter__(self):
        self.worst = float('inf')
        self._orig = O.segment_reduce

        def probe(data, segment_ids, num_segments, operation, return_arg=False):
            if operation in ('max', 'min') and data.dim() == 2 and data.shape[0] > 0:
                d = (data if operation == 'max' else -data).detach().double()
                ids = segment_ids.long()
                idx = ids.unsqueeze(1).expand_as(d)
                m1 = torch.full((num_segments, d.shape[1]), float('-inf'), dtype=d.dtype).scatter_reduce(0, idx, d, 'amax')
                top = d == m1[ids]
                ties = torch.zeros(num_segments, d.shape[1], dtype=d.dtype).scatter_add(0, idx, top.double())
                m2 = torch.full_like(m1, float('-inf')).scatter_reduce(0, idx, d.masked_fill(top, float('-inf')), 'amax')
                gap = torch.where(ties > 1, torch.zeros_like(m1), m1 - m2)
                gap = gap[torch.isfinite(gap)]
                if gap.numel():
                    self.worst = min(self.worst, float(gap.min() / d.abs().max().clamp(min=1e-30)))
            return self._orig(data, segment_ids, num_segments, operation, return_arg)
        O.segment_reduce = probe
        return self

    def __exit__(self, *exc):
        O.segment_reduce = self._orig
        return False
