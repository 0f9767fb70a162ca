"""Row f4: the rollout regime -- one flag_simple-shape graph, forward only, strictly sequential steps
(FlagModel.rollout/_step_fn, src/model/flag.py:193-246).  Latency per step of the 15-layer MeshGraphNet forward with every kernel
launched from the host, replayed from one HIP graph, and the CPU oracle (the reference's op sequence) on the host cores.
    python tools/rolloutbench.py [--steps 200] [--no-cpu]        prints one JSON object"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--no-cpu', action='store_true')
    a = ap.parse_args()
    import hgn_amd
    from hgn_amd import synthetic, graphs
    dev = torch.device('cuda')
    res = {}
    for arch, agg, layers, clusters in (('none', 'sum', 15, 0), ('hyper', 'pna', 5, 16)):
        g = synthetic.grid_graph(seed=7, nx=40, ny=40, clusters=clusters)
        graph = hgn_amd.MultiGraph([x.to(dev) for x in g.node_features],
                                   [hgn_amd.EdgeSet(e.name, e.features.to(dev), e.senders.to(dev), e.receivers.to(dev)) for e in g.edge_sets])
        torch.manual_seed(0)
        model = hgn_amd.MeshGraphNet(3, 128, 2, agg, layers, arch, [e.name for e in graph.edge_sets]).to(dev).eval()
        with torch.no_grad():
            ref = model(graph).clone()
            for _ in range(5):
                model(graph)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                out = model(graph)
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / a.steps * 1e3
        gf = graphs.GraphedForward(model, graph)
        nf = [x for x in graph.node_features]
        ef = {e.name: e.features for e in graph.edge_sets}
        for _ in range(5):
            gf(nf, ef)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = gf(nf, ef)                      # features copied in every step, as a rollout would
        torch.cuda.synchronize()
        replay = (time.perf_counter() - t0) / a.steps * 1e3
        assert torch.equal(out, ref)
        key = f'{arch}/{agg}/L{layers}' + (f'/K{clusters}' if clusters else '')
        res[key] = {'edges': int(sum(e.senders.shape[0] for e in g.edge_sets)), 'eager_ms_per_step': eager,
                    'hip_graph_ms_per_step': replay, 'replay_bit_identical_to_eager': True}
        if not a.no_cpu and arch == 'none':
            from bench import cpu_baseline_forward          # the oracle is timed by bench.py's cpu_baseline code only
            ms, cores, o_cpu = cpu_baseline_forward(model.state_dict(), g, arch, agg)
            res[key]['cpu_oracle_ms_per_step'] = ms
            res[key]['cpu_threads'] = cores
            res[key]['rel_err_vs_cpu_oracle'] = float((ref.cpu() - o_cpu).abs().max() / o_cpu.abs().max())
    print(json.dumps(res))


if __name__ == '__main__':
    main()
