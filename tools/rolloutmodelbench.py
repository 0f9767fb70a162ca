"""Row f4 through the reference's own API: FlagModel.rollout (flag.py:192-246) on a 40x40 flag mesh, per-step wall time with the
network replayed from a HIP graph (default: graphs.GraphedForwardCache behind AbstractSystemModel.forward) and with every launch
eager (`model.replay_rollout = False`).  Per step: build_graph (feature kernels, normalisers) + expand_graph + network + update.
    python tools/rolloutmodelbench.py [--steps 100]        prints one JSON object"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'hyper-graph-nets_amd')):
    sys.path.insert(0, p)
import torch


def params(connector, K, steps, agg):
    return {'size': 3, 'aggregation': agg, 'message_passing_steps': steps,
            'rmp': {'clustering': 'kmeans' if connector != 'none' else 'none', 'connector': connector, 'num_clusters': K,
                    'hyper_noise': 'none', 'hyper_node_features': True, 'frequency': 1, 'fully_connect': False,
                    'intra_cluster_sampling': {'enabled': False, 'alpha': 0.1, 'spotter_threshold': 0}},
            'graph_balancer': {'algorithm': 'none', 'frequency': 1}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=100)
    a = ap.parse_args()
    from hgn_amd import synthetic, system_model
    res = {}
    for name, connector, K, layers, agg in (('flag none/sum/L15', 'none', 0, 15, 'sum'), ('flag hyper/pna/L5/K16', 'hyper', 16, 5, 'pna')):
        frames = [synthetic.flag_frame(seed=100 + i, nx=40, ny=40) for i in range(2)]
        traj = {k: torch.stack([frames[i % 2][k] for i in range(a.steps)]).cuda() for k in frames[0]}
        torch.manual_seed(0)
        model = system_model.FlagModel(params(connector, K, layers, agg))
        f0 = {k: v.cuda() for k, v in frames[0].items()}
        model.build_graph(f0, True); model.get_target(f0, True)
        model.evaluate()
        out = {}
        for replay in (True, False):
            model.replay_rollout = replay
            model.rollout(traj, 8)                           # warm: lazy layers, topology, capture
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pred, _ = model.rollout(traj, a.steps)
            torch.cuda.synchronize()
            out['replayed' if replay else 'eager'] = (time.perf_counter() - t0) / a.steps * 1e3
            out.setdefault('pred', []).append(pred['pred_pos'])
        same = bool(torch.equal(out['pred'][0], out['pred'][1]))
        res[name] = {'ms_per_rollout_step_network_replayed': out['replayed'], 'ms_per_rollout_step_all_eager': out['eager'],
                     'predictions_bit_identical': same, 'steps': a.steps, 'nodes': 1600}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
