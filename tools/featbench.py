"""Measurement of the rows in front of the message-passing path on one MI355X:

  * per-frame latency of FlagModel.build_graph and of build_graph + expand_graph (hyper, K=16) -- what a training
    or rollout loop pays per frame -- next to the CPU oracle (the reference's op sequence) on the host cores;
  * kernel-level HBM rates of the feature kernels on a batch-sized input (64 x 9 282 edges), timed with HIP events.

    python tools/featbench.py [--frames 50] [--no-cpu]
Prints one JSON object.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'hyper-graph-nets_amd'))


def params(connector, K):
    return {'size': 3, 'aggregation': 'sum', 'message_passing_steps': 1,
            'rmp': {'clustering': 'kmeans' if connector != 'none' else 'none', 'connector': connector, 'num_clusters': K,
                    'hyper_noise': 'none', 'hyper_node_features': True, 'frequency': 1, 'fully_connect': False,
                    'intra_cluster_sampling': {'enabled': False, 'alpha': 0.1, 'spotter_threshold': 0}},
            'graph_balancer': {'algorithm': 'none', 'frequency': 1}}


def ev_time(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=50)
    ap.add_argument('--no-cpu', action='store_true')
    a = ap.parse_args()
    from hgn_amd import features, synthetic, system_model
    from hgn_amd.normalizer import Normalizer
    res = {}
    frames = [synthetic.flag_frame(seed=i, nx=40, ny=40) for i in range(4)]
    cells = frames[0]['cells'].cuda()
    dev_frames = [{k: (cells if k == 'cells' else v.cuda()) for k, v in f.items()} for f in frames]

    for name, conn in (('build_graph', 'none'), ('build_graph+expand_graph(hyper,K=16)', 'hyper')):
        model = system_model.FlagModel(params(conn, 16))
        def one(i):
            g = model.build_graph(dev_frames[i % 4], True)
            return model.expand_graph(g, 1 + i, 10 ** 9, True) if conn != 'none' else g
        g = model.build_graph(dev_frames[0], True)
        if conn != 'none':
            model.expand_graph(g, 0, 10 ** 9, True)          # clustering: once per trajectory, not per frame
        for i in range(5):
            one(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.frames):
            one(i)
        torch.cuda.synchronize()
        res[name + ' ms/frame (gpu, eager)'] = (time.perf_counter() - t0) / a.frames * 1e3

    # the whole batch of frames in one go (FlagModel.build_graph_batch)
    model = system_model.FlagModel(params('none', 16))
    Bf = 128
    stacked = {k: (torch.stack([frames[i % 4][k] for i in range(Bf)]).cuda() if k not in ('cells', 'mesh_pos') else dev_frames[0][k])
               for k in frames[0]}
    for _ in range(3):
        model.build_graph_batch(stacked, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        model.build_graph_batch(stacked, True)
    torch.cuda.synchronize()
    res['build_graph_batch (128 frames) ms'] = (time.perf_counter() - t0) / 20 * 1e3

    # kernel-level rates at batch size
    B = 64
    s1, r1, _ = features.cells_to_edges(cells)
    N, E1 = 1600, s1.shape[0]
    off = (torch.arange(B, device='cuda') * N).repeat_interleave(E1)
    s, r = (s1.repeat(B) + off).contiguous(), (r1.repeat(B) + off).contiguous()
    world = torch.cat([f['world_pos'] for f in dev_frames] * (B // 4)).contiguous()
    mesh = torch.cat([f['mesh_pos'] for f in dev_frames] * (B // 4)).contiguous()
    E = s.shape[0]
    t = ev_time(lambda: features.rel_edge_features(world, mesh, s, r, want_len=True))
    bytes_rel = E * (16 + 28 + 4) + world.numel() * 4 + mesh.numel() * 4      # ids + row + length; positions once
    res['rel_edge_features'] = {'edges': E, 'ms': t, 'GB/s': bytes_rel / t / 1e6, 'algorithmic_bytes': bytes_rel}
    feat, _ = features.rel_edge_features(world, mesh, s, r)
    t = ev_time(lambda: features.col_stats(feat))
    res['col_stats'] = {'rows': E, 'ms': t, 'GB/s': feat.numel() * 4 / t / 1e6}
    nz = Normalizer(7, 'e')
    nz(feat)
    t = ev_time(lambda: features.normalize(feat, nz._acc_sum, nz._acc_sum_squared, nz._acc_count, 1e-8))
    res['normalize'] = {'rows': E, 'ms': t, 'GB/s': 2 * feat.numel() * 4 / t / 1e6}
    big = torch.cat([frames[0]['cells'] + i * N for i in range(B)]).cuda()
    t0 = time.perf_counter()
    for _ in range(5):
        features.cells_to_edges(big)
    torch.cuda.synchronize()
    res['cells_to_edges (64 meshes, 194 688 cells) ms'] = (time.perf_counter() - t0) / 5 * 1e3

    if not a.no_cpu:
        from bench import cpu_baseline_features          # the oracle is timed by bench.py's cpu_baseline code only
        res['cpu oracle'] = cpu_baseline_features(frames)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
