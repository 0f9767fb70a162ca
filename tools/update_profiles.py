"""Copies one tools/measure_round.sh result directory into profiles/ (the files profiles/README_r02.md lists):
    python tools/update_profiles.py gpurun_out/r02g <commit> [round tag, default r02]"""
import json, os, shutil, subprocess, sys
O, C = sys.argv[1], sys.argv[2]
R = sys.argv[3] if len(sys.argv) > 3 else 'r03'
subprocess.run(f'python tools/make_traffic_json.py {O}/pmc_f {O}/pmc_w 1188096 204800 {C} > /tmp/pmc_traffic.json', shell=True, check=True)
shutil.copy('/tmp/pmc_traffic.json', 'profiles/pmc_traffic.json'); shutil.copy('/tmp/pmc_traffic.json', f'profiles/{R}_pmc_traffic.json')
for a, b in (('kernel_split.csv', f'{R}_kernel_split.csv'), ('kernel_stats.csv', f'{R}_kernel_stats.csv'),
             ('bench_under_rocprof.json', f'{R}_bench_under_rocprof.json'), ('bench_default.json', f'{R}_bench_default.json')):
    shutil.copy(f'{O}/{a}', f'profiles/{b}')


def last_json(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith('{'):
            return json.loads(line)


out = {}
for n in ('pna', 'hyper', 'plate', 'cylinder_fp16', 'b1', 'b8', 'b21', 'b64', 'b256', 'eager', 'bf16x3', 'bf16', 'fp16', 'two_launch_bwd'):
    if os.path.exists(f'{O}/bench_{n}.json'):
        d = last_json(f'{O}/bench_{n}.json')
        if d is None:
            continue
        out[n] = {'ms_per_step': d['ms_per_step'], 'edges_per_s': d['value'], 'edges_per_step': d['config'].get('edges_per_step'),
                  'steps': d['steps'], 'workload': d['config']['workload'], 'graphs_per_gpu': d['config'].get('graphs_per_gpu')}
for tag, f in (('gpus2_gloo_one_gpu_rehearsal', 'bench_gpus2_gloo.json'), ('gpus2_gloo_one_gpu_rehearsal_strong_1_graph_per_rank', 'bench_gpus2_gloo_strong.json'),
               ('gpus2_gloo_one_gpu_rehearsal_eager_4_buckets', 'bench_gpus2_gloo_eager_buckets.json')):
    if os.path.exists(f'{O}/{f}'):
        d = last_json(f'{O}/{f}')
        out[tag] = {'ms_per_step': d['ms_per_step'], 'edges_per_s': d['value'], 'n_gpus': d['n_gpus'], 'scaling': d['scaling'], 'config': d['config'], 'collective': d.get('collective')}
out['_note'] = ('two_launch_bwd = HGN_NO_FUSED_BWD=1 (the round-1 edge backward), fp16 / bf16 = '
                f'--precision; everything else the defaults; one box, tools/measure_round.sh at commit {C}')
json.dump(out, open(f'profiles/{R}_other_configs.json', 'w'), indent=1)
dp = {'_note': 'bench.py --dp-rehearsal --no-prof --steps 20 --warmup 5: ONE rank through the N>1 code path over a one-rank RCCL communicator '
               '(GraphedShardStep replay / eager with bucketed all-reduces; collective + scaling + Adam eager)'}
for tag, n in (('captured_forward_backward_one_collective', 'graph'), ('eager_4_buckets_overlapped', 'eager'), ('eager_1_bucket', 'eager1')):
    if os.path.exists(f'{O}/bench_dp_rehearsal_{n}.json'):
        d = last_json(f'{O}/bench_dp_rehearsal_{n}.json')
        dp[tag] = {k: d[k] for k in ('ms_per_step', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'scaling', 'collective')}
        dp[tag]['parallelism'] = d['config']['parallelism']
if len(dp) > 1:
    json.dump(dp, open(f'profiles/{R}_dp_rehearsal.json', 'w'), indent=1)
if os.path.exists(f'{O}/rollout.json'):
    json.dump(last_json(f'{O}/rollout.json'), open(f'profiles/{R}_rollout.json', 'w'), indent=1)
if os.path.exists(f'{O}/rollout_kernels.csv'):
    shutil.copy(f'{O}/rollout_kernels.csv', f'profiles/{R}_rollout_kernels_latency_form.csv')
t = json.load(open('profiles/pmc_traffic.json'))
print('kernel sources', t['kernel_source_sha'], 'edge-level bytes per edge and layer', round(t['edge_level_bytes_per_edge_and_layer']))
