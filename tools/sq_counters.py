"""Per-kernel SQ counter summary of a `rocprofv3 --pmc SQ_...` run directory: where the waves' cycles go (parked on s_waitcnt /
barriers, stalled at issue, issuing).  python tools/sq_counters.py <dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for f in files:
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].split('(')[0][:60], r.get('Grid_Size', r.get('Grid_Size_X', '')))
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (k, r['Dispatch_Id'])
        if key not in seen:
            seen.add(key); cnt[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:8]:
    wc = c.get('SQ_WAVE_CYCLES', 1.0)
    print(f'{k[0]} grid {k[1]} launches {cnt[k]}')
    for name, v in sorted(c.items()):
        print(f'    {name:28s} {v / cnt[k]:16.0f} per launch   {v / wc:7.3f} of SQ_WAVE_CYCLES')

# --json OUT COMMIT: machine-readable record for bench.py (`roofline.counters`), stamped like profiles/pmc_traffic.json
if '--json' in sys.argv:
    import json, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    i = sys.argv.index('--json')
    out_path, commit = sys.argv[i + 1], (sys.argv[i + 2] if len(sys.argv) > i + 2 else None)
    names = {'edge_bwd_fused_kernel<6>': 'edge_bwd_fused', 'mlp6_fwd_edge_kernel<6>': 'mlp_fwd_edge', 'mlp6_fwd_kernel<2, 6, 4>': 'mlp_fwd_edge_general'}
    rec = {'commit': commit, 'kernel_source_sha': bench.kernel_source_sha(), 'workload': 'tools/fusedbench.py: 128 flag_simple-shape graphs, 1 188 096 edge rows',
           'units': 'SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* tick once per 4 cycles per wave; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over the 1 024 SIMDs; '
                    'SQ_INSTS_VALU includes the MFMA instructions', 'kernels': {}}
    for k, c in acc.items():
        tag = next((v for kk, v in names.items() if kk in k[0]), None)
        if tag is None or 'SQ_WAVE_CYCLES' not in c:
            continue
        n = cnt[k]
        # SQ_WAVE_CYCLES is collected in every pass: average it over the passes it appears in
        e = {name: v / n for name, v in c.items()}
        wc = e['SQ_WAVE_CYCLES']
        d = {'per_launch': e, 'grid': k[1]}
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in e and 'SQ_BUSY_CYCLES' in e:
            pass
        rec['kernels'][tag] = d
    json.dump(rec, open(out_path, 'w'), indent=1)
