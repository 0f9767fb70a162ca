"""Per-kernel averages of whatever counters a `rocprofv3 --pmc ...` run directory holds, for the kernels of the edge block:
python tools/pmc_kernels.py <dir> [name fragment ...]"""
import csv, glob, sys, collections
d = sys.argv[1]
frags = sys.argv[2:] or ['edge_bwd_fused3_kernel', 'mlp6_fwd_edge_kernel<', 'wgrad6s_kernel', 'seg_sum_pair128', 'seg_fwd128', 'linear6_']
acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        tag = next((x for x in frags if x in r['Kernel_Name']), None)
        if tag is None:
            continue
        k = (tag, r.get('Grid_Size', ''))
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        seen[k].add(r['Dispatch_Id'])
for k, c in sorted(acc.items()):
    n = len(seen[k])
    print(f'{k[0]} grid {k[1]} launches {n}')
    for name, v in sorted(c.items()):
        print(f'    {name:40s} {v / n:18.0f} per launch')
