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
# (the machine-readable record bench.py reads -- profiles/sq_counters.json -- is written by tools/sq_round.sh)
