"""Average a PMC counter per kernel name from a rocprofv3 --pmc output directory:  python tools/pmc_summary.py DIR"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].split('(')[0], r['Counter_Name'])
        acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    if 'hgn' in k:
        print('%-40s %-12s launches %4d  mean %.1f' % (k, c, n, s / n))
