"""Builds profiles/r01_pmc_traffic.json from two rocprofv3 PMC passes of bench.py (FETCH_SIZE, WRITE_SIZE; separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes): per-launch HBM-side bytes of the edge kernels.
    python tools/make_traffic_json.py FETCH_DIR WRITE_DIR ROWS_PER_LAUNCH > profiles/r01_pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 bytes)."""
import collections, csv, glob, json, sys

def means(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                k = (r['Kernel_Name'].split('(')[0], int(r['Grid_Size']))
                acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    return {k: (s / n, n) for k, (s, n) in acc.items()}

fetch, write = means(sys.argv[1], 'FETCH_SIZE'), means(sys.argv[2], 'WRITE_SIZE')
rows = int(sys.argv[3])
out = {}
for bench_name, kern in (('mlp_fwd_edge', 'mlp6_fwd_kernel'), ('mlp_bwd_edge', 'mlp6_bwd_kernel')):
    keys = [k for k in fetch if kern in k[0]]
    if not keys:
        continue
    k = max(keys, key=lambda kk: kk[1])                       # the edge launches have the largest grid
    f_kib, n = fetch[k]
    w_kib = write[k][0]
    out[bench_name] = {'kernel': k[0], 'grid_threads': k[1], 'launches_averaged': n, 'rows_per_launch': rows,
                       'FETCH_SIZE_KiB': f_kib, 'WRITE_SIZE_KiB': w_kib,
                       'traffic_bytes_per_launch': (2 * f_kib + w_kib) * 1024,
                       'read_bytes_per_row': 2 * f_kib * 1024 / rows, 'write_bytes_per_row': w_kib * 1024 / rows}
print(json.dumps(out, indent=1))
