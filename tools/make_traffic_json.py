"""Builds profiles/pmc_traffic.json from two rocprofv3 PMC passes of bench.py (FETCH_SIZE, WRITE_SIZE; separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes): per-launch HBM-side bytes of the big kernels, stamped with the commit and
the sha of the kernel sources they were measured on (bench.py drops the figure when the sources have changed since).
    python tools/make_traffic_json.py FETCH_DIR WRITE_DIR ROWS_PER_LAUNCH NODE_ROWS COMMIT > profiles/pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 bytes)."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def means(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                k = (r['Kernel_Name'].split('(')[0], int(r['Grid_Size']))
                acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    return {k: (s / n, n) for k, (s, n) in acc.items()}


fetch, write = means(sys.argv[1], 'FETCH_SIZE'), means(sys.argv[2], 'WRITE_SIZE')
rows, node_rows = int(sys.argv[3]), int(sys.argv[4])
import bench
out = {'commit': sys.argv[5] if len(sys.argv) > 5 else None, 'kernel_source_sha': bench.kernel_source_sha(),
       'counters': 'FETCH_SIZE x 2 (gfx950: 128-byte requests tallied at 64 bytes) + WRITE_SIZE, KiB, mean over the launches of '
                   'that (kernel, grid) in `bench.py --steps 2 --warmup 1`'}
per_step = 0.0
fused_present = any('edge_bwd_fused' in kk[0] for kk in fetch)
for bench_name, kern in (('mlp_fwd_edge', 'mlp6_fwd_kernel'), ('mlp_bwd_edge', 'mlp6_bwd_kernel'), ('wgrad', 'wgrad6s_kernel'),
                         ('edge_bwd_fused', 'edge_bwd_fused'), ('seg_fwd', 'seg_fwd128_kernel'), ('seg_pair', 'seg_sum_pair128_kernel')):
    if bench_name == 'mlp_bwd_edge' and fused_present:
        continue                                              # only the encoder's backward is left on that kernel at the edge grid
    keys = [k for k in fetch if kern in k[0]]
    if bench_name == 'mlp_fwd_edge' and any('mlp6_fwd_edge_kernel' in k[0] for k in fetch):
        keys = [k for k in fetch if 'mlp6_fwd_edge_kernel' in k[0]]          # the training edge block's own kernel (round 4)
    if not keys:
        continue
    k = max(keys, key=lambda kk: (fetch[kk][1], kk[1]))       # the processor's edge launches: the most frequent (kernel, grid), then the largest
    f_kib, n = fetch[k]
    w_kib = write.get(k, (0.0, 0))[0]
    out[bench_name] = {'kernel': k[0], 'grid_threads': k[1], 'launches_averaged': n, 'rows_per_launch': rows,
                       'FETCH_SIZE_KiB': f_kib, 'WRITE_SIZE_KiB': w_kib,
                       'traffic_bytes_per_launch': (2 * f_kib + w_kib) * 1024,
                       'read_bytes_per_edge_row': 2 * f_kib * 1024 / rows, 'write_bytes_per_edge_row': w_kib * 1024 / rows}
# launches per processor layer and step (profiles/r02_kernel_split.csv): one each; with the fused backward the sender AND the
# receiver sums of dz1 are stand-alone launches (the two-launch backward forms the receiver sums itself)
# (round 4: ONE paired launch, seg_pair, forms both sums; seg_fwd is then absent from the sum configuration)
for bench_name in ('mlp_fwd_edge', 'mlp_bwd_edge', 'wgrad', 'edge_bwd_fused', 'seg_fwd', 'seg_pair'):
    if bench_name in out:
        e = out[bench_name]
        e['launches_per_layer'] = 2 if (bench_name == 'seg_fwd' and fused_present) else 1
        per_step += e['launches_per_layer'] * e['traffic_bytes_per_launch'] / rows
out['edge_level_bytes_per_edge_and_layer'] = per_step
print(json.dumps(out, indent=1))
