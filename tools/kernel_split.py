"""Per-launch split of a rocprofv3 --kernel-trace run: (kernel, grid size) -> calls, avg / min / max duration in microseconds.
The stats CSV of rocprofv3 lumps edge-level and node-level launches of one kernel template together; this keeps them apart, so
the dominant kernel's duration is readable from profiles/ alone.
    python tools/kernel_split.py TRACE_DIR > profiles/rNN_kernel_split.csv"""
import collections, csv, glob, sys

acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        grid = int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0)
        wg = int(r.get('Workgroup_Size') or r.get('Workgroup_Size_X') or 0)
        acc[(name, grid, wg)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
total = sum(sum(v) for v in acc.values())
w = csv.writer(sys.stdout)
w.writerow(['kernel', 'grid_threads', 'workgroup', 'calls', 'avg_us', 'min_us', 'max_us', 'total_ms', 'share'])
for (name, grid, wg), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([name, grid, wg, len(v), f'{sum(v) / len(v):.2f}', f'{min(v):.2f}', f'{max(v):.2f}', f'{sum(v) / 1e3:.3f}', f'{sum(v) / total:.4f}'])
