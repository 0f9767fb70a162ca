#!/bin/bash
# SQ counter passes of the two edge kernels (one rocprofv3 --pmc pass per counter set, kernel trace only: never combined with other
# trace domains): bash tools/sq_round.sh r04 <commit>   -> gpurun_out/<tag>/sq_counters.txt, sq_counters.json
R=${1:-r04}; COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$R; mkdir -p $O
i=0
for set in "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/sq$i -- python3 tools/fusedbench.py --iters 3 > $O/sq$i.log 2> $O/sq$i.err; echo "sq pass $i rc=$?"
done
{
  echo "# rocprofv3 --kernel-trace --pmc <set> -- python3 tools/fusedbench.py --iters 3   (four separate passes, one counter set each; tools/sq_counters.py)"
  echo "# 128 flag_simple-shape graphs, 1 188 096 edge rows; per launch.  SQ_WAVE_CYCLES and the SQ_WAIT_* / SQ_ACTIVE_* counters tick once per 4 cycles per wave;"
  echo "# SQ_VALU_MFMA_BUSY_CYCLES is in cycles summed over the 1 024 SIMDs: = SQ_INSTS_MFMA x 16 (v_mfma_f32_16x16x32_f16 / _bf16);  SQ_INSTS_VALU INCLUDES the MFMA instructions"
  for j in 1 2 3 4; do python tools/sq_counters.py $O/sq$j | grep -A6 "edge_bwd_fused3_kernel\|edge_bwd_fused_kernel<6>\|mlp6_fwd_edge_kernel<"; done
} > $O/sq_counters.txt
python - "$O" "$COMMIT" <<'PY'
import csv, glob, json, os, sys, collections
O, commit = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.getcwd())
import bench
names = {'edge_bwd_fused3_kernel': 'edge_bwd_fused', 'edge_bwd_fused_kernel<6>': 'edge_bwd_fused', 'mlp6_fwd_edge_kernel<': 'mlp_fwd_edge'}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + '/sq*/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        tag = next((v for k, v in names.items() if k in r['Kernel_Name']), None)
        if tag:
            per[(tag, r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
    for (tag, _), c in per.items():
        for n, v in c.items():
            acc[tag][n].append(v)
rec = {'commit': commit, 'kernel_source_sha': bench.kernel_source_sha(), 'workload': 'tools/fusedbench.py: 128 flag_simple-shape graphs, 1 188 096 edge rows',
       'units': 'per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* tick once per 4 cycles per wave; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over the 1 024 SIMDs; SQ_INSTS_VALU includes the MFMA instructions',
       'kernels': {}}
for tag, c in acc.items():
    e = {n: sum(v) / len(v) for n, v in c.items()}
    wc = e.get('SQ_WAVE_CYCLES', 0.0)
    d = dict(e)
    if wc:
        d['waves_issuing_frac'] = e.get('SQ_ACTIVE_INST_ANY', 0) / wc
        d['waves_stalled_at_issue_frac'] = e.get('SQ_WAIT_INST_ANY', 0) / wc
        d['waves_parked_frac'] = e.get('SQ_WAIT_ANY', 0) / wc
    if 'SQ_BUSY_CYCLES' in e and 'SQ_VALU_MFMA_BUSY_CYCLES' in e:
        # SQ_BUSY_CYCLES: per SE-summed busy cycles of the SQs (32 per chip); the kernel's duration in cycles = SQ_BUSY_CYCLES / 32
        dur = e['SQ_BUSY_CYCLES'] / 32.0
        d['kernel_cycles'] = dur
        d['matrix_pipe_busy_frac'] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * dur)
    if 'SQ_INSTS_VALU' in e and 'SQ_INSTS_MFMA' in e:
        d['vector_instructions_without_mfma'] = e['SQ_INSTS_VALU'] - e['SQ_INSTS_MFMA']
    d['summary'] = ('matrix pipe %.0f %% busy, waves issuing %.0f %% / stalled at issue %.0f %% / parked %.0f %% of their cycles' % (
        100 * d.get('matrix_pipe_busy_frac', float('nan')), 100 * d.get('waves_issuing_frac', float('nan')),
        100 * d.get('waves_stalled_at_issue_frac', float('nan')), 100 * d.get('waves_parked_frac', float('nan'))))
    rec['kernels'][tag] = d
json.dump(rec, open(O + '/sq_counters.json', 'w'), indent=1)
print(json.dumps({k: v['summary'] for k, v in rec['kernels'].items()}, indent=1))
PY
rm -rf $O/sq*/*/*.db 2>/dev/null
