#!/bin/bash
# Measurement set of a round:  bash tools/measure_round.sh r04 <commit> [A|B|all]   (two gpurun calls of <= 1200 s: A, then B)
# A: default bench line, kernel trace + per-launch split, PMC traffic passes (separate runs), SQ counter passes of the edge kernels.
# B: other configurations, N=2 rehearsals, data-parallel rehearsal, rollout.
R=${1:-r04}; COMMIT=${2:-unknown}; PART=${3:-all}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$R; mkdir -p $O
if [ "$PART" != "B" ]; then
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default done rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-cold --no-secondary > $O/bench_under_rocprof.json 2> $O/trace.err; echo "trace done rc=$?"
python tools/kernel_split.py $O/trace > $O/kernel_split.csv
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cold --no-secondary > $O/pmc_f.json 2> $O/pmc_f.err; echo "pmc fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cold --no-secondary > $O/pmc_w.json 2> $O/pmc_w.err; echo "pmc write rc=$?"
python tools/make_traffic_json.py $O/pmc_f $O/pmc_w 1188096 204800 $COMMIT > $O/pmc_traffic.json; echo "traffic json rc=$?"
rm -rf $O/trace/*/*.db $O/pmc_f/*/*.db $O/pmc_w/*/*.db 2>/dev/null
bash tools/sq_round.sh $R $COMMIT; echo "sq rc=$?"
python -c "import json;d=json.load(open('$O/bench_default.json'));print('default', d['ms_per_step'], d['value'], json.dumps(d['roofline']), json.dumps(d.get('cpu_baseline')), json.dumps(d.get('cold_step')), json.dumps(d.get('secondary')))"
fi
if [ "$PART" = "A" ]; then exit 0; fi
for cfg in "pna --agg pna" "hyper --arch hyper --agg pna --layers 5 --clusters 16" "plate --workload plate --arch hetero --agg pna --layers 5 --clusters 31" "cylinder_fp16 --workload cylinder --arch hyper --agg pna --layers 25 --clusters 16 --precision fp16 --no-prof" "b1 --batch 1" "b8 --batch 8" "b21 --batch 21" "b64 --batch 64" "b256 --batch 256" "eager --eager" "bf16x3 --precision fp32-bf16x3" "bf16 --precision bf16" "fp16 --precision fp16"; do
  set -- $cfg; n=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-cold --no-secondary --steps 20 --warmup 5 "$@" > $O/bench_$n.json 2> $O/bench_$n.err || echo "config $n failed"
  python -c "import json;d=json.load(open('$O/bench_$n.json'));print('$n', round(d['ms_per_step'],2), round(d['value']/1e6,2))" || true
done
HGN_NO_FUSED_BWD=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-cold --no-secondary --steps 20 --warmup 5 > $O/bench_two_launch_bwd.json 2> $O/bench_two_launch_bwd.err
for n in two_launch_bwd; do python -c "import json;d=json.load(open('$O/bench_$n.json'));print('$n', round(d['ms_per_step'],2), round(d['value']/1e6,2))" || true; done
timeout -k 10 600 python bench.py --gpus 2 --backend gloo --no-cold --no-secondary --steps 10 --warmup 3 --batch 64 > $O/bench_gpus2_gloo.json 2> $O/bench_gpus2_gloo.err; echo "gpus2 rc=$?"
timeout -k 10 600 python bench.py --gpus 2 --backend gloo --no-cold --no-secondary --steps 10 --warmup 3 --global-batch 2 > $O/bench_gpus2_gloo_strong.json 2> $O/bench_gpus2_gloo_strong.err; echo "gpus2 strong rc=$?"
timeout -k 10 600 python bench.py --gpus 2 --backend gloo --eager --no-cold --no-secondary --steps 10 --warmup 3 --batch 16 > $O/bench_gpus2_gloo_eager_buckets.json 2> $O/bench_gpus2_gloo_eager_buckets.err; echo "gpus2 eager rc=$?"
for v in "graph" "eager --eager" "eager1 --eager --buckets 1"; do
  set -- $v; n=$1; shift
  timeout -k 10 300 python bench.py --dp-rehearsal --no-prof --steps 20 --warmup 5 "$@" > $O/bench_dp_rehearsal_$n.json 2> $O/bench_dp_rehearsal_$n.err; echo "dp rehearsal $n rc=$?"
done
timeout -k 10 300 python tools/rolloutbench.py > $O/rollout.json 2> $O/rollout.err; echo "rollout rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rollout_trace -- python3 tools/rolloutbench.py --no-cpu --steps 20 > /dev/null 2> $O/rollout_trace.err; python tools/kernel_split.py $O/rollout_trace > $O/rollout_kernels.csv
rm -rf $O/rollout_trace/*/*.db 2>/dev/null
