#!/bin/bash
# Round-end measurement set (one gpurun call): default bench line, kernel trace, PMC traffic passes, other configurations.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_f.json 2> $O/pmc_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_w.json 2> $O/pmc_w.err
python tools/make_traffic_json.py $O/pmc_f $O/pmc_w 1188096 > profiles/r01_pmc_traffic.json
cp profiles/r01_pmc_traffic.json $O/
echo "pmc done"
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "default done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/trace.err
echo "trace done"
for cfg in "pna --agg pna" "hyper --arch hyper --agg pna --layers 5 --clusters 16" "hetero --arch hetero --agg pna --layers 5 --clusters 31 --world-edges 300" "b1 --batch 1" "b21 --batch 21" "b64 --batch 64" "b256 --batch 256" "side --side-stream" "eager --eager"; do
  set -- $cfg; n=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $O/bench_$n.json 2> $O/bench_$n.err || echo "config $n failed"
  python -c "import json;d=json.load(open('$O/bench_$n.json'));print('$n', round(d['ms_per_step'],2), round(d['value']/1e6,2))" || true
done
HGN_FP32_MFMA=1 python bench.py --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err
python -c "import json;d=json.load(open('$O/bench_default.json'));print('default', d['ms_per_step'], d['value'], json.dumps(d['roofline']), json.dumps(d.get('cpu_baseline')))"
