#!/bin/bash
# One gpurun call at a round's end: the whole GPU suite, the driver's bench command, kernel splits of the small-batch and pna
# configurations (launch counts behind DESIGN 5.4 / 10-5):  bash tools/confirm_round.sh r05
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${R}_confirm; mkdir -p $O
timeout -k 10 560 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -3 $O/gputests.log
[ $rc -eq 0 ] || exit $rc
python bench.py > $O/bench_default.json 2> $O/bench_default.err && echo "default done" || exit 1
for cfg in "b1 --batch 1" "b21 --batch 21" "pna --agg pna"; do
  set -- $cfg; n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$n -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-cold --no-secondary "$@" > $O/bench_under_rocprof_$n.json 2> $O/trace_$n.err && echo "trace $n done" || exit 1
  python tools/kernel_split.py $O/trace_$n > $O/kernel_split_$n.csv
  rm -rf $O/trace_$n/*/*.db 2>/dev/null
done
python -c "import json;d=json.load(open('$O/bench_default.json'));print('default', d['ms_per_step'], d['value'], json.dumps(d['roofline'])[:400])"
