set -e
for b in 8 16 24 32 64; do
  timeout -k 10 200 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/sweep_b$b.json 2> gpurun_out/sweep_b$b.err
  python -c "import json;d=json.load(open('gpurun_out/sweep_b$b.json'));print($b, d['ms_per_step'], d['value'])"
done
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-stream > gpurun_out/serial_b128.json 2> gpurun_out/serial_b128.err
python -c "
import json;d=json.load(open('gpurun_out/serial_b128.json'));print(d['ms_per_step'], d['value'])
for k,v in d['kernels'].items(): print(k, round(v['ms_per_launch'],3), v['launches_per_step'], round(v['ms_per_launch']*v['launches_per_step'],2))"
