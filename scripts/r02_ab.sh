set -u
mkdir -p gpurun_out/ab1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "conv or rpn_heads or head" 2>&1 | tail -3
for dt in f32 bf16 f32s; do
  timeout 300 python bench.py --dtype $dt --steps 50 --warmup 5 --no-cpu-baseline --no-split-variant > gpurun_out/ab1/bench_$dt.json 2> gpurun_out/ab1/bench_$dt.err; echo "$dt rc=$?"
  python - <<P
import json
d=json.load(open('gpurun_out/ab1/bench_$dt.json'))
print('$dt', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms; conv chain', round(d['roofline'].get('conv_ms_per_image',0),3), 'frac', round(d['roofline']['frac'],3))
print({k: v for k, v in d['stages_ms'].items()})
P
done
for dt in f32 f32s; do
  timeout 300 python bench.py --mode train --dtype $dt --steps 10 --warmup 3 > gpurun_out/ab1/train_$dt.json 2> gpurun_out/ab1/train_$dt.err; echo "train $dt rc=$?"; cut -c1-260 gpurun_out/ab1/train_$dt.json
done
