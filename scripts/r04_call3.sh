#!/bin/bash
# Round 4, call 3: the fused conv1 pair launch -- word-by-word and timed against the two launches it replaces (torch-free), its GPU tests, the bf16
# line with and without it, and the stage-2 step with device-drawn dropout masks.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r04c; mkdir -p $O; B=scripts/micro/_bin
{ for rw in 6 4; do echo "== RW $rw"; FRCNN_BF16_PAIR_RW=$rw timeout 60 $B/conv_pair_micro; done; } > $O/r04_conv_pair_micro.txt 2>&1; cat $O/r04_conv_pair_micro.txt
timeout 900 python -m pytest tests -m gpu -q -s --timeout 600 -k "conv1_pair or vgg16_forward_600x1000_bf16 or staging or vgg16_bf16" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_sel.log | tail -2; grep -E "^PARITY conv1" $O/pytest_sel.log
for f in 1 0; do FRCNN_BF16_CONV1_PAIR=$f timeout 600 python bench.py --dtype bf16 --steps 100 --warmup 5 --no-cpu-baseline > $O/r04_bench_bf16_pair$f.json 2> $O/bench_pair$f.err; echo "bench bf16 pair=$f rc=$?"
python - "$O/r04_bench_bf16_pair$f.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
    print(round(d["value"], 1), "img/s", round(d["ms_per_step"], 4), "ms; conv", round(d["roofline"]["conv_ms_per_image"], 4), "ms frac", round(d["roofline"]["frac"], 4), {k: v for k, v in d["stages_ms"].items() if k.startswith("conv1") or k.startswith("conv2")})
except Exception as e:
    print("no line:", e)
PY
done
timeout 600 python bench.py --mode train-rcnn --steps 20 --warmup 3 > $O/r04_bench_train_rcnn.json 2> $O/train_rcnn.err; echo "train-rcnn rc=$?"
python - "$O/r04_bench_train_rcnn.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
    print(round(d["ms_per_step"], 3), "ms/step", d["stages_ms"], d["losses"])
except Exception as e:
    print("no line:", e)
PY
