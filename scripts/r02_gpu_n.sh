#!/bin/bash
# Round 2: XCD-aware tile order in the bf16 / split conv kernels: parity + bench lines (bf16 A/B, f32s).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r02n}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest bf16/f32s"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "bf16 or f32s or conv1" --timeout 800 > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2

echo "== bench bf16"; timeout 600 python bench.py --dtype bf16 --steps 100 --warmup 5 --no-cpu-baseline > $O/bench_bf16.json 2>> $O/bench.err; cut -c1-200 $O/bench_bf16.json
echo "== bench f32s"; timeout 600 python bench.py --dtype f32s --steps 100 --warmup 5 --no-cpu-baseline > $O/bench_f32s.json 2>> $O/bench.err; cut -c1-200 $O/bench_f32s.json
python - <<'PY'
import json
for n in ("bench_bf16","bench_f32s"):
    d=json.load(open("gpurun_out/r02n/%s.json"%n)); print(n, round(d["value"],1), round(d["ms_per_step"],4), round(d["roofline"]["conv_ms_per_image"],4), round(d["roofline"]["frac"],4))
PY
