#!/bin/bash
# Round 2, GPU call B: re-check the two failed tests, RoI A/B after the VALU diet, proposal pipeline split, bf16 tile-shape sweep.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=r02b
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest subset"; timeout 900 python -m pytest tests -m gpu -q -s --timeout 600 -k "nms or proposal or roi or train_step_600 or conv_bf16 or empty" > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^E |wgrad vs" $O/pytest.log | head -10 | cut -c1-400
echo "== roi bench"; timeout 300 python scripts/roi_bench.py > $O/roi_bench.log 2>&1; tail -7 $O/roi_bench.log
FRCNN_ROI_SPLIT_MUL=2 timeout 300 python scripts/roi_bench.py 2>&1 | grep cells
echo "== proposals"; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prop" -o prop -- python "$R/scripts/prop_bench.py" > "$R/$O/prop.log" 2>&1; echo "rc=$?"; cd "$R"; grep -v amdgpu.ids $O/prop.log | tail -8
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r02b/prop/prop_kernel_stats.csv")):
    n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:50]
    if any(k in n for k in ("nms", "sort", "rank", "decode")):
        print("%-52s calls %5s avg %8.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
echo "== bf16 conv sweep"; FRCNN_BF16_DMAS="141 231 321 223 233 323 224 324 124 133" timeout 900 python scripts/conv_bf16_sweep.py > $O/bf16_sweep.log 2>&1; echo "rc=$?"
FRCNN_BF16_SPLIT=2 FRCNN_BF16_DMAS="231 223 224" timeout 600 python scripts/conv_bf16_sweep.py > $O/bf16_sweep_split2.log 2>&1
FRCNN_BF16_SPLIT=4 FRCNN_BF16_DMAS="231 223 224" timeout 600 python scripts/conv_bf16_sweep.py > $O/bf16_sweep_split4.log 2>&1
for f in bf16_sweep bf16_sweep_split2 bf16_sweep_split4; do echo "-- $f"; grep -v amdgpu.ids $O/$f.log | tail -12 | cut -c1-420; done
