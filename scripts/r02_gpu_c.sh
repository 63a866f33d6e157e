#!/bin/bash
# Round 2, GPU call C: staged NMS / column scan, 16-wave staged-output RoI kernel, full-size training parity with the float64 arbiter.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=r02c
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest subset"; timeout 900 python -m pytest tests -m gpu -q -s --timeout 600 -k "nms or proposal or roi or train_step_600 or empty or end_to_end or 600x1000_fp32" > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^E |full-size weight" $O/pytest.log | head -10 | cut -c1-700
echo "== roi bench"; timeout 300 python scripts/roi_bench.py > $O/roi_bench.log 2>&1; tail -7 $O/roi_bench.log
echo "== proposals"; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prop" -o prop -- python "$R/scripts/prop_bench.py" > "$R/$O/prop.log" 2>&1; echo "rc=$?"; cd "$R"; grep -v "amdgpu.ids\|rocprofv3\|output_stream\|HSA version" $O/prop.log | tail -8
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r02c/prop/prop_kernel_stats.csv")):
    n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:50]
    if any(k in n for k in ("nms", "sort", "rank", "decode")):
        print("%-52s calls %5s avg %8.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
echo "== roi pmc"; cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d "$R/$O/rpmc1" -o p1 -- python "$R/scripts/roi_bench.py" > "$R/$O/rpmc1.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d "$R/$O/rpmc2" -o p2 -- python "$R/scripts/roi_bench.py" > "$R/$O/rpmc2.log" 2>&1
cd "$R"
python scripts/pmc_summary.py $O/rpmc1 $O/rpmc2 "roi_pool_cells_kernel<38, false>" 2>/dev/null | awk 'NR<3 || /==/ {print}' | cut -c1-600
echo "== bench f32"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
b = json.load(open("gpurun_out/r02c/bench.json"))
print(b["value"], b["ms_per_step"], b["nms_roi"])
PY
