#!/bin/bash
# Round 2, GPU call D: the whole GPU suite, the contract line, per-kernel stats, proposal pipeline split, training / bf16 lines.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r02d}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -s --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3; grep -E "^E " $O/pytest_gpu.log | head -5 | cut -c1-300
echo "== bench f32"; timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
python - <<PY
import json
b = json.load(open("$O/bench.json"))
print(b["value"], b["ms_per_step"], b["roofline"]["frac"], b["nms_roi"]["proposals_nms_us"], b["nms_roi"]["roi_pool_us"], b["nms_roi"]["roi_pool_frac_of_hbm_peak"], b["parity"]["ok"], b["cpu_baseline"]["value"])
PY
echo "== proposals"; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prop" -o prop -- python "$R/scripts/prop_bench.py" > "$R/$O/prop.log" 2>&1; echo "rc=$?"; cd "$R"; grep -v "amdgpu.ids\|rocprofv3\|output_stream\|HSA version" $O/prop.log | tail -6
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/prop/prop_kernel_trace.csv")))
seq = []
for r in rows:
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:24]
    if any(k in n for k in ("nms", "sort", "rank", "decode")):
        seq.append((int(r["Start_Timestamp"]), n, r["Grid_Size_X"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
seq.sort()
dec = [i for i, s in enumerate(seq) if s[1].startswith("proposal_decode")]
for mid in (dec[200], dec[-5]):
    t0 = seq[mid][0]
    for s in seq[mid:mid + 7]:
        print("%8.1f us  %-26s grid %-8s %.1f us" % ((s[0] - t0) / 1e3, s[1], s[2], s[3]))
    print()
PY
echo "== rocprof stats of the bench"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o $TAG -- python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$R/$O/prof.log" 2>&1; echo "rocprof rc=$?"; cd "$R"
echo "== bench train"; timeout 600 python bench.py --mode train --steps 10 --warmup 3 > $O/bench_train.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-200 $O/bench_train.json; python -c "
import json; b=json.load(open('$O/bench_train.json')); print(b['value'], b['ms_per_step'], b['ms_per_step_without_proposal_layer'], b['stages_ms'])"
echo "== bench bf16"; timeout 600 python bench.py --dtype bf16 --steps 50 --warmup 5 > $O/bench_bf16.json 2>> $O/bench.err; echo "rc=$?"; python -c "
import json; b=json.load(open('$O/bench_bf16.json')); print(b['value'], b['ms_per_step'], b['roofline']['frac'], b['nms_roi'], b['parity']['ok'])"
