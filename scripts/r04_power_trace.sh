#!/bin/bash
# Round 4: (1) the sustained bf16 MFMA rate with clock / power samples beside it ("evidence the power ceiling"): bursts of > 1 s of
# back-to-back MFMAs on constant and on random operands, rocm-smi polled every 0.1 s in the background; (2) rocprofv3 kernel statistics of the ISOLATED
# bf16 conv launches (the torch-free per-layer harness + the conv1 pair harness); (3) the contract line once more (roi_pool_us_behind_nms_in_one_graph).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r04p; mkdir -p $O; B=scripts/micro/_bin
( for i in $(seq 1 140); do echo "t=$(date +%s.%N) $(/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Socket Graphics Package Power|Average Graphics Package Power|Current Socket' | tr -s ' ' | tr '\n' '|')"; sleep 0.1; done ) > $O/smi_trace.txt 2>&1 &
SMI=$!
sleep 1
{ echo "start $(date +%s.%N)"; timeout 120 $B/mfma_peak_micro 1 20000 1500; echo "end $(date +%s.%N)"; } > $O/mfma_peak_long.txt 2>&1
wait $SMI
{ echo "== scripts/micro/mfma_peak_micro 1 20000 1500 (each line: 1500 back-to-back repetitions of 20000 MFMAs per wave, > 1 s of load) with rocm-smi --showclocks --showpower polled every 0.1 s beside it"; cat $O/mfma_peak_long.txt; echo "== rocm-smi samples"; cat $O/smi_trace.txt; } > $O/r04_mfma_power_trace.txt
head -12 $O/mfma_peak_long.txt; grep -c sclk $O/smi_trace.txt; sed -n 20,24p $O/smi_trace.txt | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_micro" -o r04_micro -- bash -c "cd $R && ./scripts/micro/_bin/conv_bf16_micro && ./scripts/micro/_bin/conv_pair_micro" > "$R/$O/prof_micro.log" 2>&1; echo "rocprof micro rc=$?"
cd "$R"
timeout 600 python bench.py --no-cpu-baseline --no-split-variant --no-bf16-variant > $O/r04_bench_nocpu.json 2> $O/bench.err; python - "$O/r04_bench_nocpu.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]); print(round(d["value"], 1), d["roofline"]["secondary"])
PY
