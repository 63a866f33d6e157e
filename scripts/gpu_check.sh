#!/bin/bash
# One gpurun call: smoke -> GPU parity tests -> bench -> rocprofv3 kernel stats.  Outputs under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== conv sweep"; timeout 900 python scripts/conv_sweep.py --out gpurun_out/conv_sweep.json 2>&1 | grep -v amdgpu.ids
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
echo "== rocprof"; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o r01 -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$R/gpurun_out/prof.log" 2>&1; echo "rocprof rc=$?"
cd "$R"; find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -40
