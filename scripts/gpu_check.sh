#!/bin/bash
# One gpurun call: smoke -> GPU parity tests -> bench (inference contract line + training step) -> rocprofv3 kernel stats.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r01}
cd "$R"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log
echo "== hbm traffic (PMC passes of the bench command; the bench line below carries the figure)"; bash scripts/gpu_traffic.sh > gpurun_out/traffic.log 2>&1; echo "traffic rc=$?"
cp gpurun_out/${TAG}_hbm_traffic_pmc.json profiles/${TAG}_hbm_traffic_pmc.json 2>/dev/null
echo "== bench"; timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/bench.err
echo "== bench train"; timeout 600 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/${TAG}_bench_train.json 2>> gpurun_out/bench.err; echo "rc=$?"; cat gpurun_out/${TAG}_bench_train.json
echo "== bench bf16"; timeout 600 python bench.py --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_bf16.json 2>> gpurun_out/bench.err; echo "rc=$?"; cut -c1-300 gpurun_out/${TAG}_bench_bf16.json
echo "== rocprof"; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o ${TAG} -- python "$R/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > "$R/gpurun_out/prof.log" 2>&1; echo "rocprof rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_train" -o ${TAG}_train -- python "$R/bench.py" --mode train --steps 5 --warmup 2 > "$R/gpurun_out/prof_train.log" 2>&1; echo "rocprof train rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bf16" -o ${TAG}_bf16 -- python "$R/bench.py" --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > "$R/gpurun_out/prof_bf16.log" 2>&1; echo "rocprof bf16 rc=$?"
cd "$R"; head -12 gpurun_out/prof/${TAG}_kernel_stats.csv | cut -c1-200
