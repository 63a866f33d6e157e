#!/bin/bash
# Round 2, GPU call A: the full GPU suite (incl. the 600x1000 parity tests), the contract bench line with its parity block,
# RoI pooling A/B (cell-major vs plane kernel) + LDS counters, per-kernel rocprof stats, bf16 / training lines, 2-rank smoke.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=r02a
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -s --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3; grep "^PARITY" $O/pytest_gpu.log | cut -c1-400
echo "== bench f32"; timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-1500 $O/bench.json; tail -3 $O/bench.err
echo "== roi bench"; timeout 300 python scripts/roi_bench.py > $O/roi_bench.log 2>&1; echo "rc=$?"; cat $O/roi_bench.log | tail -8
echo "== rocprof stats"; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o $TAG -- python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$R/$O/prof.log" 2>&1; echo "rocprof rc=$?"
echo "== roi pmc"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d "$R/$O/rpmc1" -o p1 -- python "$R/scripts/roi_bench.py" > "$R/$O/rpmc1.log" 2>&1; echo "rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d "$R/$O/rpmc2" -o p2 -- python "$R/scripts/roi_bench.py" > "$R/$O/rpmc2.log" 2>&1; echo "rc=$?"
cd "$R"
python scripts/pmc_summary.py $O/rpmc1 $O/rpmc2 roi_pool_cells_kernel 2>/dev/null | awk 'NR<3 || /==/ {print}' | cut -c1-600
python scripts/pmc_summary.py $O/rpmc1 $O/rpmc2 roi_pool_planes_kernel 2>/dev/null | awk 'NR<3 || /==/ {print}' | cut -c1-600
echo "== bench bf16"; timeout 600 python bench.py --dtype bf16 --steps 50 --warmup 5 > $O/bench_bf16.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-700 $O/bench_bf16.json
echo "== bench train"; timeout 600 python bench.py --mode train --steps 10 --warmup 3 > $O/bench_train.json 2>> $O/bench.err; echo "rc=$?"; cat $O/bench_train.json
echo "== 2-rank gloo smoke (self-launch)"; timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_2rank_infer.json 2> $O/bench_2rank.err; echo "rc=$?"; cut -c1-500 $O/bench_2rank_infer.json
timeout 600 python bench.py --gpus 2 --mode train --steps 4 --warmup 1 > $O/bench_2rank_train.json 2>> $O/bench_2rank.err; echo "rc=$?"; cut -c1-500 $O/bench_2rank_train.json; tail -3 $O/bench_2rank.err
head -30 $O/prof/${TAG}_kernel_stats.csv | cut -c1-160
