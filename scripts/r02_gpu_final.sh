#!/bin/bash
# Round 2, final evidence run: GPU suite, contract line (+ parity, CPU baseline), bf16 and training lines, 2-rank smoke of the
# self-launching N>1 path, rocprofv3 kernel stats of the bench commands, PMC passes (HBM traffic; RoI LDS / VALU counters).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r02z}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -s --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -2
grep -E "^PARITY|full-size weight" $O/pytest_gpu.log > $O/parity_reports.txt
echo "== hbm traffic PMC"; cd /tmp && export TMPDIR=/tmp
for dt in f32 bf16 f32s; do for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/$O/traffic_${dt}_$c" -o t -- python "$R/bench.py" --dtype $dt --steps 3 --warmup 2 --no-cpu-baseline --no-stage-events --no-split-variant > "$R/$O/traffic_${dt}_$c.log" 2>&1; echo "$dt $c rc=$?"
done; done
cd "$R"; python scripts/r02_gpu_traffic.py $O f32 > $O/traffic_f32_summary.txt 2>&1; python scripts/r02_gpu_traffic.py $O bf16 > $O/traffic_bf16_summary.txt 2>&1; python scripts/r02_gpu_traffic.py $O f32s > $O/traffic_f32s_summary.txt 2>&1; tail -12 $O/traffic_f32_summary.txt
mkdir -p profiles; cp $O/r02_hbm_traffic_pmc.json $O/r02_hbm_traffic_pmc_bf16.json $O/r02_hbm_traffic_pmc_f32s.json profiles/ 2>/dev/null
echo "== bench f32"; timeout 600 python bench.py --steps 20 --warmup 5 > $O/r02_bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/r02_bench.json
echo "== bench f32 default steps"; timeout 600 python bench.py --no-cpu-baseline > $O/r02_bench_default_steps.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-200 $O/r02_bench_default_steps.json
echo "== bench bf16"; timeout 600 python bench.py --dtype bf16 --steps 50 --warmup 5 > $O/r02_bench_bf16.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-200 $O/r02_bench_bf16.json
echo "== bench f32s"; timeout 600 python bench.py --dtype f32s --steps 100 --warmup 5 > $O/r02_bench_f32s.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-200 $O/r02_bench_f32s.json
echo "== bench train"; timeout 600 python bench.py --mode train --steps 20 --warmup 3 > $O/r02_bench_train.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-200 $O/r02_bench_train.json
echo "== bench train f32s"; timeout 600 python bench.py --mode train --dtype f32s --steps 20 --warmup 3 > $O/r02_bench_train_f32s.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-200 $O/r02_bench_train_f32s.json
echo "== 2-rank gloo smoke (self-launch)"; timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 > $O/r02_bench_2rank_gloo_infer.json 2> $O/bench_2rank.err; echo "rc=$?"; cut -c1-200 $O/r02_bench_2rank_gloo_infer.json
timeout 600 python bench.py --gpus 2 --mode train --steps 6 --warmup 1 > $O/r02_bench_2rank_gloo_train.json 2>> $O/bench_2rank.err; echo "rc=$?"; cut -c1-200 $O/r02_bench_2rank_gloo_train.json
echo "== rocprof kernel stats"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_f32s" -o r02_f32s -- python "$R/bench.py" --dtype f32s --steps 20 --warmup 5 --no-cpu-baseline > "$R/$O/prof_f32s.log" 2>&1; echo "rocprof f32s rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o r02 -- python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-split-variant > "$R/$O/prof.log" 2>&1; echo "rocprof rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_bf16" -o r02_bf16 -- python "$R/bench.py" --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > "$R/$O/prof_bf16.log" 2>&1; echo "rocprof bf16 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_train" -o r02_train -- python "$R/bench.py" --mode train --steps 5 --warmup 2 > "$R/$O/prof_train.log" 2>&1; echo "rocprof train rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prop" -o prop -- python "$R/scripts/prop_bench.py" > "$R/$O/prop.log" 2>&1; echo "prop rc=$?"
echo "== roi pmc"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d "$R/$O/rpmc1" -o p1 -- python "$R/scripts/roi_bench.py" > "$R/$O/rpmc1.log" 2>&1; echo "rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d "$R/$O/rpmc2" -o p2 -- python "$R/scripts/roi_bench.py" > "$R/$O/rpmc2.log" 2>&1; echo "rc=$?"
cd "$R"; grep -v amdgpu.ids $O/rpmc1.log | tail -7
head -20 $O/prof/r02_kernel_stats.csv | cut -c1-150
