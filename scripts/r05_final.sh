#!/bin/bash
# Round 5, the evidence run on the final tree: everything scripts/gpu_evidence.sh collects (smoke, the whole GPU suite without -x, every bench line -- the contract
# line now carries `with_feed` --, RCCL at world size 1 / gloo with two ranks, rocprofv3 kernel statistics, HBM-traffic and MFMA / RoI PMC passes, the micro
# harnesses) plus: the stage-2 training step (device-drawn and NumPy-stream dropout masks), the conv1 pair launch against its two-launch chain, the bf16 line with /
# without it, the RoI backward kernel's PMC pass, the sustained MFMA rates, package power beside the bench lines.
# Outputs under gpurun_out/$TAG/ (default r05z; gpurun does not overwrite files of an earlier call: a re-run takes a fresh TAG); scripts/collect_profiles.py <TAG> r05 copies what is judged into profiles/r05_*.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
START=$(date +%s); STAGES="${STAGES:-tests bench prof pmc dist micro}" TAG=${TAG:=r05z} P=r05 bash scripts/gpu_evidence.sh
O=gpurun_out/$TAG; B=scripts/micro/_bin
for rng in device numpy; do
  timeout 600 python bench.py --mode train-rcnn --dropout-rng $rng --steps 20 --warmup 3 > $O/r05_bench_train_rcnn_$rng.json 2>> $O/bench.err; echo "train-rcnn ($rng masks) rc=$?"; cut -c1-160 $O/r05_bench_train_rcnn_$rng.json | tail -1
done
{ for pr in 2 1 0; do echo "== FRCNN_BF16_PAIR_PRIO=$pr (0 none, 1 consumers first, 2 producers first)"; FRCNN_BF16_PAIR_PRIO=$pr timeout 60 $B/conv_pair_micro; done; } > $O/r05_conv_pair_micro.txt 2>&1; grep "^pair" $O/r05_conv_pair_micro.txt
{ echo "=== chain: default picks vs conv_dma_bf16_kernel's (old)"; timeout 120 $B/conv_bf16_micro --check --modes "def old"; } > $O/r05_conv_bf16_micro.txt 2>&1; tail -3 $O/r05_conv_bf16_micro.txt
{ timeout 60 $B/mfma_peak_micro 1 20000 10; } > $O/r05_mfma_peak_micro.txt 2>&1
# the stage-2 step's other forms: split-product convolutions; the zero-padded head backward over all RoI rows and the count-first order (the A/B of the round's changes)
timeout 600 python bench.py --mode train-rcnn --dtype f32s --dropout-rng device --steps 20 --warmup 3 > $O/r05_bench_train_rcnn_f32s.json 2>> $O/bench.err; echo "train-rcnn f32s rc=$?"; cut -c1-160 $O/r05_bench_train_rcnn_f32s.json | tail -1
FRCNN_RCNN_BWD_ROWS=all FRCNN_RCNN_HEAD_FWD=late timeout 600 python bench.py --mode train-rcnn --dropout-rng device --steps 20 --warmup 3 > $O/r05_bench_train_rcnn_allrows_late.json 2>> $O/bench.err; echo "train-rcnn all rows, count first rc=$?"; cut -c1-160 $O/r05_bench_train_rcnn_allrows_late.json | tail -1
# ... and its kernels
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_rcnn" -o r05_train_rcnn -- python "$R/bench.py" --mode train-rcnn --dropout-rng device --steps 10 --warmup 2 > "$R/$O/prof_rcnn.log" 2>&1; echo "rocprof train-rcnn rc=$?" ); find $O/prof_rcnn -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
for f in 1 0; do FRCNN_BF16_CONV1_PAIR=$f timeout 600 python bench.py --dtype bf16 --steps 100 --warmup 5 --no-cpu-baseline > $O/r05_bench_bf16_pair$f.json 2>> $O/bench.err; echo "bench bf16, conv1 pair launch = $f: rc=$?"; cut -c1-140 $O/r05_bench_bf16_pair$f.json | tail -1; done
# two images in flight per GPU (graph.ForwardsInFlight's mechanism, torch-level probe): serial vs two / three instances, outputs compared
{ for a in "bf16 2" "bf16 3" "f32 2" "f32s 2"; do timeout 300 python scripts/two_streams_probe.py $a; done; } 2>&1 | grep -v amdgpu.ids > $O/r05_two_streams_probe.txt; cat $O/r05_two_streams_probe.txt
# ... and what the kernels look like under it: rocprofv3 kernel statistics of the two-instance run (durations stretch where two images share the chip)
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_two" -o r05_two -- python "$R/scripts/two_streams_probe.py" bf16 2 > "$R/$O/prof_two.log" 2>&1; echo "rocprof two-in-flight rc=$?" ); find $O/prof_two -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
# the backward RoI kernel's counters (torch-free harness; the bwd launches of roi_micro)
scripts/micro/roi_pmc.sh 'roi_pool_bwd_runs_kernel<2' DEFAULT=1 > $O/r05_roi_bwd_pmc.txt 2>&1; tail -12 $O/r05_roi_bwd_pmc.txt
# package power / clocks beside the bench lines (rocm-smi polled every 0.25 s)
O=$O/power bash scripts/bench_power.sh > $O/r05_bench_power.txt 2>&1; tail -12 $O/r05_bench_power.txt
grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -20
echo "whole evidence run: $(( $(date +%s) - START )) s"
