#!/bin/bash
# Round 4, the evidence run on the final tree: everything scripts/r04_gpu_evidence.sh collects (smoke, the whole GPU suite without -x, every bench line,
# RCCL at world size 1 / gloo with two ranks, rocprofv3 kernel statistics, HBM-traffic and MFMA / RoI PMC passes, the micro harnesses) plus what is new in
# round 4: the stage-2 training step (device-drawn and NumPy-stream dropout masks), the conv1 pair launch against its two-launch chain, the resident forms.
# Outputs under gpurun_out/r04z/; scripts/collect_profiles_r04.py r04z copies what is judged into profiles/r04_*.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
STAGES="${STAGES:-tests bench prof pmc dist micro}" TAG=r04z bash scripts/r04_gpu_evidence.sh
O=gpurun_out/r04z; B=scripts/micro/_bin
for rng in device numpy; do
  timeout 600 python bench.py --mode train-rcnn --dropout-rng $rng --steps 20 --warmup 3 > $O/r04_bench_train_rcnn_$rng.json 2>> $O/bench.err; echo "train-rcnn ($rng masks) rc=$?"; cut -c1-160 $O/r04_bench_train_rcnn_$rng.json | tail -1
done
{ for pr in 2 1 0; do echo "== form 2, FRCNN_BF16_PAIR_PRIO=$pr (0 none, 1 consumers first, 2 producers first)"; FRCNN_BF16_PAIR_PRIO=$pr timeout 60 $B/conv_pair_micro; done
  for rw in 6 4; do echo "== form 1 (one wave per SIMD, weights in registers), RW $rw"; FRCNN_BF16_PAIR_FORM=1 FRCNN_BF16_PAIR_RW=$rw timeout 60 $B/conv_pair_micro; done; } > $O/r04_conv_pair_micro.txt 2>&1; grep "^pair" $O/r04_conv_pair_micro.txt
{ echo "=== chain: default picks vs conv_dma_bf16_kernel's (old)"; timeout 120 $B/conv_bf16_micro --check --modes "def old"; } > $O/r04_conv_bf16_micro.txt 2>&1; tail -3 $O/r04_conv_bf16_micro.txt
{ timeout 60 $B/mfma_peak_micro 1 20000 10; } > $O/r04_mfma_peak_micro.txt 2>&1
for f in 1 0; do FRCNN_BF16_CONV1_PAIR=$f timeout 600 python bench.py --dtype bf16 --steps 100 --warmup 5 --no-cpu-baseline > $O/r04_bench_bf16_pair$f.json 2>> $O/bench.err; echo "bench bf16, conv1 pair launch = $f: rc=$?"; cut -c1-140 $O/r04_bench_bf16_pair$f.json | tail -1; done
grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -20
