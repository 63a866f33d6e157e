#!/bin/bash
# The first GPU call of round 4: everything the strip picks of late round 3 still owe --
#   1. the WHOLE GPU suite without -x (38 tests never executed; the K-split report line names the offending words),
#      bench.py (contract line) and --dtype bf16, rocprofv3 kernel statistics of both, HBM traffic + MFMA counters of the shipped bf16 chain,
#   2. the torch-free chain old vs new with the word-by-word comparison, and form D's two epilogues side by side,
#   3. the sustained MFMA rate once more, with bursts of 6 ms and of over a second.
# ~8-10 GPU-minutes.  Outputs under gpurun_out/r04a/ (scripts/collect_profiles_r04.py copies what is judged into profiles/r04_*).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
STAGES="tests bench prof pmc" TAG=r04a bash scripts/r04_gpu_evidence.sh
O=gpurun_out/r04a; mkdir -p $O; B=scripts/micro/_bin
{ echo "=== chain: default picks vs conv_dma_bf16_kernel's (old)"; timeout 120 $B/conv_bf16_micro --check --modes "def old";
  echo "=== form D: LDS-transpose epilogue (909) vs direct stores (910)"; timeout 60 $B/conv_bf16_micro --check --modes "909 910" conv2_2 conv3_1 conv3_2 conv3_3 conv4_1 conv4_2 conv4_3; } > $O/r04_conv_bf16_micro.txt 2>&1
{ timeout 60 $B/mfma_peak_micro 1 20000 10; timeout 120 $B/mfma_peak_micro 1 20000 2000; } > $O/r04_mfma_peak_micro.txt 2>&1
tail -14 $O/r04_conv_bf16_micro.txt; tail -12 $O/r04_mfma_peak_micro.txt
grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -20
