#!/bin/bash
# Round 5: does the number of hardware queues the HIP runtime maps streams onto (GPU_MAX_HW_QUEUES, default 4) change the multi-stream training steps / the fed inference line?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r05u; mkdir -p $O
for q in default 2 8 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for m in train train-rcnn; do
    timeout 300 python bench.py --mode $m --steps 40 --warmup 3 2>>$O/err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('queues $q  $m  %.3f ms/step' % d['ms_per_step'])"
  done
done
