#!/bin/bash
# Round 5: the RPN training step after the bias-gradient (four loads in flight per thread) and slab-reduction (eight slabs in flight) changes; three runs each line.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r05p; mkdir -p $O
for i in 1 2 3; do timeout 300 python bench.py --mode train --steps 60 --warmup 3 2>>$O/err.log | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('train', d['ms_per_step'], d.get('ms_per_step_without_proposal_layer'))"; done
timeout 300 python bench.py --mode train-rcnn --steps 30 --warmup 3 2>>$O/err.log | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('train-rcnn', d['ms_per_step'])"
timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 -k "train_step_small or train_step_vgg16 or bias or sgd or conv_backward" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_subset.log
