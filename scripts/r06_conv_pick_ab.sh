#!/bin/bash
# Round 6: the fp32 convolution's pick rule, round 5's (FRCNN_CONV_PICK=5) against round 6's, and round 6's with 238 kept on the pooled layers (=w): the whole VGG-16 chain through
# the torch-free harness (fused ReLU + pool where VGG pools), three interleaved repeats, then the bench's conv-chain graph.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for rep in 1 2 3; do for pk in 5 6 w; do echo "== rep $rep pick $pk"; FRCNN_CONV_PICK=$pk CONV_MICRO_BURST=8 timeout 120 ./scripts/micro/_bin/conv_f32_micro | sed 's/GFLOP */GFLOP /'; done; done
for pk in 5 6 5 6; do echo "== bench pick $pk"; FRCNN_CONV_PICK=$pk timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-split-variant --no-bf16-variant --no-feed-variant --no-two-streams-variant 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), 'img/s conv_ms', round(d['roofline']['conv_ms_per_image'],4), 'frac', round(d['roofline']['frac'],4))"; done
