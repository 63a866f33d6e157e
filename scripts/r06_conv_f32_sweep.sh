#!/bin/bash
# Round 6: re-sweep of the fp32 convolution's decompositions (LDS-DMA forms x whole-tile / stream-K / forced stream-K) per VGG-16 layer shape through the torch-free
# harness (plain ReLU epilogue; cfg -1 = the library's pick).  The picks of pick_conv_config date from round 1's sweep, several kernel rewrites ago.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for L in conv1_2 conv2_1 conv2_2 conv3_1 conv3_2 conv4_1 conv4_2 conv5_1; do
  echo "== $L"
  for c in -1 30 34 35 36 37 38 39 46 130 134 135 136 137 138 139 146 230 234 236 238 239; do
    printf "cfg %4s: " $c; CONV_MICRO_CFG=$c CONV_MICRO_BURST=4 timeout 60 ./scripts/micro/_bin/conv_f32_micro $L 2>&1 | grep "^$L" | awk '{print $(NF-3), $(NF-2), $(NF-1), $NF}'
  done
done
