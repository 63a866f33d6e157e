#!/bin/bash
# Round 6, third sweep: LDS-DMA forms of tile shapes the shipped rule never had on that staging (research build: MICRO_CONV_F32_FORMS=1 scripts/micro/build_micro.sh), forced
# stream-K and whole tiles, against the shipped picks (-1), two interleaved repeats.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for rep in 1 2; do
for L in conv1_2 conv2_1 conv2_2 conv3_2 conv4_2 conv5_1; do
  for c in -1 240 241 242 243 244 245 247 248 40 41 45 47; do
    printf "$L rep $rep cfg %4s: " $c; CONV_MICRO_CFG=$c CONV_MICRO_BURST=8 timeout 60 ./scripts/micro/_bin/conv_f32_micro_forms $L 2>&1 | grep "^$L" | sed 's/.*GFLOP *//'
  done
done
done
