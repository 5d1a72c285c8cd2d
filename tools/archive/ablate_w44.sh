#!/bin/bash
# conv_wino44 timing ablations (LOG R5.1).  Build one library per mask in the build container (only the two conv_wino44 translation units):
#   for m in 0 1 2 3 4 8 11 16 32 64 127; do bash tools/build_variant.sh abl$m "-DFV_X_W44_ABL=$m" "conv_wino44_k7 conv_wino44_k11" & done; wait
# then on the GPU box time each in its own process, two interleaved rounds:   bash tools/ablate_w44.sh "0 1 2 3 4 8 11 16 32 64 127"
MASKS=${1:-"0 1 2 3 4 8 11 16 32 64 127"}
for r in 1 2; do
  for m in $MASKS; do FV_LIB_PATH=$PWD/vocoder_amd/csrc/libfishvoc_x_abl$m.so python tools/probe_w44_ablation.py "mask $m" 2>/dev/null; done
done
