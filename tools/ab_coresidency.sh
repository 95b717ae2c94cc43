#!/bin/bash
# Co-residency A/B (verdict r4 item 2): do the training step's HBM-bound passes (GroupNorm backward, adjoint, elementwise) run UNDER the
# matrix-core kernels when those leave CUs free?  The persistent producer / consumer convs of the main queue are capped at
# NM355_CONV_WGS workgroups (256 = every CU) and the weight gradients of the third queue at NM355_WGRAD_WGS (default 224), detector-mode
# training step at the bench shape, ms per step, every row inside this one call.  usage (through gpurun): bash tools/ab_coresidency.sh
cd "${GRAFT_REPO_ROOT:-.}"
L=neural_marionette_amd/libnm355.so
for MODE in bf16 split16; do
  for CW in 0 240 224 192 160; do
    for WW in 0 256; do
      echo -n "mode=$MODE conv_wgs=${CW/#0/256(all)} wgrad_wgs=${WW/#0/224(default)}: "
      NM355_CONV_WGS=$CW NM355_WGRAD_WGS=$WW python tools/ab_lib.py $L $MODE 8 2>/dev/null | tail -1
    done
  done
done
