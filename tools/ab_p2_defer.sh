#!/bin/bash
# A/B of conv_f16p2's deferred epilogue (NM355_P2_DEFER): the three conv_f16p2 shapes of the forward, stand-alone
# through nm_op_conv3d (op = conv + GroupNorm statistics), alternating, one call on one device.
LIB=$PWD/neural_marionette_amd/libnm355.so
for i in 1 2; do
  for D in 0 1; do
    echo "NM355_P2_DEFER=$D"
    for SH in "64 64 32 64" "128 128 16 64" "32 64 32 64"; do
      NM355_P2_DEFER=$D python3 tools/time_conv_lib.py $LIB $SH 20 2>&1 | grep -v amdgpu.ids | sed 's/^/  /'
    done
  done
done
