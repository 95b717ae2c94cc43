#!/bin/bash
# MFMA-shape A/B (verdict r5 item 2): conv_up2c_kernel (v_mfma_f32_32x32x16_f16) against conv_up2c_x16_kernel (v_mfma_f32_16x16x32_f16)
# on the decoder's 64 -> 32 fused-upsample layer at the bench shape (64 frames, 32^3 -> 64^3), random data, alternating runs in ONE call
# on ONE device (devices differ by up to 12 %: MI355X_MICROARCH.md 'DVFS give-back' item 5).  usage (through gpurun): bash tools/ab_mfma_shape.sh [reps]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R=${1:-9}
echo "parity of both kernels (tests/test_ops_gpu.py::test_conv3d_fused_upsample_composite):"
for X in 0 1; do
  echo "NM355_UP2C_X16=$X: $(NM355_UP2C_X16=$X python3 -m pytest tests/test_ops_gpu.py -q -k fused_upsample_composite 2>&1 | tail -1)"
done
for i in 1 2 3; do
  for X in 0 1; do
    echo "X16=$X $(NM355_UP2C_X16=$X python3 tools/time_up2c.py $R 2>&1 | grep median)"
  done
done
echo "main kernel only (rocprofv3 --kernel-trace --stats, 3 + $R launches each):"
export TMPDIR=/tmp
for X in 0 1; do
  rm -rf /tmp/abx$X
  NM355_UP2C_X16=$X rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abx$X -- python3 tools/time_up2c.py $R > /dev/null 2>&1
  grep -h "conv_up2c" $(find /tmp/abx$X -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4,6-7 | sed "s/^/X16=$X /"
done
