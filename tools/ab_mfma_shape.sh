#!/bin/bash
# MFMA-shape A/B (verdict r5 item 2) on the decoder's 64 -> 32 fused-upsample layer at the bench shape (64 frames, 32^3 -> 64^3), random
# data, alternating runs in ONE call on ONE device (devices differ by up to 12 %: MI355X_MICROARCH.md 'DVFS give-back' item 5).  Arms:
#   A  conv_up2c_kernel<false,0>        v_mfma_f32_32x32x16_f16, two accumulators (hi x hi | correction terms) - the round-5 kernel
#   B  conv_up2c_kernel<false,0,true>   v_mfma_f32_32x32x16_f16, ONE accumulator (hi operand of the main product pre-scaled by 2^11)
#   C  conv_up2c_x16_kernel<false>      v_mfma_f32_16x16x32_f16, one accumulator                                  - shipped
# usage (through gpurun): bash tools/ab_mfma_shape.sh [reps]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R=${1:-9}
export TMPDIR=/tmp
arm() { case $1 in A) echo "NM355_UP2C_X16=0 NM355_UP2C_DIAG=0";; B) echo "NM355_UP2C_X16=0 NM355_UP2C_DIAG=64";; C) echo "NM355_UP2C_X16=1 NM355_UP2C_DIAG=0";; esac; }
echo "parity (tests/test_ops_gpu.py::test_conv3d_fused_upsample_composite, 12 cases: interior / faces / edges / corners vs ATen at 2e-5, run-to-run identity):"
for A in A B C; do echo "  arm $A: $(env $(arm $A) python3 -m pytest tests/test_ops_gpu.py -q -k fused_upsample_composite 2>&1 | tail -1)"; done
echo "whole op (compose + pack + conv + shell + gn_finalize), torch events, median of $R:"
for i in 1 2 3; do for A in A B C; do echo "  arm $A $(env $(arm $A) python3 tools/time_up2c.py $R 2>&1 | grep -o 'median.*ms  min [0-9.]* ms')"; done; done
echo "main kernel alone, rocprofv3 --kernel-trace --stats (3 + $R launches): calls, total ns, average ns, min ns, max ns"
for A in A B C; do
  rm -rf /tmp/abx$A
  env $(arm $A) rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abx$A -- python3 tools/time_up2c.py $R > /dev/null 2>&1
  grep -h "conv_up2c_[kx]" $(find /tmp/abx$A -name "*kernel_stats.csv" | head -1) | sed 's/(anonymous namespace):://g; s/void //' | awk -F'","' -v a=$A '{gsub(/"/,""); split($0,f,","); print "  arm " a " " $0}' | cut -c1-160
done
echo "matrix-pipe busy and clock of the main kernel (own pass: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE; clock = GRBM_GUI_ACTIVE / 8 / duration):"
for A in A B C; do
  rm -rf /tmp/abp$A
  env $(arm $A) rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/abp$A -- python3 tools/time_up2c.py $R > /dev/null 2>&1
  python3 tools/pmc_mfma.py $(find /tmp/abp$A -name "*.db" | head -1) /tmp/abp$A.json 2>&1 | grep -E "conv_up2c_(kernel|x16)" | sed "s/^/  arm $A /"
done
