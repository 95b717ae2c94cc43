# diagnostic driver (GPU box): conv_f16p step stamps under ablation flags
OUT=$1; shift
mkdir -p gpurun_out/$OUT
for V in "$@"; do
  make -C neural_marionette_amd/csrc clean >/dev/null
  if [ $V = BASE ]; then F="-DNM_DIAG"; else F="-DNM_DIAG -DNM_EXP_$V"; fi
  make -C neural_marionette_amd/csrc DIAGFLAGS="$F" 2>&1 | grep -i "error"
  echo "=== $V" >> gpurun_out/$OUT/abl.log
  timeout 120 python tools/diag_f16p_steps.py 2>&1 | grep -v amdgpu.ids | head -3 >> gpurun_out/$OUT/abl.log
done
cat gpurun_out/$OUT/abl.log
