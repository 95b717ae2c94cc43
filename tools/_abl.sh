# diagnostic driver (GPU box): phase stamps of conv_f16s under ablation flags; usage: bash tools/_abl.sh OUTDIR NCASES V1 V2 ...
OUT=$1; NC=$2; shift 2
mkdir -p gpurun_out/$OUT
for V in "$@"; do
  make -C neural_marionette_amd/csrc clean >/dev/null
  if [ $V = BASE ]; then F="-DNM_DIAG"; else F="-DNM_DIAG -DNM_EXP_$V"; fi
  make -C neural_marionette_amd/csrc DIAGFLAGS="$F" 2>&1 | grep -i "error"
  echo "=== $V" >> gpurun_out/$OUT/abl.log
  timeout 120 python tools/diag_conv_phases.py $NC 2>&1 | grep -v amdgpu.ids >> gpurun_out/$OUT/abl.log
done
cat gpurun_out/$OUT/abl.log
