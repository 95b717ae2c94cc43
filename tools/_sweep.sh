# diagnostic driver (GPU box): total time of the four conv_f16s shapes over NM355_STAGGER values
mkdir -p gpurun_out/$1; make -C neural_marionette_amd/csrc clean >/dev/null; make -C neural_marionette_amd/csrc DIAGFLAGS=-DNM_DIAG 2>&1 | grep -i error
for D in 0 3000 6000 9000 12000 16000 24000; do
  echo "=== stagger $D" >> gpurun_out/$1/sweep.log
  NM355_STAGGER=$D timeout 120 python tools/diag_conv_phases.py 4 2>&1 | grep "us total" >> gpurun_out/$1/sweep.log
done
cat gpurun_out/$1/sweep.log
