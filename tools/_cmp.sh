# diagnostic driver (GPU box): conv_f16p vs conv_f16s on the four benchmark shapes
mkdir -p gpurun_out/$1; make -C neural_marionette_amd/csrc clean >/dev/null; make -C neural_marionette_amd/csrc DIAGFLAGS=-DNM_DIAG 2>&1 | grep -i error
for F in 1; do echo "=== NM355_F16P=$F" >> gpurun_out/$1/cmp.log; NM355_F16P=$F timeout 120 python tools/diag_f16p_steps.py 2>&1 | grep -v amdgpu >> gpurun_out/$1/cmp.log; done
cat gpurun_out/$1/cmp.log
