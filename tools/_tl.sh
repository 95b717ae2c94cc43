mkdir -p gpurun_out/$1; make -C neural_marionette_amd/csrc clean >/dev/null; make -C neural_marionette_amd/csrc DIAGFLAGS=-DNM_DIAG 2>&1 | grep -i error
for c in 0 1; do for D in 0 8000; do echo "=== case $c stagger $D" >> gpurun_out/$1/tl.log; NM355_STAGGER=$D timeout 300 python tools/diag_conv_timeline.py $c 2>&1 | grep -v amdgpu.ids >> gpurun_out/$1/tl.log; done; done
cat gpurun_out/$1/tl.log
