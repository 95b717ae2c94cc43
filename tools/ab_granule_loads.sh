#!/bin/bash
# Granule loads of the persistent VRNN chains: 8-byte loads of single granules (shipped, libnm355.so) against the round-5 16-byte loads of granule
# pairs (libnm355_x4.so, `make -C neural_marionette_amd/csrc x4`): config-5 rollout and stand-alone encode, alternating, ONE gpurun call.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
X4=$PWD/neural_marionette_amd/libnm355_x4.so
[ -f "$X4" ] || { echo "build libnm355_x4.so first (make x4)"; exit 1; }
echo "bit-identity tests of the chains with the 8-byte loads: $(python3 -m pytest tests/test_network_gpu.py -q -k 'persistent' 2>&1 | tail -1)"
for i in 1 2 3; do
  echo "dwordx2 (shipped): $(python3 tools/time_rollout.py /tmp/ro_a.pt 2>&1 | grep us/step | tr '\n' ';')"
  echo "dwordx4 (round 5): $(NM355_LIB_PATH=$X4 python3 tools/time_rollout.py /tmp/ro_b.pt /tmp/ro_a.pt 2>&1 | grep -E 'us/step|identical' | tr '\n' ';')"
  echo "dwordx2 (shipped): $(python3 tools/time_encode.py 2>&1 | grep timestep | tr '\n' ';')"
  echo "dwordx4 (round 5): $(NM355_LIB_PATH=$X4 python3 tools/time_encode.py 2>&1 | grep timestep | tr '\n' ';')"
done
