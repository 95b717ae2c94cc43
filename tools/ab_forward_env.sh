#!/bin/bash
# Alternating A/B of the headline forward step (bench.py --no-extras --no-cpu-baseline, 64^3 T=16 B=4) under NM355_* switches, ONE gpurun call.
# usage: bash tools/ab_forward_env.sh "<env A>" "<env B>" [rounds]      e.g. "NM355_UP2C_ALL=0" "NM355_UP2C_ALL=1"
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
A="$1"; B="$2"; R=${3:-3}
for i in $(seq $R); do
  for E in "$A" "$B"; do
    echo "[$E] $(env $E python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f ms per step, %.0f voxel-frames/s" % (d["ms_per_step"], d["value"]))')"
  done
done
