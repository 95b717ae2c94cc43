#!/bin/bash
# Round evidence on the GPU box (run through gpurun): bench line, rocprofv3 kernel stats (forward / training / configs 4 and 5), PMC
# passes (HBM traffic, matrix-core utilisation), forward + training timelines.  Everything lands in gpurun_out/evidence/; the summaries
# are copied to profiles/ by hand.  Every summary carries the commit / tree id it was taken from (tools/tree_id.py).
# usage: bash tools/collect_evidence.sh r04
R=${1:?usage: collect_evidence.sh <round, e.g. r04>}
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}" || exit 1
export TMPDIR=/tmp
export NM355_ROUND="$R"
E=gpurun_out/evidence
mkdir -p gpurun_out || exit 1
rm -rf "./$E"; mkdir -p "$E" || exit 1
TREE=$(python3 tools/tree_id.py)
echo "$TREE" > "$E/${R}_tree_id.txt"
CMD="python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $E/fwd -- $CMD > $E/fwd.log 2>&1
cp $(find $E/fwd -name "*kernel_stats.csv" | head -1) $E/${R}_bench_kernel_stats.csv
python3 tools/trace_timeline.py $E/fwd "conv_k5occ_f16_kernel<1" 2 --starved --list > $E/${R}_forward_timeline.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $E/train -- python3 bench.py --workload train --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $E/train.log 2>&1
cp $(find $E/train -name "*kernel_stats.csv" | head -1) $E/${R}_train_kernel_stats.csv
python3 tools/trace_timeline.py $E/train "conv_k5occ_f16_kernel<1" 2 --starved --list > $E/${R}_train_timeline.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $E/trainb -- python3 bench.py --workload train --conv-mode bf16 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $E/trainb.log 2>&1
cp $(find $E/trainb -name "*kernel_stats.csv" | head -1) $E/${R}_train_bf16_kernel_stats.csv
python3 tools/trace_timeline.py $E/trainb "conv_k5occ_f16_kernel<1" 2 --starved --list > $E/${R}_train_bf16_timeline.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $E/c4 -- python3 tools/time_config4.py 96 > $E/c4.log 2>&1
cp $(find $E/c4 -name "*kernel_stats.csv" | head -1) $E/${R}_config4_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $E/c5 -- python3 tools/time_rollout.py /tmp/ro.pt > $E/c5.log 2>&1
cp $(find $E/c5 -name "*kernel_stats.csv" | head -1) $E/${R}_config5_rollout_kernel_stats.csv
# FETCH_SIZE calibration on known byte counts (per access pattern), then the PMC passes, each in its own run (FETCH_SIZE and WRITE_SIZE do not fit one pass)
bash tools/calib/run_fetch_calib.sh > $E/${R}_fetch_calib.txt 2>&1; cp gpurun_out/fetch_calib.json $E/${R}_fetch_calib.json
PCMD="python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $E/tf -- $PCMD > $E/tf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $E/tw -- $PCMD > $E/tw.log 2>&1
python3 tools/pmc_traffic_all.py $E/tf $E/tw $R $E/${R}_pmc_traffic.json "$PCMD" > $E/pmc_traffic_summary.txt 2>&1
for M in split16 bf16; do
  TCMD="python3 bench.py --workload train --conv-mode $M --steps 2 --warmup 1 --no-extras --no-cpu-baseline"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $E/ttf -- $TCMD > $E/ttf.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $E/ttw -- $TCMD > $E/ttw.log 2>&1
  python3 tools/pmc_traffic_all.py $E/ttf $E/ttw $R $E/${R}_pmc_traffic_train_$M.json "$TCMD" > $E/pmc_traffic_train_${M}_summary.txt 2>&1
  rm -rf $E/ttf $E/ttw
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $E/pm -- $PCMD > $E/pm.log 2>&1
python3 tools/pmc_mfma.py $(find $E/pm -name "*.db" | head -1) $E/${R}_pmc_mfma.json > $E/pmc_mfma_summary.txt 2>&1
# matrix-core utilisation of the training step in BASELINE config 3's precision (the one-product kernels)
TBCMD="python3 bench.py --workload train --conv-mode bf16 --steps 2 --warmup 1 --no-extras --no-cpu-baseline"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $E/pmb -- $TBCMD > $E/pmb.log 2>&1
python3 tools/pmc_mfma.py $(find $E/pmb -name "*.db" | head -1) $E/${R}_pmc_mfma_train_bf16.json > $E/pmc_mfma_train_bf16_summary.txt 2>&1
# same-call A/Bs of the round (r06): MFMA shape of conv_up2c, hand-off latency by placement / store flavour, granule loads of the persistent
# chains (needs libnm355_x4.so: make -C neural_marionette_amd/csrc x4), the rollout chain's forms + where a step's time goes, encode,
# forward-step switches, R processes on one GPU
bash tools/ab_mfma_shape.sh 7 > $E/${R}_mfma_shape_ab.txt 2>&1
(hipcc -O3 -w --offload-arch=gfx950 tools/calib/hop_latency.hip -o /tmp/hop_latency && timeout 120 /tmp/hop_latency) > $E/${R}_hop_latency.txt 2>&1
if [ -f neural_marionette_amd/libnm355_x4.so ]; then bash tools/ab_granule_loads.sh 2>&1 | grep -v amdgpu.ids > $E/${R}_granule_loads_ab.txt; fi
# conv_f16p2's deferred epilogue (built, slower, opt-in) against the shipped exposed one; in-kernel stamps of both (libnm355_stamps.so: make stamps)
{ bash tools/ab_p2_defer.sh; if [ -f neural_marionette_amd/libnm355_stamps.so ]; then for D in 0 1; do echo "in-kernel stamps, NM355_P2_DEFER=$D (tools/diag_f16p2_steps.py):"; NM355_P2_DEFER=$D python3 tools/diag_f16p2_steps.py 2>&1 | grep -v amdgpu.ids | head -5; done; fi; } > $E/${R}_p2_defer_ab.txt 2>&1
# the measured upper bound of a Winograd F(2,3) form of conv_f16p2 (needs libnm355_diag.so: make -C neural_marionette_amd/csrc diag)
if [ -f neural_marionette_amd/libnm355_diag.so ]; then python3 tools/diag_winograd_emu.py 2>&1 | grep -v amdgpu.ids > $E/${R}_winograd_emu.txt; fi
{
  echo "config-5 rollout (tools/time_rollout.py: generate, Tcond = 5 posterior + 64 prior steps), alternating, one call:"
  for i in 1 2; do
    echo "one-XCD chain (default at B = 1; forced at B = 3): $(NM355_CHAIN_XCD=2 python3 tools/time_rollout.py /tmp/ro_x.pt 2>&1 | grep us/step | tr '\n' ';')"
    echo "cross-XCD chain, one poller/WG : $(NM355_CHAIN_XCD=0 NM355_CHAIN_WGPOLL=1 python3 tools/time_rollout.py /tmp/ro_w.pt /tmp/ro_x.pt 2>&1 | grep -E 'us/step|identical' | tr '\n' ';')"
    echo "cross-XCD chain, every wave polls: $(NM355_CHAIN_XCD=0 NM355_CHAIN_WGPOLL=0 python3 tools/time_rollout.py /tmp/ro_c.pt /tmp/ro_x.pt 2>&1 | grep -E 'us/step|identical' | tr '\n' ';')"
    echo "three launches per prior step  : $(NM355_VRNN_CHAIN=0 python3 tools/time_rollout.py /tmp/ro_l.pt /tmp/ro_x.pt 2>&1 | grep -E 'us/step|identical' | tr '\n' ';')"
  done
  echo; echo "where a prior step's time goes (tools/diag_chain_stamps.py, s_memrealtime stamps, steps 8..63 of a rollout):"
  for X in 2 0; do NM355_CHAIN_XCD=$X NM355_CHAIN_WGPOLL=1 python3 tools/diag_chain_stamps.py 2>&1 | grep -v amdgpu.ids; done
} > $E/${R}_rollout_ab.txt 2>&1
python3 tools/time_encode_ab.py 2>&1 | grep -v amdgpu.ids > $E/${R}_encode_ab.txt
{
  echo "headline forward step (bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline), alternating runs in one call on one device:"
  bash tools/ab_forward_env.sh "NM355_UP2C_X16=0" "NM355_UP2C_X16=1" 3
  bash tools/ab_forward_env.sh "NM355_VRNN_POST_CHAIN=1" "NM355_VRNN_POST_CHAIN=2" 3
  bash tools/ab_forward_env.sh "NM355_UP2C_ALL=0" "NM355_UP2C_ALL=1" 2
  bash tools/ab_forward_env.sh "NM355_F16P_LATE=0" "NM355_F16P_LATE=1" 3
  bash tools/ab_forward_env.sh "NM355_FAST_DECODE=0" "NM355_FAST_DECODE=1" 3
} > $E/${R}_forward_ab.txt 2>&1
{
  echo "R processes (one nm_ctx each) on ONE GPU, bench.py --ranks-on-one-gpu R --steps 12 --warmup 4: aggregate voxel-frames/s"
  for RK in 1 4 8; do
    if [ $RK = 1 ]; then A=""; else A="--ranks-on-one-gpu $RK"; fi
    python bench.py $A --steps 12 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ranks", d.get("ranks_on_one_gpu",1), "ms/step %.2f" % d["ms_per_step"], "aggregate %.0f voxel-frames/s" % d["value"], d["distributed"])'
  done
  python3 tools/host_ahead.py 20 2>&1 | grep "host per step"
} > $E/${R}_ranks_on_one_gpu.txt 2>&1
rm -rf $E/fwd $E/train $E/trainb $E/c4 $E/c5 $E/tf $E/tw $E/pm $E/pmb
# the default bench line last: its roofline.traffic reads this run's PMC summary and calibration (profiles/ on the box is a scratch copy)
mkdir -p profiles; cp $E/${R}_pmc_traffic.json $E/${R}_fetch_calib.json profiles/
python bench.py --steps 20 --warmup 5 > $E/bench_default.log 2> $E/bench_default.err
tail -1 $E/bench_default.log > $E/${R}_bench.json.log
# stamp the tree id into every summary
python3 tools/tree_id.py --stamp "$TREE" $E/${R}_*.json $E/${R}_*.csv $E/${R}_*.txt $E/${R}_bench.json.log
head -12 $E/pmc_traffic_summary.txt; head -10 $E/pmc_mfma_summary.txt; head -12 $E/pmc_mfma_train_bf16_summary.txt; tail -c 600 $E/${R}_bench.json.log
