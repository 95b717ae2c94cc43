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
python bench.py --steps 20 --warmup 5 > $E/bench_default.log 2> $E/bench_default.err
tail -1 $E/bench_default.log > $E/${R}_bench.json.log
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
# same-call A/Bs of the round: posterior chain (encode alone, and forced on inside the fused forward), one-product conv kernels; training step with the
# round's one-product changes switched off
python3 tools/time_encode_ab.py 2>&1 | grep -v amdgpu.ids > $E/${R}_encode_ab.txt
python3 tools/ab_f16q2.py 64 2>&1 | grep -v amdgpu.ids > $E/${R}_f16q2_ab.txt
python3 tools/ab_f16r.py 64 2>&1 | grep -v amdgpu.ids > $E/${R}_f16r_ab.txt
if [ -f neural_marionette_amd/libnm355_diag.so ]; then python3 tools/diag_f16q2.py 0,1,2,3,5,6,7,8,9,0 2>&1 | grep -v amdgpu.ids >> $E/${R}_f16q2_ab.txt; fi
for M in bf16 f16; do
  for SW in "" "NM355_F16R=0 NM355_UP2_MAT=0" "NM355_F16R=0 NM355_UP2_MAT=0 NM355_F16Q2=0"; do
    echo "train $M [$SW]: $(env $SW python3 bench.py --workload train --conv-mode $M --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])') ms per step" >> $E/${R}_train_ab.txt
  done
done
rm -rf $E/fwd $E/train $E/trainb $E/c4 $E/c5 $E/tf $E/tw $E/pm $E/pmb
# stamp the tree id into every summary
python3 tools/tree_id.py --stamp "$TREE" $E/${R}_*.json $E/${R}_*.csv $E/${R}_*.txt $E/${R}_bench.json.log
head -12 $E/pmc_traffic_summary.txt; head -10 $E/pmc_mfma_summary.txt; head -12 $E/pmc_mfma_train_bf16_summary.txt; tail -c 600 $E/${R}_bench.json.log
