#!/bin/bash
# Profiling pass of one round on the GPU box:  bash scripts/profile_round.sh <tag>   (run through gpurun)
# Per bench leg (f64 = the headline, f32 = the fixed-point leg): one --stats pass and five --pmc passes (counters in passes of
# their own, never combined with other trace domains), on a shorter run of the same workload (512 rounds per step).
set -e
TAG=${1:-r01_x}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export NLMC_BENCH_ROUNDS_PER_STEP=512
for LEG in f64 f32; do
  B="python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-leg --headline $LEG"
  timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_$LEG" -o s -- $B > "$OUT/stats_$LEG.log" 2>&1
  timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch_$LEG" -o f -- $B > "$OUT/fetch_$LEG.log" 2>&1
  timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d "$OUT/write_$LEG" -o w -- $B > "$OUT/write_$LEG.log" 2>&1
  timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d "$OUT/sq1_$LEG" -o q -- $B > "$OUT/sq1_$LEG.log" 2>&1
  timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d "$OUT/sq2_$LEG" -o q -- $B > "$OUT/sq2_$LEG.log" 2>&1
  timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES -d "$OUT/sq3_$LEG" -o q -- $B > "$OUT/sq3_$LEG.log" 2>&1
  python3 "$REPO/scripts/summarize_prof.py" "$TAG" "$LEG" "$OUT/stats_$LEG" "$OUT/fetch_$LEG" "$OUT/write_$LEG" "$OUT/sq1_$LEG" "$OUT/sq2_$LEG" "$OUT/sq3_$LEG"
  echo "leg $LEG profiled"
done
unset NLMC_BENCH_ROUNDS_PER_STEP
if [ -z "$SKIP_SECONDARY" ]; then
# secondary kernels: backbone inference (k_lbp) and the APT + iso-cluster round (C5)
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/lbp" -o s -- python3 $REPO/scripts/lbp_throughput.py > "$OUT/lbp.log" 2>&1
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/c5" -o s -- python3 $REPO/scripts/c5_only.py > "$OUT/c5.log" 2>&1
RESTARTS=8 timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/c3nmc" -o s -- python3 $REPO/scripts/npt_nmc_throughput.py > "$OUT/c3nmc.log" 2>&1
cp "$OUT/c3nmc/s_kernel_stats.csv" "$REPO/profiles/${TAG}_c3nmc_kernel_stats.csv"
cp "$OUT/lbp/s_kernel_stats.csv" "$REPO/profiles/${TAG}_lbp_kernel_stats.csv"
cp "$OUT/c5/s_kernel_stats.csv" "$REPO/profiles/${TAG}_c5_kernel_stats.csv"
fi
cp "$REPO"/profiles/${TAG}_* "$REPO"/profiles/current_sweep_pmc.json "$OUT"/
python3 "$REPO/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
cp "$OUT/bench.json" "$REPO/profiles/${TAG}_bench_n1.json"
cp "$REPO/profiles/${TAG}_bench_n1.json" "$OUT"/
tail -1 "$OUT/bench.json"
