#!/bin/bash
# (stats pass = the default bench command; PMC passes use a shorter run of the same workload)
# Profiling pass of one round on the GPU box:  bash scripts/profile_round.sh <tag>   (run through gpurun)
set -e
TAG=${1:-r01_x}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f64-leg"
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats" -o s -- python3 $REPO/bench.py --no-cpu-baseline > "$OUT/stats.log" 2>&1
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o f -- $B > "$OUT/fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o w -- $B > "$OUT/write.log" 2>&1
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d "$OUT/sq1" -o q -- $B > "$OUT/sq1.log" 2>&1
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d "$OUT/sq2" -o q -- $B > "$OUT/sq2.log" 2>&1
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES -d "$OUT/sq3" -o q -- $B > "$OUT/sq3.log" 2>&1
python3 "$REPO/scripts/summarize_prof.py" "$TAG" "$OUT/stats" "$OUT/fetch" "$OUT/write" "$OUT/sq1" "$OUT/sq2" "$OUT/sq3"
# secondary kernels: backbone inference (k_lbp) and the APT + iso-cluster round (C5)
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/lbp" -o s -- python3 $REPO/scripts/lbp_throughput.py > "$OUT/lbp.log" 2>&1
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/c5" -o s -- python3 $REPO/scripts/c5_only.py > "$OUT/c5.log" 2>&1
RESTARTS=8 timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/c3nmc" -o s -- python3 $REPO/scripts/npt_nmc_throughput.py > "$OUT/c3nmc.log" 2>&1
cp "$OUT/c3nmc/s_kernel_stats.csv" "$REPO/profiles/${TAG}_c3nmc_kernel_stats.csv"
cp "$OUT/lbp/s_kernel_stats.csv" "$REPO/profiles/${TAG}_lbp_kernel_stats.csv"
cp "$OUT/c5/s_kernel_stats.csv" "$REPO/profiles/${TAG}_c5_kernel_stats.csv"
cp "$REPO"/profiles/${TAG}_* "$REPO"/profiles/current_sweep_pmc.json "$OUT"/
python3 "$REPO/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -1 "$OUT/bench.json"
