# Ablations of k_sweep_fused on the bench shape with the -DNLMC_DEBUG_KNOBS build (scripts/kloop.py; NLMC_DBG_FLAGS: see SweepArgs::dbg_flags)
set -e
export NLMC_LIB=$PWD/nonlocal-monte-carlo_amd/lib/libnlmc_hip_knobs.so
for P in ${PRECS:-f64 f32}; do
for F in ${FLAGS:-0 1 2048 2 4 3 7 39 135 391}; do
  NLMC_DBG_FLAGS=$F PRECISION=$P TAG="flags=$F" timeout -k 5 120 python scripts/kloop.py
done
done
