cd /tmp && export TMPDIR=/tmp
for M in aware plain; do
  if [ $M = plain ]; then export NLMC_NO_BANK_AWARE=1; else unset NLMC_NO_BANK_AWARE; fi
  PRECISION=f64 W=12 timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/gpurun_out/r4/pmc_bank_$M -o q -- python3 $GRAFT_REPO_ROOT/scripts/kloop.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/r4/pmc_bank_$M/**/*counter_collection.csv", recursive=True)[-1]
acc={}
for r in csv.DictReader(open(f)):
    if "k_sweep_fused" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
print("$M", {k: sum(v)/len(v) for k,v in acc.items()})
PY
done
