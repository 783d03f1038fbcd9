#!/usr/bin/env python3
"""Collect rocprofv3 outputs of one profiling round into the small CSVs kept under profiles/.

    python scripts/summarize_prof.py <tag> <leg> <stats_dir> <fetch_dir> <write_dir> [<sq_dir> ...]

<leg>: "f64" (the headline leg of bench.py) or "f32" (its fixed-point leg): file names and the key of current_sweep_pmc.json.

<stats_dir>: rocprofv3 --kernel-trace --stats;  <fetch_dir>/<write_dir>: separate --pmc FETCH_SIZE / WRITE_SIZE passes;
<sq_dir>: optional further --pmc passes.  Units and the gfx950 read correction follow MI355X_MICROARCH.md (HBM section).
"""
import csv
import glob
import os
import sys
from collections import defaultdict

KERNELS = ("k_sweep_fused", "k_rounds_fused", "k_sweep_philox")     # dominant kernel: the first of these that appears in the trace


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits[-1]


def counters(d):
    with open(find(d, "counter_collection.csv")) as f:
        rows = list(csv.DictReader(f))
    kernel = next(k for k in KERNELS if any(k in r["Kernel_Name"] for r in rows))
    acc, n = defaultdict(float), defaultdict(int)
    for row in rows:
        if kernel in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
            n[row["Counter_Name"]] += 1
    out = {k: (n[k], acc[k] / n[k]) for k in acc}
    out["__kernel__"] = kernel
    return out


def main():
    tag, leg, stats_dir, fetch_dir, write_dir, *sq_dirs = sys.argv[1:]
    cmdline = f"NLMC_BENCH_ROUNDS_PER_STEP=512 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-leg --headline {leg}"
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    with open(find(stats_dir, "kernel_stats.csv")) as f, open(os.path.join(out, f"{tag}_{leg}_bench_n1_kernel_stats.csv"), "w") as g:
        g.write(f.read())
    fc = counters(fetch_dir)
    kname = fc["__kernel__"]
    fs, ws = fc["FETCH_SIZE"], counters(write_dir)["WRITE_SIZE"]
    hbm = int(round((2.0 * fs[1] + ws[1]) * 1024))
    with open(os.path.join(out, f"{tag}_{leg}_sweep_hbm_traffic_pmc.csv"), "w") as g:
        g.write(f"# rocprofv3 --kernel-trace --pmc FETCH_SIZE  and (separate pass)  --pmc WRITE_SIZE  -- {cmdline}\n"
                f"# per launch of {kname} (256 chains x 1e4 spins x 10 sweeps); counter unit = KiB\n"
                "# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of wide coalesced reads -> "
                "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024\n"
                "counter,n_launches,mean_per_launch\n"
                f"FETCH_SIZE,{fs[0]},{fs[1]:.1f}\nWRITE_SIZE,{ws[0]},{ws[1]:.1f}\n"
                f"# hbm_bytes_per_launch,{hbm}\n# algorithmic_bytes_per_launch,{256 * 10_000 * 10 * (91 if leg == 'f64' else 63)}\n")
    if sq_dirs:
        with open(os.path.join(out, f"{tag}_{leg}_sweep_pmc_summary.csv"), "w") as g:
            g.write(f"# rocprofv3 --kernel-trace --pmc <counters, one pass per line group> -- {cmdline}\n"
                    f"# {kname}, per launch (256 chains x 1e4 spins x 10 sweeps)\n"
                    "counter,mean_per_launch\n")
            for d in sq_dirs:
                for k, val in sorted(counters(d).items()):
                    if k != "__kernel__":
                        g.write(f"{k},{val[1]:.6g}\n")
    # the figures bench.py quotes in its roofline objects (PMC cannot be read live)
    import json
    allc = {}
    for d in sq_dirs:
        allc.update({k: v[1] for k, v in counters(d).items() if k != "__kernel__"})
    cur = {"source": f"profiles/{tag}_{leg}_sweep_pmc_summary.csv, profiles/{tag}_{leg}_sweep_hbm_traffic_pmc.csv (rocprofv3 --pmc, separate passes)",
           "kernel": kname, "hbm_bytes_per_launch": hbm}
    if "SQ_INSTS_VALU" in allc:
        cur["valu_wave_insts_per_launch"] = allc["SQ_INSTS_VALU"]
    if "SQ_INSTS_LDS" in allc:
        cur["lds_wave_insts_per_launch"] = allc["SQ_INSTS_LDS"]
    if "SQ_INSTS_VMEM_RD" in allc:
        cur["vmem_rd_wave_insts_per_launch"] = allc["SQ_INSTS_VMEM_RD"]
    if allc.get("SQ_LDS_IDX_ACTIVE"):
        cur["lds_bank_conflict_frac"] = allc["SQ_LDS_BANK_CONFLICT"] / allc["SQ_LDS_IDX_ACTIVE"]
    if allc.get("SQ_WAVE_CYCLES"):
        cur["wait_any_frac"] = allc["SQ_WAIT_ANY"] / allc["SQ_WAVE_CYCLES"]
    path = os.path.join(out, "current_sweep_pmc.json")
    try:
        with open(path) as g:
            both = json.load(g)
        if "kernel" in both:               # the one-leg layout of earlier rounds
            both = {}
    except (OSError, ValueError):
        both = {}
    both[leg] = cur
    with open(path, "w") as g:
        json.dump(both, g, indent=1)
    print("hbm_bytes_per_launch", hbm)


if __name__ == "__main__":
    main()
