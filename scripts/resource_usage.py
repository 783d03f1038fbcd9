"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr of the build) as one line per kernel:
registers, spills, scratch, occupancy, LDS.  Usage: python scripts/resource_usage.py [remarks.txt]; without a file the
library is rebuilt into a temporary path with the remarks switched on (the product library is not touched)."""
import os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def remarks():
    if len(sys.argv) > 1:
        return open(sys.argv[1]).read()
    with tempfile.TemporaryDirectory() as d:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(d, "lib.so"),
               os.path.join(REPO, "nonlocal-monte-carlo_amd", "csrc", "nlmc.hip")]
        return subprocess.run(cmd, capture_output=True, text=True, check=True).stderr


def main():
    rows, cur = [], None
    for line in remarks().splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    try:
        names = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + [r["name"] for r in rows], capture_output=True,
                               text=True, check=True).stdout.splitlines()
    except Exception:  # noqa: BLE001
        names = [r["name"] for r in rows]
    print(f"{'kernel':<58} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch B/lane':>14} {'VGPR spill':>10} {'SGPR spill':>10} {'waves/SIMD':>10} {'LDS B':>7}")
    for r, nm in zip(rows, names):
        nm = re.sub(r"\(.*\)$", "", nm).replace("void ", "")
        print(f"{nm:<58} {r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('TotalSGPRs', r.get('SGPRs', '?')):>5} "
              f"{r.get('ScratchSize [bytes/lane]', '?'):>14} {r.get('VGPRs Spill', r.get('VGPR Spill', '?')):>10} "
              f"{r.get('SGPRs Spill', r.get('SGPR Spill', '?')):>10} {r.get('Occupancy [waves/SIMD]', '?'):>10} {r.get('LDS Size [bytes/block]', '?'):>7}")


if __name__ == "__main__":
    main()
