"""A short run of scripts/fuzz_fused.py (randomised differential test: fused-window kernels vs the sweep-by-sweep
kernels on random graphs, all three schedule entry formats, flags, self-couplings, per-sweep outputs) as a child process."""
import os
import subprocess
import sys

import pytest

from conftest import REPO


@pytest.mark.gpu
def test_fused_kernels_against_the_sweep_by_sweep_kernels_on_random_cases():
    env = dict(os.environ, CASES="24", SEED="3")
    p = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "fuzz_fused.py")], capture_output=True, text=True,
                       timeout=600, env=env)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-2000:])
    assert "mismatches: 0" in p.stdout and p.stdout.count("fused_used=True") >= 20
