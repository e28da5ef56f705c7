"""A short run of the randomised parity soak (tools/fuzz.py): random inputs with SNPs, indels, repeats, N runs, 2-4 samples,
every alternative code path, traced and untraced, against the CPU oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_short(seed):
    env = {k: v for k, v in os.environ.items() if not k.startswith("RV_")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz.py"), "6", str(seed)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "identical to the oracle" in r.stdout


def test_fuzz_single_steps_short():
    """tools/fuzz_steps.py: splitindex-driven recursion and repeated extract on the same random inputs"""
    env = {k: v for k, v in os.environ.items() if not k.startswith("RV_")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_steps.py"), "6", "5"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "identical to the oracle" in r.stdout
