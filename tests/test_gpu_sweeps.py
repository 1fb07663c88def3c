"""A short fixed-seed run of the randomised parity sweeps (tests/tools/fuzz_flat.py, fuzz_search.py) inside the GPU
suite: random shapes / metrics / precisions / k / ef / filters / deletions / near-duplicate blocks, the corner-shape
variant included; every case is compared with the oracle (bit-exact for f32 / f16, tolerance for int8)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,args", [
    ("fuzz_flat.py", ["10", "7"]),
    ("fuzz_flat.py", ["10", "8", "-", "wide"]),
    ("fuzz_search.py", ["12", "5"]),
    ("fuzz_search.py", ["16", "41", "-", "wide"]),
])
def test_randomised_sweep(tool, args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", tool)] + args, capture_output=True, text=True,
                       timeout=900)
    tail = "\n".join(p.stdout.splitlines()[-15:])
    assert p.returncode == 0, tail + "\n" + p.stderr[-2000:]
    assert "mismatching cases: 0" in p.stdout, tail
