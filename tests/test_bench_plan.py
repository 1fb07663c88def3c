"""bench.py --dry-run: the per-rank HBM plan of a multi-GPU command, without a GPU (so that the first 8-GPU run does not die on plumbing)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--dry-run"], capture_output=True, text=True, timeout=300)
    return p.returncode, json.loads(p.stdout.strip().splitlines()[-1]), p.stderr


def test_config4_plan_fits_one_mi355x_per_rank():
    rc, plan, err = _plan("--gpus", "8", "--preset", "config4")
    assert rc == 0 and plan["fits"], (plan, err)
    assert plan["ranks"] == 8 and plan["corpus_rows_total"] == 100_000_000
    b = plan["per_rank_bytes"]
    assert b["rows_f32"] == (12_500_000 + 1) * 768 * 4                       # 38.4 GB of rows per rank
    assert 60 < plan["per_rank_total_GB"] < 0.92 * 288


def test_plan_refuses_what_does_not_fit():
    rc, plan, err = _plan("--gpus", "8", "--rows", "70000000")
    assert rc == 2 and not plan["fits"] and "more than an MI355X holds" in err
