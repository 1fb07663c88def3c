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


def test_the_bench_line_stays_machine_readable():
    """Round 5's line grew to 20 KB and the driver's bounded tail of stdout could not be parsed (BENCH_r05.json parsed = null).
    The line bench.py prints is compact_record() of the full record: the full-shaped record of that very run (every side leg
    present) must come out as ONE json.loads-able line under 6 KB that still carries the contract's fields, `roofline` and
    `cpu_baseline`; the rest goes to bench_extras.json."""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "bench_r05h.json")))
    assert len(json.dumps(full)) > 15000                       # the fixture IS the oversized record
    rec, line = bench.compact_record(full)
    assert "\n" not in line and len(line.encode()) < 6000, len(line)
    back = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "rccl_ranks_seen", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "recall_at_10", "config", "roofline", "cpu_baseline", "legs"):
        assert key in back, key
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]
    assert back["config"]["pcie_inclusive_qps"] == full["config"]["pcie_inclusive_qps"] and "workload" in back["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert back["roofline"][key] == full["roofline"][key], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in back["cpu_baseline"], key
    legs = back["legs"]
    assert all(not isinstance(v, (dict, list)) for v in legs.values())            # scalars only
    for key in ("flat_frac", "flat_kernel_ms", "heap_order_ms", "callers64_qps", "callers64_p50_ms", "callers64_flag_qps", "ref_graph_qps", "ref_graph_frac"):
        assert key in legs, key
    assert any(k.startswith("shape_") and k.endswith("_frac") for k in legs)
    # a record with a hundred more legs still fits: legs are dropped from the end, the contract never
    fat = dict(full, reference_benchmark_shapes={"shapes": [{"shape": f"Shape-{i} cosine, M=16 efS={i}", "frac_of_hbm_peak": 0.1, "kernel_ms": 1.0} for i in range(400)]})
    rec2, line2 = bench.compact_record(fat)
    assert len(line2.encode()) <= 6000 and json.loads(line2)["roofline"]["frac"] == full["roofline"]["frac"]
