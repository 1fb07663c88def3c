"""bench.py launched the way the driver launches it: `python bench.py --gpus N` with NO launcher around it must start N ranks
itself (and say so in the line: n_gpus == N, rccl_ranks_seen == N), `--cluster` must drive N shards through ONE
kdb_cluster handle, and a run that cannot reach N ranks must refuse instead of benching fewer GPUs under the asked-for label
(task 1).  On the 1-GPU test box the ranks share device 0 under gloo; with two visible GPUs the RCCL variant
runs as well."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--rows", "20000", "--dim", "64", "--batch", "512", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu", "--efc", "60",
         "--build-batch", "2048"]


def _run(args, timeout=600):
    env = dict(os.environ)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):   # no launcher: the driver's plain command
        env.pop(v, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_bench_gpus_2_starts_its_own_ranks(backend):
    import torch
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("two RCCL ranks need two visible GPUs")
    p, line = _run(["--gpus", "2", "--backend", backend] + SMALL)
    assert p.returncode == 0, p.stderr[-2000:]
    assert line is not None, p.stdout[-2000:]
    assert line["n_gpus"] == 2 and line["rccl_ranks_seen"] == 2
    assert line["config"]["total_rows"] == 40000 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["recall_at_10"] >= 0.9
    assert "starting 2 ranks" in p.stderr


def test_bench_cluster_two_shards_one_process():
    p, line = _run(["--gpus", "2", "--cluster"] + SMALL)
    assert p.returncode == 0, p.stderr[-2000:]
    assert line is not None, p.stdout[-2000:]
    import torch
    nd = min(2, torch.cuda.device_count())
    assert line["n_gpus"] == 2 and line["devices_used"] == nd and line["rccl_ranks_seen"] == nd
    assert line["route"].startswith("single process")
    assert line["value"] > 0 and line["recall_at_10"] >= 0.9
    assert line["config"]["total_rows"] == 40000


def test_bench_reference_benchmark_shapes():
    """`bench.py --shapes`: the reference's published benchmarks (GloVe-100/200/300 cosine, SIFT-1M L2 at the published M /
    efConstruction / efSearch) on synthetic rows of those shapes -- built on the GPU, searched, recall measured against the exact
    scan of the same index.  Every shape must reach the recall the reference publishes for it."""
    p, line = _run(["--shapes"], timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert line is not None and len(line["shapes"]) == 6
    for sh in line["shapes"]:
        published = float(sh["reference_published_recall_qps"].split("/")[0])
        assert sh["recall_at_10"] >= published, sh
        assert sh["qps"] > 1000 * float(sh["reference_published_recall_qps"].split("/")[1].split()[0]) / 10, sh


def test_bench_refuses_a_world_smaller_than_asked():
    """a launcher that started ONE rank for --gpus 2 (what round 3's bench silently accepted): exit code 2, no JSON line"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"] + SMALL, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2
    assert "refusing" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_cluster_poisoned_after_a_failure_inside_an_rccl_group(oracle, hip):
    """an injected failure inside the all-gather group (the collective queued on some devices and not on the last) must come
    back as an error, abort the communicators, and make the NEXT call fail at once -- not hang on a collective that waits for
    a peer that never joined.  A failure BETWEEN collectives (a shard refusing its arguments) leaves the handle usable."""
    import time
    from conftest import make_corpus
    O = oracle
    n, dim, k = 3000, 32, 5
    X = make_corpus(n, dim, "normal", seed=5)
    shards, bases = [], []
    for g in range(2):
        orc = O.OracleIndex(dim, 0, O.F32, 16, 40, seed=g)
        orc.add_many(X[g * 1500:(g + 1) * 1500])
        idx = hip.HipIndex(dim, 0, 0, 16, 40, capacity=1508)
        idx.upload_rows(orc.rows()[1:], 1)
        idx.upload_graph_obj(orc.export_graph())
        shards.append(idx)
        bases.append(g * 1500)
    Q = make_corpus(16, dim, "normal", seed=6)
    for stage in (2, 1):
        cl = hip.Cluster(shards, bases)
        good = cl.search_batch(Q, k, 40)
        assert cl.comm_info() == {"ranks_in_communicator": 1, "poisoned": False}
        # between collectives: a k the exact scan refuses (first shard) -- nothing inconsistent, the handle stays usable
        with pytest.raises(hip.KdbError):
            cl.flat_scan_batch(Q, 5000)
        assert not cl.comm_info()["poisoned"]
        again = cl.search_batch(Q, k, 40)
        assert all(np.array_equal(a, b) for a, b in zip(good, again))
        cl.debug_fail_next(stage)
        with pytest.raises(hip.KdbError) as e1:
            cl.search_batch(Q, k, 40)
        assert "injected" in str(e1.value)
        assert cl.comm_info()["poisoned"]
        t0 = time.perf_counter()
        with pytest.raises(hip.KdbError) as e2:
            cl.search_batch(Q, k, 40)
        assert time.perf_counter() - t0 < 1.0
        assert "poisoned" in str(e2.value)
        with pytest.raises(hip.KdbError):
            cl.flat_scan_batch(Q, k)
        cl.close()
    # the shards themselves are untouched: a NEW cluster over them answers as before
    cl = hip.Cluster(shards, bases)
    fresh = cl.search_batch(Q, k, 40)
    assert all(np.array_equal(a, b) for a, b in zip(good, fresh))
    cl.close()
