"""The C++ host mirror (include/kektor_hip.hpp) compiles against the C ABI with plain g++ and links the shared
library; without a GPU it must fail loudly (exit 77), on the GPU box it reproduces the reference's
client_test / stress_test invariants."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_exe(tmp_path):
    import kektordb_amd
    kektordb_amd.build_library()
    exe = str(tmp_path / "host_mirror_test")
    libdir = os.path.dirname(kektordb_amd.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp"), "-L", libdir, "-lkektor_hip",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return exe


@pytest.mark.skipif(__import__("conftest").HAS_GPU, reason="CPU-only behaviour")
def test_cpp_host_mirror_links_and_fails_loudly_without_gpu(tmp_path):
    p = subprocess.run([build_exe(tmp_path)], capture_output=True, text=True)
    assert p.returncode == 77, (p.returncode, p.stdout, p.stderr)
    assert "no CPU fallback" in p.stdout


@pytest.mark.gpu
def test_cpp_host_mirror_on_gpu(tmp_path):
    # a pure C++ process: the system HIP runtime under /opt/rocm serves it (no torch in this process)
    p = subprocess.run([build_exe(tmp_path)], capture_output=True, text=True)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    assert "ok" in p.stdout


@pytest.mark.gpu
def test_micro_batcher_against_oracle(tmp_path, oracle):
    """32 threads of one-query callers through the C++ MicroBatcher (walks, filtered walks, and selective filters that
    route to the exact scan): every answer equals the CPU restatement's for the same graph, ids and scores bit for bit"""
    import struct
    import numpy as np
    import kektordb_amd
    orc = oracle
    n, dim, k = 2000, 16, 5
    rng = np.random.default_rng(7)
    X = rng.random((n, dim), dtype=np.float32)
    o = orc.OracleIndex(dim, orc.L2, orc.F32, 8, 20, seed=3)
    o.add_many(X)
    o.set_arith(orc.ARITH_HIP_WAVE)  # the walk and the scores in the GPU's accumulation order: the bar is bit-exact
    g = o.export_graph()
    even = np.zeros((n >> 6) + 1, dtype=np.uint64)
    few = np.zeros((n >> 6) + 1, dtype=np.uint64)
    for i in range(2, n + 1, 2):
        even[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    for i in range(50, n + 1, 50):  # 2 % of the ids: below the batcher's routing threshold -> exact scan
        few[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    cases = []
    for c in range(1280):
        row = c % 500 + 1
        ef = 50 if c % 2 else 12
        kind = 1 if c % 4 == 1 else 2 if c % 4 == 3 else 0
        q = X[row - 1]
        if kind == 2:
            ids, dist = o.flat_scan(q, k, allow=few)
        else:
            ids, dist = o.search(q, k, allow=even if kind == 1 else None, ef=ef)
        cases.append((row, ef, kind, np.asarray(ids, dtype=np.uint32), np.asarray(dist, dtype=np.float64)))
    path = str(tmp_path / "cases.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<6I", n, dim, k, g.max_level + 1, g.entry, len(cases)))
        f.write(X.tobytes())
        f.write(np.ascontiguousarray(g.levels, dtype=np.uint8).tobytes())
        for l in range(g.max_level + 1):
            f.write(np.ascontiguousarray(g.offsets[l], dtype=np.uint64).tobytes())
            nb = np.ascontiguousarray(g.neighbors[l], dtype=np.uint32)
            f.write(struct.pack("<Q", nb.size))
            f.write(nb.tobytes())
        f.write(even.tobytes())
        f.write(few.tobytes())
        for row, ef, kind, ids, dist in cases:
            f.write(struct.pack("<4I", row, ef, kind, ids.size))
            f.write(np.pad(ids, (0, k - ids.size)).astype(np.uint32).tobytes())
            f.write(np.pad(dist, (0, k - dist.size)).astype(np.float64).tobytes())
    kektordb_amd.build_library()
    exe = str(tmp_path / "batcher_oracle_test")
    libdir = os.path.dirname(kektordb_amd.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "batcher_oracle_test.cpp"), "-L", libdir, "-lkektor_hip",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    r = subprocess.run([exe, path], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.strip().splitlines()[-1].startswith("ok"), r.stdout   # (the lines before it report the direct callers' launches)


def test_micro_batcher_logic_under_thread_sanitizer(tmp_path):
    """the batcher's grouping / leader / turn-taking / Stop() logic on a test double, built with -fsanitize=thread
    (the reference runs its concurrency tests under the race detector): no report, every caller gets
    the answer computed from its own query, calls are coalesced"""
    import kektordb_amd
    kektordb_amd.build_library()
    exe = str(tmp_path / "batcher_logic_test")
    libdir = os.path.dirname(kektordb_amd.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "batcher_logic_test.cpp"), "-L", libdir, "-lkektor_hip",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "unexpected memory mapping" in r.stderr:
        # the sanitizer runtime cannot place its shadow under this kernel's address-space randomisation: retry without it
        r = subprocess.run(["setarch", "-R", exe], capture_output=True, text=True, timeout=300)
        if r.returncode != 0 and "unexpected memory mapping" in r.stderr:
            pytest.skip("ThreadSanitizer cannot map its shadow memory on this kernel")
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert r.stdout.startswith("ok"), r.stdout
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
