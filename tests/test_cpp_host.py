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
