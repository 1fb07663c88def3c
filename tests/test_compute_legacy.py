"""The ten symbols of the reference's existing native library (native/compute/include/kektordb_compute.h:8-24), exported
by libkektor_hip.so and libkektordb_compute.a so that the reference's `-tags rust` build links unchanged
(pkg/core/distance/distance_rust.go:12-17).  Host-only code: runs without a GPU.

Checked against the known answers the reference's own tests hold (tests/golden/reference_kats.json:
pkg/core/distance/distance_test.go:37-84, native/compute/src/lib.rs:423-458) and, bit for bit, against the oracle's
restatement of the crate's AVX2/FMA arithmetic (ORC_ARITH_RUST) on random vectors of every tail length.
dot_product_i8 is checked against the exact integer dot product (= the default Go build's dotProductGoInt8): the crate's own
AVX2 reduction drops two of the four 32-bit lanes for len >= 32 (lib.rs:171-176), which is deliberately not reproduced
(include/kektor_compute_legacy.h)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(hip):
    L = C.CDLL(hip.LIB_PATH)
    L.squared_euclidean_f32.restype = C.c_float
    L.dot_product_f32.restype = C.c_float
    L.squared_euclidean_f16.restype = C.c_float
    L.dot_product_i8.restype = C.c_int32
    for f in (L.squared_euclidean_f32, L.dot_product_f32, L.squared_euclidean_f16, L.dot_product_i8):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_reference_known_answers(lib):
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")))
    d = kat["distance_test.go:37-84"]
    a, b, want = d["l2_f32"]
    x, y = np.array(a, np.float32), np.array(b, np.float32)
    assert lib.squared_euclidean_f32(_p(x), _p(y), x.size) == want
    a, b, want = d["l2_f16"]
    x, y = np.array(a, np.float16).view(np.uint16), np.array(b, np.float16).view(np.uint16)
    assert lib.squared_euclidean_f16(_p(x), _p(y), x.size) == want
    a, b, want = d["dot_i8"]
    x, y = np.array(a, np.int8), np.array(b, np.int8)
    assert lib.dot_product_i8(_p(x), _p(y), x.size) == want
    v, want, tol = d["cosine_f32_self"]
    x = np.array(v, np.float32)
    x /= np.linalg.norm(x)
    assert abs(1.0 - lib.dot_product_f32(_p(x), _p(x), x.size) - want) <= tol
    r = kat["lib.rs:423-458"]
    a, b, want = r["dot_f32"]
    x, y = np.array(a, np.float32), np.array(b, np.float32)
    assert lib.dot_product_f32(_p(x), _p(y), x.size) == want
    a, b, want = r["dot_i8_neg"]
    x, y = np.array(a, np.int8), np.array(b, np.int8)
    assert lib.dot_product_i8(_p(x), _p(y), x.size) == want


def test_bits_match_the_crates_arithmetic(lib, oracle):
    """every length 0..40 (all tail shapes) and a few long ones: the same f32 bits / i32 as the oracle's restatement of
    native/compute/src/lib.rs (8-lane FMA, fold, scalar tail)"""
    O = oracle
    rng = np.random.default_rng(5)
    for n in list(range(0, 41)) + [127, 128, 129, 768, 1000, 1536]:
        x = rng.standard_normal(n).astype(np.float32)
        y = rng.standard_normal(n).astype(np.float32)
        hx, hy = x.astype(np.float16).view(np.uint16), y.astype(np.float16).view(np.uint16)
        ix = rng.integers(-128, 128, n).astype(np.int8)
        iy = rng.integers(-128, 128, n).astype(np.int8)
        got = (np.float32(lib.squared_euclidean_f32(_p(x), _p(y), n)), np.float32(lib.dot_product_f32(_p(x), _p(y), n)),
               np.float32(lib.squared_euclidean_f16(_p(hx), _p(hy), n)), int(lib.dot_product_i8(_p(ix), _p(iy), n)))
        OL = O.lib()
        fx, fy = hx.view(np.float16).astype(np.float32), hy.view(np.float16).astype(np.float32)  # exact widening
        want = (np.float32(OL.orc_l2_f32_avx2(_p(x), _p(y), n)), np.float32(OL.orc_dot_f32_avx2(_p(x), _p(y), n)),
                np.float32(OL.orc_l2_f32_avx2(_p(fx), _p(fy), n)), int(OL.orc_dot_i8(_p(ix), _p(iy), n)))
        assert got[0].view(np.uint32) == want[0].view(np.uint32), n
        assert got[1].view(np.uint32) == want[1].view(np.uint32), n
        assert got[2].view(np.uint32) == want[2].view(np.uint32), n
        assert got[3] == want[3] == int(np.dot(ix.astype(np.int64), iy.astype(np.int64))), n


def test_embedder_stubs_report_no_model(lib):
    lib.kektordb_embed_init.restype = C.c_int
    lib.kektordb_embed.restype = C.c_int
    lib.kektordb_embed_batch.restype = C.c_int
    assert lib.kektordb_embed_init(b"model.onnx", b"tokenizer.json") == -1
    vec, dim = C.POINTER(C.c_float)(), C.c_int(7)
    assert lib.kektordb_embed(b"hello", C.byref(vec), C.byref(dim)) == -1 and dim.value == 0 and not vec
    vecs, cnt = C.POINTER(C.POINTER(C.c_float))(), C.c_int(3)
    texts = (C.c_char_p * 2)(b"a", b"b")
    assert lib.kektordb_embed_batch(texts, 2, C.byref(vecs), C.byref(cnt), C.byref(dim)) == -1 and cnt.value == 0
    lib.kektordb_free_embedding(None, 0)
    lib.kektordb_free_embeddings(None, 0, 0)
    lib.kektordb_embed_destroy()


def test_reference_style_link(tmp_path):
    """what `#cgo LDFLAGS: -lkektordb_compute -lstdc++` does: a C program that includes a header with the reference's
    prototypes links against the static archive and runs"""
    src = tmp_path / "t.c"
    src.write_text('#include "kektor_compute_legacy.h"\n#include <stdio.h>\nint main(void){float x[2]={1,2},y[2]={3,4};'
                   'int8_t a[2]={10,20},b[2]={2,3};printf("%g %g %d %d\\n",squared_euclidean_f32(x,y,2),dot_product_f32(x,y,2),'
                   'dot_product_i8(a,b,2),kektordb_embed_init("m","t"));return 0;}\n')
    exe = tmp_path / "t"
    subprocess.run(["gcc", str(src), "-I", os.path.join(ROOT, "include"), "-L", os.path.join(ROOT, "kektordb_amd", "lib"),
                    "-lkektordb_compute", "-lstdc++", "-lm", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out == ["8", "11", "80", "-1"]
