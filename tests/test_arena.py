"""Arena-file loader: the layout of pkg/storage/mmap/arena.go restated in a fixture writer
(64 MiB chunks, 64-byte header, dense rows, logical id -> physical slot table) and read back through the
C ABI.  The CPU test needs no GPU (kdb_arena_read_rows is host-only)."""
import os
import struct

import numpy as np
import pytest

CHUNK = 64 * 1024 * 1024
HEADER = 64
MAGIC = 0x4B414F4E  # arena.go:17


def write_arena(dirpath, rows_by_slot, dim, precision):
    """rows_by_slot: {physical slot: row}.  Files are sparse: only headers and used rows are written."""
    elem = {0: 4, 1: 2, 2: 1}[precision]
    vsize = dim * elem
    vpc = (CHUNK - HEADER) // vsize  # arena.go:90-92
    files = {}
    for slot, row in rows_by_slot.items():
        cid = slot // vpc
        if cid not in files:
            path = os.path.join(dirpath, "arena_%04d.bin" % cid)
            f = open(path, "wb+")
            f.truncate(CHUNK)  # arena.go:322-327
            f.seek(0)
            f.write(struct.pack("<III", MAGIC, 1, dim) + bytes([precision]))  # arena.go:335-342
            files[cid] = f
        f = files[cid]
        f.seek(HEADER + (slot % vpc) * vsize)  # arena.go:403
        f.write(np.ascontiguousarray(row).tobytes())
    for f in files.values():
        f.close()
    return vpc


def make_case(tmp_path, dim=2048, n=9000, precision=0, seed=3):
    rng = np.random.default_rng(seed)
    dt = {0: np.float32, 1: np.uint16, 2: np.int8}[precision]
    X = (rng.standard_normal((n, dim)) * 20).astype(dt)
    # slot table with holes and reuse, as AllocSlot/FreeSlot produce (arena.go:121-171)
    slots = rng.permutation(n + 50)[:n].astype(np.uint32)
    table = np.full(n + 1, 0xFFFFFFFF, dtype=np.uint32)
    table[1:] = slots
    dead = [7, 4000]
    for d in dead:
        table[d] = 0xFFFFFFFF
    vpc = write_arena(str(tmp_path), {int(slots[i]): X[i] for i in range(n) if (i + 1) not in dead}, dim, precision)
    return X, table, dead, vpc


def test_arena_reader_cpu(tmp_path):
    from kektordb_amd.index import arena_read_rows
    import kektordb_amd as K
    X, table, dead, vpc = make_case(tmp_path)
    n, dim = X.shape
    assert vpc == (CHUNK - HEADER) // (dim * 4) and n + 50 > vpc  # the case spans two chunk files
    assert len([f for f in os.listdir(tmp_path) if f.startswith("arena_")]) == 2
    got = arena_read_rows(str(tmp_path), dim, K.F32, 1, n, slot_table=table)
    for d in dead:
        assert not got[d - 1].any()
        X[d - 1] = 0
    assert np.array_equal(got, X)
    part = arena_read_rows(str(tmp_path), dim, K.F32, 100, 17, slot_table=table)
    assert np.array_equal(part, X[99:116])
    # header validation (arena.go:343-357): wrong dim / precision are rejected
    with pytest.raises(K.KdbError):
        arena_read_rows(str(tmp_path), dim + 1, K.F32, 1, 4, slot_table=table)
    with pytest.raises(K.KdbError):
        arena_read_rows(str(tmp_path), dim, K.F16, 1, 4, slot_table=table)
    with pytest.raises(K.KdbError):
        arena_read_rows(str(tmp_path / "missing"), dim, K.F32, 1, 4)


def test_arena_identity_slots_f16_i8(tmp_path):
    from kektordb_amd.index import arena_read_rows
    import kektordb_amd as K
    for prec, sub in ((K.F16, "h"), (K.I8, "b")):
        d = tmp_path / sub
        d.mkdir()
        rng = np.random.default_rng(prec)
        dt = {1: np.uint16, 2: np.int8}[prec]
        X = rng.integers(-100, 100, size=(300, 96)).astype(dt)
        write_arena(str(d), {i: X[i] for i in range(300)}, 96, prec)  # id i+1 -> slot i (no frees)
        assert np.array_equal(arena_read_rows(str(d), 96, prec, 1, 300), X)


@pytest.mark.gpu
def test_arena_upload_gpu(tmp_path, hip):
    X, table, dead, vpc = make_case(tmp_path, dim=2048, n=9000)
    n, dim = X.shape
    idx = hip.HipIndex(dim, 0, 0, 16, 50, capacity=n)
    idx.upload_arena(str(tmp_path), n, slot_table=table)
    for d in dead:
        X[d - 1] = 0
    assert np.array_equal(idx.download_rows(1, n), X)
    idx.set_count(n)
    q = X[:5].astype(np.float32)
    ids, dist, cnt = idx.flat_scan_batch(q, 3)
    assert ids[:, 0].tolist() == [1, 2, 3, 4, 5] and np.all(dist[:, 0] == 0.0)
