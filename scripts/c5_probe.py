"""Config 5 (10M x 1536 cosine, 100 one-percent filters, 1024 queries grouped by filter): the grouped exact scan timed under the
measurement knobs of flat_scan_small_kernel (KDB_FSS_CS: steps per register chunk, 0 = the generic instantiation; KDB_GROUP_STRIPES).
The answers of every setting must be identical (a signature is printed).  With the measurement build (make -C kektordb_amd/csrc dbg;
KEKTOR_HIP_LIB=kektordb_amd/lib/libkektor_hip_dbg.so) the KDB_FSS_DBG switches are walked instead: what each part of the kernel costs.
    python scripts/c5_probe.py [rows]"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import kektordb_amd as K  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    idx, Q, Qs, cats, offs, allowed, total, d_lists = bench.c5_case(K, dev, g, n, 1024)
    o = bench.outs(1024, 10, dev)
    alg = total * 1536 * 2 + 1024 * 1536 * 2 + 1024 * 10 * 8
    tiles = sum((int(offs[j + 1] - offs[j]) + 15) // 16 for j in range(len(cats)))
    print(f"{len(cats)} filters, {tiles} sixteen-query tiles, {total} allowed rows, {alg / 1e9:.2f} GB algorithmic")
    idx.flat_scan_groups_dev(Qs, 10, offs, d_lists, *o, max_total_allowed=total)  # (makes the half-precision ranking copy)
    idx.sync()
    print(f"ceilings on THIS table (10M x 3072-byte half-precision rows, every row read about once -- nothing for the 256 MB memory-side "
          f"cache to reuse): uniform random whole-row gather {idx.probe_gather(10_000_000, shadow=True):.0f} GB/s, "
          f"one coalesced pass {idx.probe_stream(shadow=True):.0f} GB/s")
    knobs = ("KDB_FSS_CS", "KDB_FSS_DBG", "KDB_GROUP_STRIPES")
    cases = [{}, {"KDB_FSS_CS": "6"}, {"KDB_FSS_CS": "0"}, {"KDB_GROUP_STRIPES": "16"}, {"KDB_GROUP_STRIPES": "32"}, {"KDB_GROUP_STRIPES": "48"}, {}]
    if "dbg" in os.environ.get("KEKTOR_HIP_LIB", ""):  # the measurement build: parts of the kernel switched off (answers are wrong)
        cases = [{"KDB_FSS_DBG": d} for d in ("0", "1", "2", "4", "6", "10", "14")]  # (8 alone would select with unloaded ids)
    for case in cases:
        for kn in knobs:
            os.environ.pop(kn, None)
        os.environ.update(case)
        for _ in range(2):
            idx.flat_scan_groups_dev(Qs, 10, offs, d_lists, *o, max_total_allowed=total)
        idx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            idx.flat_scan_groups_dev(Qs, 10, offs, d_lists, *o, max_total_allowed=total)
        idx.sync()
        wall = (time.perf_counter() - t0) / 5
        kms = float(np.mean([x["kernel_ms"] for x in idx.launch_stats(5)]))
        sig = hashlib.sha1(o[0].cpu().numpy().tobytes() + o[1].cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"{' '.join(f'{a}={b}' for a, b in case.items()) or 'defaults':<50}: kernel {kms:.3f} ms = {alg / kms / 1e6:.0f} GB/s ({alg / kms / 1e6 / 8000:.3f} of peak), "
              f"call {wall * 1e3:.3f} ms, answers {sig}")


if __name__ == "__main__":
    main()
