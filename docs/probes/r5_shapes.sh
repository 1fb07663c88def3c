#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_golden_v2.py tests/test_golden_v3.py tests/test_gpu_sweeps.py -x -q -m gpu -k "search or beam or golden or sweep or duplicate or tied" 2>&1 | tail -6 ) > gpurun_out/r5_shapes_tests.log 2>&1
tail -3 gpurun_out/r5_shapes_tests.log
( timeout 600 python tests/tools/fuzz_search.py 40 5 2>&1 | tail -2 )
python scripts/r5_shape_probe.py 100 400000 1 100 8192 2>&1 | grep "^dim"
python scripts/r5_shape_probe.py 128 1000000 0 100 8192 2>&1 | grep "^dim"
python scripts/r5_shape_probe.py 768 1000000 1 100 8192 2>&1 | grep "^dim"
KEKTOR_HIP_LIB=$PWD/kektordb_amd/lib/libkektor_hip_dbgs.so python scripts/r5_shape_probe.py 100 400000 1 100 8192 2>&1 | grep "^q " | head -6
