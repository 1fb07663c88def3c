#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_golden_v2.py tests/test_golden_v3.py -x -q -m gpu -k "heap or tied or tie or duplicate or golden or int8" 2>&1 | tail -15 ) > gpurun_out/r5_heap_tests.log 2>&1
tail -4 gpurun_out/r5_heap_tests.log
( timeout 600 python tests/tools/fuzz_search.py 2>&1 | tail -5 ) > gpurun_out/r5_heap_fuzz.log 2>&1
tail -3 gpurun_out/r5_heap_fuzz.log
timeout 900 python bench.py --no-pmc --no-cpu --legs heap_order --steps 5 --warmup 2 2> gpurun_out/r5_heap_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(d.get('heap_order'),indent=1)); print('headline ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"
