#!/bin/bash
# round 5: the heap-order pass beside the search kernel (KDB_HEAP_OVERLAP_MIN_B / _WG): parity subset (TESTS=1), then the heap_order bench leg per setting
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
if [ -n "$TESTS" ]; then
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_golden_v2.py tests/test_golden_v3.py -x -q -m gpu --poison -k "heap or tied or tie or duplicate or golden or int8" 2>&1 | tail -6 ) > gpurun_out/r5_overlap_tests.log 2>&1
tail -3 gpurun_out/r5_overlap_tests.log
fi
for cfg in ${CFGS:-0_1 4096_1 4096_2 4096_3}; do
  a=${cfg%_*}; b=${cfg#*_}
  echo "== KDB_HEAP_OVERLAP_MIN_B=$a KDB_HEAP_OVERLAP_WG=$b"
  KDB_HEAP_OVERLAP_MIN_B=$a KDB_HEAP_OVERLAP_WG=$b timeout 600 python bench.py --no-pmc --no-cpu --legs heap_order --steps 5 --warmup 2 > gpurun_out/r5_overlap_bench_$cfg.json 2> gpurun_out/r5_overlap_bench_$cfg.log
  python - gpurun_out/r5_overlap_bench_$cfg.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(json.dumps(d.get("heap_order"))); print("headline ms", d["ms_per_step"])
PY
done
