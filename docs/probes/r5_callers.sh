#!/bin/bash
# round 5: concurrent callers on one handle -- tests first, then the bench's micro_batcher leg under a few slot counts
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_cpp_host.py tests/test_gpu_parity.py -x -q -m gpu -k "batcher or concurrent or cpp_host or edge_cases or heap or tied or duplicate or above_128" 2>&1 | tail -15 ) > gpurun_out/r5_callers_tests.log 2>&1
for s in ${SLOTS:-4}; do
  KDB_SLOTS=$s timeout 900 python bench.py --no-pmc --no-cpu --legs micro_batcher --steps 5 --warmup 2 > gpurun_out/r5_callers_bench_s$s.json 2> gpurun_out/r5_callers_bench_s$s.log
done
tail -3 gpurun_out/r5_callers_tests.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5_callers_bench_s*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f)
        for k,v in (d.get('micro_batcher') or {}).items(): print('  ',k,v)
    except Exception as e: print(f,'ERR',e)
PY
for nd in "16 10000" "6 15000"; do
  set -- $nd
  KDB_NAP_DIV=$1 KDB_NAP_MIN_NS=$2 KDB_SLOTS=4 timeout 900 python bench.py --no-pmc --no-cpu --legs micro_batcher --steps 5 --warmup 2 2> gpurun_out/r5_nap_$1.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('nap div $1 min $2')
for k,v in (d.get('micro_batcher') or {}).items():
    if 'direct' in k: print('  ',k,v)"
done
