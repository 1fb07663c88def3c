#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for cfg in "4 10 15000" "8 16 10000" "12 16 10000" "8 24 8000"; do
  set -- $cfg
  KDB_SPIN_WATCHERS=$1 KDB_NAP_DIV=$2 KDB_NAP_MIN_NS=$3 timeout 600 python bench.py --no-pmc --no-cpu --legs micro_batcher --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('spin $1 div $2 min $3')
for k,v in (d.get('micro_batcher') or {}).items():
    if isinstance(v,dict) and ('64_' in k or '256_' in k or '16_' in k) and 'direct' in k: print('  ',k,v['qps'],v['per_caller_p50_ms'],v['per_caller_p99_ms'])"
done
