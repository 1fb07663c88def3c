#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for rep in 1 2; do
for cfg in ${CFGS:-"120 4" "200 4" "300 4"}; do
  set -- $cfg
  KDB_SESSION_US=$1 KDB_SLOTS=$2 timeout 600 python bench.py --no-pmc --no-cpu --legs micro_batcher --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('session_us $1 slots $2')
for k,v in (d.get('micro_batcher') or {}).items():
    if isinstance(v,dict) and 'direct' in k: print('  ',k.split('_direct')[0],v['qps'],v['per_caller_p50_ms'],v['per_caller_p99_ms'],v['gpu_calls'])"
done
done
