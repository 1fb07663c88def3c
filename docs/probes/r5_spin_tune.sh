#!/bin/bash
# round 5: how many watchers of the completion words may spin (KDB_SPIN_WATCHERS) and nap length (KDB_NAP_DIV), micro_batcher leg
cd "$(dirname "$0")/.." || exit 1
for rep in 1 2; do
for cfg in ${CFGS:-4_10 0_10 1_10 0_6}; do
  sp=${cfg%_*}; nd=${cfg#*_}
  KDB_SPIN_WATCHERS=$sp KDB_NAP_DIV=$nd timeout 600 python bench.py --no-pmc --no-cpu --legs micro_batcher --steps 5 --warmup 2 2>/dev/null > /tmp/spin_$cfg.json
  python - /tmp/spin_$cfg.json "$cfg" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("spin_napdiv", sys.argv[2])
for k, v in (d.get("micro_batcher") or {}).items():
    if isinstance(v, dict) and ("64_" in k or "256_" in k or k.startswith("1_")): print("  ", k[:24], v["qps"], v["per_caller_p50_ms"], v["per_caller_p99_ms"], v["combined_call_us"]["naps_per_call"])
PY
done
done
