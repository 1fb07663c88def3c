#!/bin/bash
# shared thresholds on/off on one box: parity tests, then timings at B=8192 / 1024 (1M x 768) and config 3
cd /tmp && export TMPDIR=/tmp
R=/root/repo
cd $R && timeout 900 python -m pytest tests/test_gpu_flat_big.py tests/test_gpu_flat.py -x -q 2>&1 | tail -3
cd /tmp
for i in 1 2; do
  for v in "KDB_FB_NOSHARE=1" "KDB_X=1"; do
    echo -n "$v  "; env $v python $R/scripts/flat_probe.py --bs 8192,1024 --reps 5 2>&1 | grep "B=" | tr '\n' ' '; echo
  done
done
for v in "KDB_FB_NOSHARE=1" "KDB_X=1"; do
  echo -n "cfg3 $v  "; env $v python $R/scripts/config_probe.py --config 3 --hnsw 0 2>&1 | tail -1
done
