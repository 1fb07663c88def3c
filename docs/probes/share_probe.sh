#!/bin/bash
# two builds interleaved on one box: the big-tile flat scan at 8192 / 1024 queries and config 3, after the parity tests
cd /tmp && export TMPDIR=/tmp
R=/root/repo
cd $R && timeout 900 python -m pytest tests/test_gpu_flat_big.py -x -q 2>&1 | tail -3
cd /tmp
for i in 1 2 3; do
  for lib in libkektor_hip_prev.so libkektor_hip.so; do
    echo -n "$lib  "; KEKTOR_HIP_LIB=$R/kektordb_amd/lib/$lib python $R/scripts/flat_probe.py --bs 8192,1024 --reps 5 2>&1 | grep "B=" | tr '\n' ' '; echo
  done
done
for lib in libkektor_hip_prev.so libkektor_hip.so; do
echo -n "cfg3 $lib "; KEKTOR_HIP_LIB=$R/kektordb_amd/lib/$lib python $R/scripts/flat_probe.py --n 10000000 --metric 0 --k 100 --bs 1024 --reps 3 2>&1 | grep -E "B="
done
