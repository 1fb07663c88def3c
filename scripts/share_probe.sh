#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for i in 1 2; do
echo -n "prod   "; python $R/scripts/flat_probe.py --bs 8192,1024 --reps 5 2>&1 | grep "B=" | tr '\n' ' '; echo
echo -n "dbg=0  "; KEKTOR_HIP_LIB=$R/kektordb_amd/lib/libkektor_hip_dbg.so KDB_FB_DBG=0 python $R/scripts/flat_probe.py --bs 8192,1024 --reps 5 2>&1 | grep -E "B=" | tr '\n' ' '; echo
done
echo -n "cfg3 "; python $R/scripts/flat_probe.py --n 10000000 --metric 0 --k 100 --bs 1024 --reps 3 2>&1 | grep -E "B="
cd $R && timeout 900 python -m pytest tests/test_gpu_flat_big.py -x -q 2>&1 | tail -3
