#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
# where a walk's time goes: the measurement build prints per-query phase cycles (64 queries, one and four waves per query)
KEKTOR_HIP_LIB=$R/kektordb_amd/lib/libkektor_hip_dbgs.so KDB_WIDE_MAX_B=0 python $R/scripts/wide_probe.py --only 64 --reps 1 2>&1 | grep -E "^q " | sort -t' ' -k2 -n | awk 'NR<=200' > $R/gpurun_out/walk_timers_1wave.txt
KEKTOR_HIP_LIB=$R/kektordb_amd/lib/libkektor_hip_dbgs.so python $R/scripts/wide_probe.py --only 64 --reps 1 2>&1 | grep -E "^q " | awk 'NR<=200' > $R/gpurun_out/walk_timers_4wave.txt
tail -70 $R/gpurun_out/walk_timers_1wave.txt | head -12; echo; tail -70 $R/gpurun_out/walk_timers_4wave.txt | head -12
cd $R && timeout 600 python -m pytest tests/test_cpp_host.py tests/test_gpu_flat_big.py -x -q 2>&1 | tail -3
cd /tmp
for i in 1 2; do
  for v in "KDB_FB_NOSHARE=1" "KDB_X=1"; do
    echo -n "$v  "; env $v python $R/scripts/flat_probe.py --bs 8192,1024 --reps 5 2>&1 | grep "B=" | tr '\n' ' '; echo
  done
done
KEKTOR_HIP_LIB=$R/kektordb_amd/lib/libkektor_hip_dbg.so KDB_FB_DBG=32 python $R/scripts/flat_probe.py --bs 8192 --reps 2 2>&1 | grep -E "B=|dbg"
