#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for v in "KDB_FB_PERIOD=4 KDB_FB_SLACK=64" "KDB_FB_PERIOD=2 KDB_FB_SLACK=64" "KDB_FB_PERIOD=1 KDB_FB_SLACK=64" "KDB_FB_PERIOD=4 KDB_FB_SLACK=128" "KDB_FB_PERIOD=2 KDB_FB_SLACK=256" "KDB_FB_PERIOD=4 KDB_FB_SLACK=16"; do
  echo -n "$v  "; env $v python $R/scripts/config_probe.py --config 3 --hnsw 0 2>&1 | tail -1
done
export KEKTOR_HIP_LIB=$R/kektordb_amd/lib/libkektor_hip_dbg.so
for v in "KDB_FB_DBG=1" "KDB_FB_DBG=32"; do echo -n "$v  "; env $v python $R/scripts/config_probe.py --config 3 --hnsw 0 2>&1 | tail -1; done
