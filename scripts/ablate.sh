#!/bin/bash
cd /root/repo
for srt in 0 1; do for d in 0 1; do echo "sorted=$srt KDB_DBG=$d"; KDB_DBG=$d timeout 120 python scripts/scale_probe.py --n 1000000 --dim 768 --nq 8192 --efs 64,64 --sorted $srt 2>&1 | grep -E '"ef"|build' | tail -2 | cut -c1-200; done; done
