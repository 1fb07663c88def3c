#!/bin/bash
cd /root/repo
KDB_DBG=1 timeout 100 python scripts/scale_probe.py --n 1000000 --dim 768 --nq 8192 --efs 64,64,80,100,200 2>&1 | grep -E 'kdb|"ef"' | cut -c1-175
