#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for i in 1 2 3; do
for v in "KDB_FB_X=1" "KDB_FB_NOALT=1" "KDB_FB_PERIOD=1" "KDB_FB_PERIOD=2" "KDB_FB_NOALT=1 KDB_FB_PERIOD=2"; do echo "$v"; env $v python $R/scripts/flat_probe.py --bs 8192 --reps 5 2>&1 | grep "B=\|dbg"; done
done
