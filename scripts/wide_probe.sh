#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
KDB_WIDE_MAX_B=0 python $R/scripts/wide_probe.py 2>&1 | grep -E "B=|KDB"
KDB_WIDE_MAX_B=512 python $R/scripts/wide_probe.py 2>&1 | grep -E "B=|KDB"
