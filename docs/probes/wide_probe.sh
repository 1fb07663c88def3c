#!/bin/bash
# latency mode of the graph search on one box: one wave per query / four waves / four waves with speculative row fetch
cd /tmp && export TMPDIR=/tmp
R=/root/repo
KDB_WIDE_MAX_B=0 python $R/scripts/wide_probe.py 2>&1 | grep -E "B=|KDB|sig"
KDB_WIDE_SPEC_MAX_B=0 python $R/scripts/wide_probe.py 2>&1 | grep -E "B=|sig"
python $R/scripts/wide_probe.py 2>&1 | grep -E "B=|sig"
KDB_WIDE_SPEC_MAX_B=512 python $R/scripts/wide_probe.py 2>&1 | grep -E "B=|sig"
cd $R && timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
