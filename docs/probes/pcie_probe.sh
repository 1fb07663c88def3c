#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
export GPU_MAX_HW_QUEUES=8
KDB_HOST_CHUNK_MIN=0 python $R/scripts/pcie_probe.py 2>&1 | grep -E "B=|KDB|sig"
python $R/scripts/pcie_probe.py 2>&1 | grep -E "B=|KDB|sig"
