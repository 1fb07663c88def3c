#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
for i in 1 2 3 4; do timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_tied_walks_with_deleted" 2>&1 | grep -E "passed|failed|Error:" | tail -3; done
echo "== give up"; KDB_HEAP_OVERLAP_GIVE_UP=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_tied_walks_with_deleted or duplicate_vectors" 2>&1 | grep -E "passed|failed|Error:" | tail -3
echo "== script PREC=0"; for i in 1 2 3; do PREC=0 REPS=100,200,400,400,800,800,1600,400,800 timeout 200 python scripts/dbg/overlap_chunks.py 2>&1 | grep -v amdgpu.ids | grep -v "bad ids 0 .* bad cnt 0 " | tail -4; done
