#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
python $R/scripts/flat_probe.py --bs 16,128,1024,8192 2>&1 | grep -E "B=|agree"
run() { rm -rf $R/gpurun_out/$1; timeout 300 rocprofv3 --pmc $2 --kernel-trace -d $R/gpurun_out/$1 -o p -- python $R/scripts/flat_probe.py --bs 8192 --reps 1 > /tmp/$1.log 2>&1; }
run fs1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS"
run fs2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE"
python3 - <<'PY'
import sqlite3,glob
for d in ("fs1","fs2"):
    for f in glob.glob(f"/root/repo/gpurun_out/{d}/*.db"):
        con=sqlite3.connect(f); cur=con.cursor()
        rows=cur.execute("select counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like '%flat_scan_kernel%' group by counter_name").fetchall()
        for r in rows: print(d, r[0], "%.4g"%r[1], "launches", r[2], "dur_us %.0f"%(r[3]/1e3))
PY
