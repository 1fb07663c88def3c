#!/bin/bash
# SQ counters of hnsw_search_kernel at the bench operating point
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/pmc_sq
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace -d $R/gpurun_out/pmc_sq -o p -- python $R/scripts/scale_probe.py --n 1000000 --dim 768 --nq 8192 --efs 64 > /tmp/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/gpurun_out/pmc_sq2 -o p -- python $R/scripts/scale_probe.py --n 1000000 --dim 768 --nq 8192 --efs 64 > /tmp/p2.log 2>&1
tail -3 /tmp/p1.log | cut -c1-200
python3 - <<'PY'
import sqlite3,glob
for d in ("pmc_sq","pmc_sq2"):
    for f in glob.glob(f"/root/repo/gpurun_out/{d}/*.db"):
        con=sqlite3.connect(f); cur=con.cursor()
        rows=cur.execute("select counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like '%hnsw_search_kernel%' group by counter_name").fetchall()
        for r in rows: print(d, r[0], "%.4g"%r[1], "launches", r[2], "dur_us %.0f"%(r[3]/1e3))
PY
