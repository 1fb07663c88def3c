#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
run() { rm -rf $R/gpurun_out/$1; timeout 300 rocprofv3 --pmc $2 --kernel-trace -d $R/gpurun_out/$1 -o p -- python $R/scripts/scale_probe.py --n 1000000 --dim 768 --nq 8192 --efs 64 > /tmp/$1.log 2>&1; tail -1 /tmp/$1.log | cut -c1-150; }
run tcc1 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
run tcc2 "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_REQ_sum"
run tcc3 "TCC_EA0_ATOMIC_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RD_UNCACHED_32B_sum"
python3 - <<'PY'
import sqlite3,glob
for d in ("tcc1","tcc2","tcc3"):
    for f in glob.glob(f"/root/repo/gpurun_out/{d}/*.db"):
        con=sqlite3.connect(f); cur=con.cursor()
        for pat in ("hnsw_search_kernel","flat_scan_kernel"):
            rows=cur.execute("select counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like ? group by counter_name",(f"%{pat}%",)).fetchall()
            for r in rows: print(d, pat, r[0], "%.4g"%r[1], "launches", r[2], "dur_us %.0f"%(r[3]/1e3))
PY
