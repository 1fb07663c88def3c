#!/bin/bash
# instruction mix / stall counters of hnsw_search_kernel (f32 and int8 indexes over the same graph)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
run() { rm -rf $R/gpurun_out/$1; timeout 400 rocprofv3 --pmc $2 --kernel-trace -d $R/gpurun_out/$1 -o p -- python $R/scripts/quant_probe.py --efs 64 > /tmp/$1.log 2>&1; }
run ps1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
run ps2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
python3 - <<'PY'
import sqlite3,glob
for d in ("ps1","ps2"):
    for f in glob.glob(f"/root/repo/gpurun_out/{d}/*.db"):
        cur=sqlite3.connect(f).cursor()
        rows=cur.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like '%hnsw_search_kernel%' group by kernel_name, counter_name").fetchall()
        for r in rows:
            kn = "i8" if "ILi2E" in r[0] or "<2," in r[0] else "f32"
            print(d, kn, r[1], "%.4g"%r[2], "launches", r[3], "dur_us %.0f"%(r[4]/1e3))
PY
