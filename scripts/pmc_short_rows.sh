#!/bin/bash
# round 6: what saturates the short-row walk at 16 walks per CU?  Instruction counts and busy cycles of hnsw_search_kernel on the
# GloVe-100-shaped corpus (400k x 100 cosine, efS 100, 8192 queries).  --pmc beside --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out
rm -rf $O/psr_a $O/psr_b
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --kernel-trace -d $O/psr_a -o p -- python $R/scripts/r5_shape_probe.py 100 400000 1 100 8192 > /tmp/psr_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/psr_b -o p -- python $R/scripts/r5_shape_probe.py 100 400000 1 100 8192 > /tmp/psr_b.log 2>&1
python3 - <<PY > $O/pmc_short_rows.txt
import sqlite3, glob
print("# hnsw_search_kernel, 400k x 100 cosine, 8192 queries (scripts/r5_shape_probe.py runs ef 20 and ef 100: rows grouped by duration), counters per launch")
for d in ("psr_a", "psr_b"):
    for f in glob.glob("$O/" + d + "/*.db") + glob.glob("$O/" + d + "/*/*.db"):
        cur = sqlite3.connect(f).cursor()
        for r in cur.execute("select counter_name, round(duration/100000.0), avg(value), count(*), avg(duration) from counters_collection where kernel_name like '%hnsw_search_kernel%' group by counter_name, round(duration/100000.0) order by counter_name, avg(duration)"):
            print("%-28s launches %2d dur_us %7.0f  avg %.6g" % (r[0], r[3], r[4] / 1e3, r[2]))
PY
cat $O/pmc_short_rows.txt
