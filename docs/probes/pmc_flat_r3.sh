#!/bin/bash
# counters of the big-tile flat-scan kernel (8192 queries over 1M x 768): which unit is the busy one.  Separate --pmc passes,
# --kernel-trace only beside them.  Summaries: gpurun_out/pmc_flat_r3.txt
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out
rm -rf $O/pf_a $O/pf_b $O/pf_c
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace -d $O/pf_a -o p -- python $R/scripts/flat_probe.py --bs 8192 --reps 2 > /tmp/pfa.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/pf_b -o p -- python $R/scripts/flat_probe.py --bs 8192 --reps 2 > /tmp/pfb.log 2>&1
timeout 300 rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_max GRBM_GUI_ACTIVE --kernel-trace -d $O/pf_c -o p -- python $R/scripts/flat_probe.py --bs 8192 --reps 2 > /tmp/pfc.log 2>&1
python3 - <<PY > $O/pmc_flat_r3.txt
import sqlite3, glob, re
print("# flat_scan_big_kernel, 8192 queries over 1M x 768 cosine (scripts/flat_probe.py): counters per launch (rocprofv3 --pmc, three passes)")
for d in ("pf_a", "pf_b", "pf_c"):
    for f in glob.glob("$O/" + d + "/*.db") + glob.glob("$O/" + d + "/*/*.db"):
        cur = sqlite3.connect(f).cursor()
        for r in cur.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like '%flat_scan_big%' group by kernel_name, counter_name"):
            print("%-36s avg %.6g launches %d dur_us %.0f" % (r[1], r[2], r[3], r[4] / 1e3))
PY
cat $O/pmc_flat_r3.txt; tail -2 /tmp/pfc.log
