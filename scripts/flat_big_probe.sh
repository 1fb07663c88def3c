#!/bin/bash
# Ablation + counters of the big-tile ranking kernel (flat_scan_big_kernel), 8192 queries over 1M x 768 f32 cosine.
# KDB_FB_DBG: 1 no selection, 2 no DMA after the first slab, 4 no MFMAs (timings only: the answers are wrong).
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; TAG=${1:-fb}
for d in 0 1 2 3 4 5 6; do echo "dbg=$d"; KDB_FB_DBG=$d python $R/scripts/flat_probe.py --bs ${BS:-8192} --reps 3 2>&1 | grep "B="; done
rm -rf $O/fbp_$TAG $O/fbf_$TAG
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --kernel-trace -d $O/fbp_$TAG -o p -- python $R/scripts/flat_probe.py --bs ${BS:-8192} --reps 1 > /tmp/fbp.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fbf_$TAG -o p -- python $R/scripts/flat_probe.py --bs ${BS:-8192} --reps 1 > /tmp/fbf.log 2>&1
python3 - <<PY
import sqlite3, glob, re
for d in ("fbp_$TAG", "fbf_$TAG"):
    for f in glob.glob("$O/" + d + "/*.db"):
        cur = sqlite3.connect(f).cursor()
        print("# counters, pass", d)
        for r in cur.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like '%flat_scan%' group by kernel_name, counter_name"):
            m = re.search(r"(flat_\w+<[^>]*>|flat_\w+)", r[0])
            print("%-40s %-32s avg %.5g launches %d dur_us %.0f" % (m.group(1) if m else r[0][:40], r[1], r[2], r[3], r[4] / 1e3))
PY
