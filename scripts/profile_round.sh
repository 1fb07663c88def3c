#!/bin/bash
# Round profile: rocprofv3 kernel trace of the bench command, FETCH_SIZE / WRITE_SIZE passes (separate runs, no
# API tracing beside --pmc), MFMA / LDS counters of the flat scan.  Summaries go to gpurun_out/; copy to profiles/.
TAG=${1:-r01c}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out
rm -rf $O/prof_$TAG $O/pmc_fetch_$TAG $O/pmc_write_$TAG $O/fs1_$TAG $O/fs2_$TAG $O/fs3_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o bench -- python $R/bench.py --steps 20 --no-cpu > $O/bench_prof_$TAG.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_$TAG -o bench -- python $R/bench.py --steps 10 --no-cpu > /tmp/pf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_$TAG -o bench -- python $R/bench.py --steps 10 --no-cpu > /tmp/pw.log 2>&1
# the exact f32 tile kernel (f32 MFMA) on its own, then the default path (f16-ranked + exact settle)
KDB_FLAT_EXACT_ONLY=1 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace -d $O/fs1_$TAG -o p -- python $R/scripts/flat_probe.py --bs 8192 --reps 1 > /tmp/fs1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/fs3_$TAG -o p -- python $R/scripts/flat_probe.py --bs 8192 --reps 2 > /tmp/fs3.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fs2_$TAG -o p -- python $R/scripts/flat_probe.py --bs 1 --reps 3 > /tmp/fs2.log 2>&1
python3 $R/scripts/prof_summary.py $(ls $O/prof_$TAG/*.db | head -1) $(ls $O/pmc_fetch_$TAG/*.db | head -1) $(ls $O/pmc_write_$TAG/*.db | head -1) > $O/${TAG}_bench_rocprofv3_summary.txt
python3 - <<PY >> $O/${TAG}_bench_rocprofv3_summary.txt
import sqlite3, glob
for f in glob.glob("$O/fs3_$TAG/*.db"):
    cur = sqlite3.connect(f).cursor()
    print("# default flat scan of 8192 queries (f16-ranked + exact settle), kernel trace")
    for r in cur.execute("select name, count(*), avg(duration) from kernels where name like '%flat%' or name like '%gather_queries%' group by name"):
        print("%-90s calls %d avg_us %.1f" % (r[0][:90], r[1], r[2] / 1e3))
for d in ("fs1_$TAG", "fs2_$TAG"):
    for f in glob.glob("$O/" + d + "/*.db"):
        cur = sqlite3.connect(f).cursor()
        print("# flat scan counters, pass", d)
        for r in cur.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like '%flat_scan%' group by kernel_name, counter_name"):
            import re
            m = re.search(r"(flat_\w+<[^>]*>|flat_\w+)", r[0])
            print("%-40s %-32s avg %.5g launches %d dur_us %.0f" % (m.group(1) if m else r[0][:40], r[1], r[2], r[3], r[4] / 1e3))
PY
grep "^{" $O/bench_prof_$TAG.log | tail -1 > $O/bench_$TAG.json
tail -30 $O/${TAG}_bench_rocprofv3_summary.txt
