#!/bin/bash
# round 6: does the lockstep of the workgroups that share a row stripe (KDB_FB_DRIFT) change what the big-tile scan fetches from HBM?
# FETCH_SIZE of flat_scan_big_kernel, 8192 queries over 1M x 768, free-running (0) against drift 1 / 2.  --pmc beside --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out
for d in 0 1 2; do
  rm -rf $O/pls_$d
  KDB_FB_DRIFT=$d timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pls_$d -o p -- python $R/scripts/flat_probe.py --bs 8192 --reps 3 > /tmp/pls_$d.log 2>&1
done
python3 - <<PY > $O/pmc_flat_lockstep.txt
import sqlite3, glob
print("# flat_scan_big_kernel, 8192 queries over 1M x 768 cosine (scripts/flat_probe.py): FETCH_SIZE per launch (KB; x2 = bytes on gfx950 for 16-byte-per-lane reads), by KDB_FB_DRIFT")
for d in (0, 1, 2):
    for f in glob.glob("$O/pls_%d/*.db" % d) + glob.glob("$O/pls_%d/*/*.db" % d):
        cur = sqlite3.connect(f).cursor()
        for r in cur.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like '%flat_scan_big%' group by kernel_name, counter_name"):
            print("drift %d  %-60s %-12s avg %.6g KB -> %.2f GB  launches %d dur_us %.0f" % (d, r[0][:60], r[1], r[2], 2 * r[2] * 1024 / 1e9, r[3], r[4] / 1e3))
PY
cat $O/pmc_flat_lockstep.txt
