#!/bin/bash
# Latency regime under the tracer (run through gpurun): rocprofv3 --kernel-trace --stats of scripts/lat_probe.py at 1 / 64 / 1024
# queries per call; the summary (per kernel instance: launches, average / min / max duration) lands in gpurun_out/, copy it to
# profiles/.
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out
rm -rf $O/lat_$TAG
LAT_BS=1,64,1024 timeout 600 rocprofv3 --kernel-trace --stats -d $O/lat_$TAG -o lat -- python $R/scripts/lat_probe.py > $O/lat_$TAG.log 2>&1
python3 - <<PY > $O/${TAG}_latency_rocprofv3_summary.txt
import sqlite3, glob, re, subprocess
def demangle(n):
    try:
        return subprocess.run(["/usr/bin/c++filt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n
print("# rocprofv3 --kernel-trace --stats -- python scripts/lat_probe.py (LAT_BS=1,64,1024; 1M x 768 cosine, ef=60, k=10)")
print("# hnsw_search_kernel<PREC, METRIC, NCH, BS, VIS, WIDE>: WIDE = waves per query; launches grouped by grid size in threads (64 x WIDE per query: 256 = 1 query, 16384 = 64 queries, 262144 = 1024 queries)")
for f in glob.glob("$O/lat_$TAG/*.db") + glob.glob("$O/lat_$TAG/*/*.db"):
    con = sqlite3.connect(f); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table' or type='view'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = f"select s.kernel_name, d.grid_size_x, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id where s.kernel_name like '%hnsw_search_kernel%' group by s.kernel_name, d.grid_size_x order by d.grid_size_x"
    for r in cur.execute(q):
        m = re.search(r"hnsw_search_kernel<[^>]*>", demangle(r[0]))
        print("%-44s grid %7d threads launches %5d avg %8.1f us min %8.1f max %8.1f" % (m.group(0) if m else r[0][:44], r[1], r[2], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3))
print("# the probe's own lines (HIP events of the library / back-to-back calls):")
for l in open("$O/lat_$TAG.log"):
    if l.startswith("B="): print("# " + l.strip())
PY
cat $O/${TAG}_latency_rocprofv3_summary.txt
