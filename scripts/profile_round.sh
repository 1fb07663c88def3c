#!/bin/bash
# Round profile (run on the GPU box through gpurun; summaries land in gpurun_out/, copy them to profiles/):
#   1. rocprofv3 --kernel-trace --stats of the bench command's headline leg (--no-extras: the extra legs launch the same
#      kernel at other batch sizes, and the in-run counter passes are rocprofv3 runs themselves);
#   2. the plain bench command: its JSON line carries the HBM traffic (FETCH_SIZE / WRITE_SIZE passes of that very run) and
#      the matrix-core counters of the flat-scan leg;
#   3. counters of the big-tile flat-scan kernel (separate --pmc passes, no API tracing beside them).
TAG=${1:-r02a}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out
rm -rf $O/prof_$TAG $O/fsq_$TAG $O/fsf_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o bench -- python $R/bench.py --no-extras --no-cpu > $O/bench_prof_$TAG.json 2> $O/bench_prof_$TAG.log
timeout 1500 python $R/bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.log
cp $O/bench_extras.json $O/bench_${TAG}_extras.json
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --kernel-trace -d $O/fsq_$TAG -o p -- python $R/scripts/flat_probe.py --bs 8192 --reps 2 > /tmp/fsq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fsf_$TAG -o p -- python $R/scripts/flat_probe.py --bs 8192 --reps 2 > /tmp/fsf.log 2>&1
python3 $R/scripts/prof_summary.py $(ls $O/prof_$TAG/*.db $O/prof_$TAG/*/*.db 2>/dev/null | head -1) > $O/${TAG}_bench_rocprofv3_summary.txt
python3 - <<PY >> $O/${TAG}_bench_rocprofv3_summary.txt
import sqlite3, glob, re, json
print("# command under the tracer: rocprofv3 --kernel-trace --stats -- python bench.py --no-extras --no-cpu (the headline leg: ef sweep, 3 warm-up + 20 timed steps)")
try:
    d = json.loads(open("$O/bench_$TAG.json").read().strip().splitlines()[-1])
    print("# plain run of the same box (python bench.py): value %.0f %s, ms_per_step %.4f, roofline %s" % (d["value"], d["unit"], d["ms_per_step"], json.dumps(d["roofline"])))
    print("# legs (compact line):", json.dumps(d.get("legs")))
    full = json.loads(open("$O/bench_extras.json").read())   # the full record of the same run (bench.py emit)
    print("# flat-scan leg:", json.dumps(full.get("flat_scan_leg")))
except Exception as e:
    print("# bench json unreadable:", e)
for d in ("fsq_$TAG", "fsf_$TAG"):
    for f in glob.glob("$O/" + d + "/*.db") + glob.glob("$O/" + d + "/*/*.db"):
        cur = sqlite3.connect(f).cursor()
        print("# flat scan (8192 queries over 1M x 768, scripts/flat_probe.py) counters, pass", d)
        for r in cur.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like '%flat_scan_big%' group by kernel_name, counter_name"):
            m = re.search(r"(flat_\w+<[^>]*>|flat_\w+)", r[0])
            print("%-40s %-32s avg %.6g launches %d dur_us %.0f" % (m.group(1) if m else r[0][:40], r[1], r[2], r[3], r[4] / 1e3))
PY
tail -40 $O/${TAG}_bench_rocprofv3_summary.txt
