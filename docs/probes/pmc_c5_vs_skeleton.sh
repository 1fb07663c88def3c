#!/bin/bash
# PMC comparison: the grouped scan of bench.py --inner-c5 vs the skeleton of scripts/micro/gather_patterns
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_real_$i -o r --output-format csv -- python $R/bench.py --inner-c5 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_micro_$i -o m --output-format csv -- $R/scripts/micro/bin/gather_patterns 10000000 1 104 > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for tag, pat in (("real", "small_kernel<1, 3, 8>"), ("micro", "scan_kernel<0, 4, 0>")):
    agg = collections.OrderedDict()
    for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/*counter_collection.csv")):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in per.items():
            # several rows per dispatch (per dimension) are summed per dispatch by rocprof already? take mean over rows x rows-per-dispatch
            agg[k] = (sum(v), len(v))
    print(tag, pat)
    for k, (s, n) in agg.items():
        print(f"   {k:<36} total {s:.4g} over {n} rows")
PY
