#!/bin/bash
# A/B of two builds of the library on one box: 8192 queries x 1M x 768 cosine exact scan.  usage: fb_ab_lib.sh <libA> <libB>
cd /root/repo
for i in 1 2 3; do
  for l in "$@"; do
    echo "lib [$l]: $(KEKTOR_HIP_LIB=/root/repo/kektordb_amd/lib/$l python scripts/flat_probe.py --bs 8192 --reps 5 2>&1 | grep 'B=')"
  done
done
