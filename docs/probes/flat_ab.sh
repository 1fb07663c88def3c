#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by +-10 %): interleaved runs of scripts/flat_probe.py
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for i in 1 2 3; do
  for lib in libkektor_hip_prev.so libkektor_hip.so; do
    echo -n "$lib  "; KEKTOR_HIP_LIB=$R/kektordb_amd/lib/$lib python $R/scripts/flat_probe.py --bs ${BS:-8192} --reps 5 2>&1 | grep "B=" | tr '\n' ' '; echo
  done
done
