#!/bin/bash
# A/B of the big-tile exact scan: 8192 queries x 1M x 768 cosine, alternating variants (env switches given as arguments, "-" = none)
cd /root/repo
for i in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e="A=1"; else e="$v"; fi
    echo "variant [$v]: $(env $e python scripts/flat_probe.py --bs 8192 --reps 5 2>&1 | grep 'B=')"
  done
done
