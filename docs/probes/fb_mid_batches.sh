cd /root/repo
for v in A=1 KDB_FB_NOSEED=1; do echo "[$v]"; env $v python scripts/flat_probe.py --bs 128,256,1024,2048 --reps 5 2>&1 | grep "B="; done
echo "config-3 shape (10M x 768 L2 k=100, 1024 queries):"
for v in A=1 KDB_FB_NOSEED=1; do echo "[$v] $(env $v python scripts/flat_probe.py --n 10000000 --metric 0 --k 100 --bs 1024 --reps 3 2>&1 | grep 'B=')"; done
