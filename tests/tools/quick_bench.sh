#!/bin/bash
# quick A/B on the GPU box: parity subset + 1M probe at the bench operating point
cd /root/repo
make -C oracle -s
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "search_bit_exact or golden or allow" 2>&1 | tail -2
timeout 600 python scripts/scale_probe.py --n 1000000 --dim 768 --nq 8192 --efs ${EFS:-64,64,200} 2>&1 | grep -E "build|ef"
