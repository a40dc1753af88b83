#!/bin/bash
# One rank's share of BASELINE.json configs[2] (global batch 2048 over 2 / 4 / 8 GPUs = 1024 / 512 / 256 images per GPU)
# timed on ONE GPU: what a rank computes per step at those batch sizes, without the exchange.  bash tools/c3_single_rank.sh
for b in 256 512 1024; do
  python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('per-GPU batch $b:', d['ms_per_step'], 'ms/step', d['value'], 'img/s', d['roofline']['step_frac_of_peak'], 'of bf16 peak')"
done
