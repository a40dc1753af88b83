#!/bin/bash
# A/B of whole-step time between library builds on ONE box: bash tools/ab_step.sh libA.so libB.so [bench flags]
# (alternating runs, two rounds; prints ms_per_step of each run)
R=${GRAFT_REPO_ROOT:-$PWD}
A=$R/$1; B=$R/$2; shift 2
for round in 1 2; do
  for L in $A $B; do
    for mode in "--concurrent-wgrad 1" ""; do
      ms=$(VITGAN_HIP_LIB=$L python3 $R/bench.py --no-roofline --no-cpu-baseline --no-extra-workloads --steps 40 --warmup 10 $mode "$@" 2>/dev/null | python3 -c 'import sys,json; print(json.loads(sys.stdin.readlines()[-1])["ms_per_step"])')
      echo "$(basename $L) ${mode:-single-stream(default)} $ms"
    done
  done
done
