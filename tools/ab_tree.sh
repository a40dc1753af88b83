#!/bin/bash
# A/B of whole-step time between two source TREES on ONE box (a tree = a checkout with its own built library; the ABI
# version differs between rounds, so the library alone cannot be swapped): bash tools/ab_tree.sh ab_r02 . [bench flags]
# Alternating runs, three rounds; prints ms_per_step of each run.
R=${GRAFT_REPO_ROOT:-$PWD}
A=$R/$1; B=$R/$2; shift 2
for round in 1 2 3; do
  for T in $A $B; do
    X=""; grep -q no-extra-workloads $T/bench.py && X=--no-extra-workloads   # (trees before round 4 do not know the flag)
    ms=$(cd $T && python3 bench.py --no-roofline --no-cpu-baseline $X --steps 40 --warmup 10 "$@" 2>/dev/null | python3 -c 'import sys,json; print(json.loads(sys.stdin.readlines()[-1])["ms_per_step"])')
    echo "$(basename $T) $ms"
  done
done
