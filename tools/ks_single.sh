#!/bin/bash
# kernel stats of the single-stream step for the library named by $VITGAN_HIP_LIB (default: the product library)
#   bash tools/ks_single.sh <tag>   -> gpurun_out/<tag>/ks.csv
set -u
TAG=${1:-ks}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-workloads --no-roofline --single-stream > $OUT/log.txt 2>&1
f=$(find $OUT/run -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/ks.csv
rm -rf $OUT/run
