#!/bin/bash
# Produces the rocprofv3 evidence committed under profiles/ (run on the GPU box from the repo root):
#   bash tools/profile_round.sh r04 $(git rev-parse --short HEAD)     (the commit id is composed where .git exists: the GPU box has none)
# kernel stats of the default (single-stream) step and of the step with the weight gradients on a side stream, the roofline leg, PMC traffic of the GEMM shapes and the
# whole-step MFMA-utilisation / HBM-traffic summary.  PMC passes use --kernel-trace only (no other trace domains).
set -u
TAG=${1:-r04}
export VG_COMMIT=${2:-}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() { d=$1; shift; rocprofv3 "$@" > $OUT/$d.log 2>&1; }
prof ks_default --kernel-trace --stats --output-format csv -d $OUT/ks_default -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-workloads --no-roofline
prof ks_roof    --kernel-trace --stats --output-format csv -d $OUT/ks_roof    -- python3 $R/bench.py --roofline-only
export REPS=3
prof gf --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/gemm_fetch -- python3 $R/tools/gemm_bench.py
prof gw --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/gemm_write -- python3 $R/tools/gemm_bench.py
unset REPS
prof sm --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/step_mfma -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra-workloads --no-roofline
prof sf --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/step_fetch -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra-workloads --no-roofline
prof sw --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/step_write -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra-workloads --no-roofline
cd $R
python3 tools/pmc_traffic.py $OUT/gemm_fetch $OUT/gemm_write $OUT/${TAG}_gemm_pmc_traffic.json > $OUT/traffic.txt 2>&1
python3 tools/pmc_summary.py $OUT/step_mfma $OUT/step_fetch $OUT/step_write $OUT/${TAG}_step_pmc_summary.json > $OUT/summary.txt 2>&1
for n in default roof; do f=$(find $OUT/ks_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_ks_$n.csv; done
python3 tools/gemm_bench.py > $OUT/${TAG}_gemm_bench.txt 2>&1
python3 tools/row_bench.py > $OUT/${TAG}_row_bench.txt 2>&1
# one step as a launch-by-launch timeline (from the kernel trace of the first run), the step with the gradient penalty, the vendor-GEMM calibration,
# the stage-by-stage parity table and the bench line itself
python3 tools/step_timeline.py $OUT/ks_default > $OUT/${TAG}_step_timeline.txt 2>&1
(cd /tmp && prof ks_gp --kernel-trace --stats --output-format csv -d $OUT/ks_gp -- python3 $R/bench.py --loss wasserstein --gp 10 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-workloads --no-roofline)
f=$(find $OUT/ks_gp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_gp_step_kernel_stats.csv
python3 tools/step_timeline.py $OUT/ks_gp > $OUT/${TAG}_gp_step_timeline.txt 2>&1
(python3 tools/micro/blas_cmp.py; python3 tools/gemm_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_vendor_gemm_calibration_raw.txt
python3 -m pytest tests/test_blocks_gpu.py -m gpu -q -s 2>&1 | grep -E "tensors checked|SLN scalars|^ +(sln|transformer_layers|layer_norm)" > $OUT/stage_raw.txt
# the bench line quotes the two PMC summaries from profiles/ (when their tree hash is this tree's): put this run's there first
cp $OUT/${TAG}_gemm_pmc_traffic.json $OUT/${TAG}_step_pmc_summary.json $R/profiles/ 2>/dev/null
python3 bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line.json
tail -4 $OUT/traffic.txt; tail -12 $OUT/summary.txt; grep -v amdgpu.ids $OUT/${TAG}_gemm_bench.txt; cut -c1-400 $OUT/${TAG}_bench_line.json
