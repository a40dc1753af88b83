mkdir -p gpurun_out/r2/prof
python bench.py --roofline-only | tee gpurun_out/r2/roof_live.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2/prof/roof -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $GRAFT_REPO_ROOT/gpurun_out/r2/prof/roof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r2/prof/roof -name "*kernel_stats.csv" | head -1); echo $f; head -5 $f
tail -2 gpurun_out/r2/prof/roof.log | cut -c1-600
