mkdir -p gpurun_out/r2
python -m pytest tests/test_ops_gpu.py -q -s -k "attention" > gpurun_out/r2/t6.log 2>&1; echo "attention ops rc=$?"; grep -n "fp8 attention\|Error\|passed\|failed" gpurun_out/r2/t6.log | head -20
python -m pytest tests/test_net_gpu.py -q -s -k "fp8" > gpurun_out/r2/t7.log 2>&1; echo "fp8 net rc=$?"; tail -3 gpurun_out/r2/t7.log
for f in 0 1; do python bench.py --workload c5 --steps 10 --warmup 3 --no-cpu-baseline --fp8-attention $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 fp8=$f', d['ms_per_step'], d['value'], d['config']['last_losses'])"; done
