mkdir -p gpurun_out/r2
timeout -k 10 500 python -m pytest tests/test_gp_gpu.py -q -s -x > gpurun_out/r2/t8.log 2>&1; echo "gp tests rc=$?"; grep -n "penalty:\|gp:\|largest\|Error\|passed\|failed" gpurun_out/r2/t8.log | cut -c1-400 | head -30
