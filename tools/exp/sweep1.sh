set -x
mkdir -p gpurun_out/r2
export VITGAN_HIP_LIB=$PWD/vit-gan_amd/libvitgan_hip_tune.so
VG_GEMM_TPW=3 python -m pytest tests/test_ops_gpu.py tests/test_net_gpu.py tests/test_blocks_gpu.py -q -x > gpurun_out/r2/tpw3_tests.log 2>&1; echo "tests tpw3 rc=$?"
VG_GEMM_TPW=2 VG_GEMM_WM=2 python -m pytest tests/test_ops_gpu.py tests/test_blocks_gpu.py -q -x > gpurun_out/r2/tpw2_tests.log 2>&1; echo "tests tpw2 wm2 rc=$?"
unset VITGAN_HIP_LIB
python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -q -x > gpurun_out/r2/prod_tests.log 2>&1; echo "tests product rc=$?"
python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PRODUCT auto', d['ms_per_step'], d['value'])"
export VITGAN_HIP_LIB=$PWD/vit-gan_amd/libvitgan_hip_tune.so
for wm in 0 2; do for tpw in 1 2 3 4; do
  VG_GEMM_TPW=$tpw VG_GEMM_WM=$wm python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WM $wm TPW $tpw', d['ms_per_step'], d['value'])"
done; done
