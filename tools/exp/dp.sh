mkdir -p gpurun_out/r2
python -m pytest tests/test_engine_gpu.py -q -x -s > gpurun_out/r2/t5.log 2>&1; echo "engine tests rc=$?"; tail -3 gpurun_out/r2/t5.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --global-batch 256 > gpurun_out/r2/dp_gloo.log 2>&1; echo "gloo 2-rank rc=$?"; tail -1 gpurun_out/r2/dp_gloo.log | cut -c1-700
