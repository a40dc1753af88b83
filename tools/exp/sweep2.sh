mkdir -p gpurun_out/r2
run() { python bench.py --steps 30 --warmup 10 --no-cpu-baseline "$@" 2>gpurun_out/r2/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['value'], d['config']['last_losses'])" || tail -5 gpurun_out/r2/err.log; }
run
run --two-stream 1
run --two-stream 1 --graph 0
run --no-fuse
run --graph 0
