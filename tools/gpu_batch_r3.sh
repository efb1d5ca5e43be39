set -u
mkdir -p gpurun_out/r3p
(timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r3p/gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3p/gpu.log; grep -v "^Extension" gpurun_out/r3p/gpu.log | tail -3)
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3p/smoke.log 2>&1; tail -1 gpurun_out/r3p/smoke.log
timeout -k 10 300 python bench.py > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err; tail -2 gpurun_out/r3p/bench.err; cut -c1-200 gpurun_out/r3p/bench.json
NNBVH_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --sample-sets 1 --no-cpu-baseline > gpurun_out/r3p/bench_2rank_gloo.json 2> gpurun_out/r3p/bench_2rank_gloo.err; tail -2 gpurun_out/r3p/bench_2rank_gloo.err; cut -c1-200 gpurun_out/r3p/bench_2rank_gloo.json
