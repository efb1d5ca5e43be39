set -u
mkdir -p gpurun_out/r3q
(timeout -k 10 400 python -m pytest tests/test_bench_contract.py -m gpu -x -q > gpurun_out/r3q/gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3q/gpu.log; grep -v "^Extension" gpurun_out/r3q/gpu.log | tail -3)
NNBVH_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --sample-sets 1 --no-cpu-baseline > gpurun_out/r3q/bench_2rank_gloo.json 2> gpurun_out/r3q/bench_2rank_gloo.err; wc -l gpurun_out/r3q/bench_2rank_gloo.json; cut -c1-120 gpurun_out/r3q/bench_2rank_gloo.json
