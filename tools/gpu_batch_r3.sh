set -u
mkdir -p gpurun_out/r3m
(timeout -k 10 400 python -m pytest tests/test_kdtree.py tests/test_kd_build_gpu.py -m gpu -x -q > gpurun_out/r3m/gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3m/gpu.log; grep -v "^Extension" gpurun_out/r3m/gpu.log | tail -3)
timeout -k 10 300 python tests/fuzz_trace.py --iterations 150 --kd > gpurun_out/r3m/fuzz_kd.log 2>&1; tail -2 gpurun_out/r3m/fuzz_kd.log
timeout -k 10 300 python bench.py --tree kd --sample-sets 1 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3m/bench_kd.json 2> gpurun_out/r3m/bench_kd.err; tail -1 gpurun_out/r3m/bench_kd.err; python3 -c "import json; r=json.load(open('gpurun_out/r3m/bench_kd.json')); print(r['value'], r['ms_per_step'], r['per_class_mrays'])"
for s in bathroom killeroos; do timeout -k 10 200 python bench.py --tree kd --scene $s --sample-sets 1 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3m/bench_kd_$s.json 2> gpurun_out/r3m/bench_kd_$s.err; python3 -c "import json; r=json.load(open('gpurun_out/r3m/bench_kd_$s.json')); print('$s', r['value'], r['ms_per_step'], r['per_class_mrays'])"; done
