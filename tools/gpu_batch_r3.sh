set -u
mkdir -p gpurun_out/r3n
(timeout -k 10 300 python -m pytest tests/test_kdtree.py -m gpu -x -q > gpurun_out/r3n/gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3n/gpu.log; grep -v "^Extension" gpurun_out/r3n/gpu.log | tail -3)
timeout -k 10 700 python tests/fuzz_trace.py --iterations 1000 --kd --two-level 3 --alpha 0.3 --seed 7 > gpurun_out/r3n/fuzz_all.log 2>&1; tail -2 gpurun_out/r3n/fuzz_all.log
timeout -k 10 200 python tools/fuzz_kd_build.py --iterations 600 --seed 5 > gpurun_out/r3n/fuzz_kd_build.log 2>&1; tail -2 gpurun_out/r3n/fuzz_kd_build.log
