set -u
mkdir -p gpurun_out/r3r
(timeout -k 10 300 python -m pytest tests/test_cpp_adapter.py -m gpu -x -q > gpurun_out/r3r/gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3r/gpu.log; grep -v "^Extension" gpurun_out/r3r/gpu.log | tail -8)
tests/cpp/adapter_check
