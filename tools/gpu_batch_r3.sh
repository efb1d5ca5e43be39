set -u
mkdir -p gpurun_out/r3l
(timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3l/gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3l/gpu.log; grep -v "^Extension" gpurun_out/r3l/gpu.log | tail -3)
timeout -k 10 300 python tools/batch_order_probe.py --knobs "refill_overlap=0,1:refill_weight=8,12,16,24" > gpurun_out/r3l/knobs.txt 2>&1; grep -v amdgpu gpurun_out/r3l/knobs.txt | tail -10
timeout -k 10 200 python tests/fuzz_trace.py --iterations 200 --alpha 0.7 > gpurun_out/r3l/fuzz_alpha.log 2>&1; tail -2 gpurun_out/r3l/fuzz_alpha.log
