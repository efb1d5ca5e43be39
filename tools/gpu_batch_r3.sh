set -u
mkdir -p gpurun_out/r3j
(timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r3j/gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3j/gpu.log; grep -v "^Extension" gpurun_out/r3j/gpu.log | tail -3)
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3j/smoke.log 2>&1; tail -1 gpurun_out/r3j/smoke.log
timeout -k 10 300 python bench.py > gpurun_out/r3j/bench.json 2> gpurun_out/r3j/bench.err; tail -2 gpurun_out/r3j/bench.err; cut -c1-200 gpurun_out/r3j/bench.json
timeout -k 10 240 python tools/batch_order_probe.py --knobs > gpurun_out/r3j/knobs.txt 2>&1; grep -v amdgpu gpurun_out/r3j/knobs.txt | tail -32
