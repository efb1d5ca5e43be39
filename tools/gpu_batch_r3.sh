set -u
mkdir -p gpurun_out/r3o
(timeout -k 10 400 python -m pytest tests/test_bench_contract.py tests/test_raygen.py -m "gpu or not gpu" -x -q > gpurun_out/r3o/gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3o/gpu.log; grep -v "^Extension" gpurun_out/r3o/gpu.log | tail -3)
timeout -k 10 300 python bench.py > gpurun_out/r3o/bench.json 2> gpurun_out/r3o/bench.err; tail -3 gpurun_out/r3o/bench.err; cut -c1-200 gpurun_out/r3o/bench.json
tools/profile_bench.sh gpurun_out/r3o/final --sample-sets 2 > gpurun_out/r3o/final.log 2>&1
python3 tools/prof_summary.py gpurun_out/r3o/final gpurun_out/r3o/final/summary.json --meta workload=bench_default spp=8 > gpurun_out/r3o/final.summary.txt 2>&1
cp $(find gpurun_out/r3o/final/kt -name "*kernel_stats.csv" | head -1) gpurun_out/r3o/final/kernel_stats.csv
for d in rdreq write sq1 sq2 tcc fetch tcp kt; do rm -rf gpurun_out/r3o/final/$d; done
grep "trace_kernel<3" gpurun_out/r3o/final.summary.txt | cut -c1-500
timeout -k 10 200 python bench.py --scene killeroos --steps 5 --warmup 1 --sample-sets 2 > gpurun_out/r3o/bench_killeroos.json 2> gpurun_out/r3o/bench_killeroos.err; cut -c1-160 gpurun_out/r3o/bench_killeroos.json
timeout -k 10 200 python bench.py --scene bathroom --steps 5 --warmup 1 --sample-sets 2 > gpurun_out/r3o/bench_bathroom.json 2> gpurun_out/r3o/bench_bathroom.err; cut -c1-160 gpurun_out/r3o/bench_bathroom.json
