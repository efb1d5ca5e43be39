set -u
mkdir -p gpurun_out/r3i
timeout -k 10 200 python tests/fuzz_trace.py --iterations 150 --alpha 0.5 > gpurun_out/r3i/fuzz_alpha.log 2>&1; tail -2 gpurun_out/r3i/fuzz_alpha.log
tools/profile_bench.sh gpurun_out/r3i/final --sample-sets 2 > gpurun_out/r3i/final.log 2>&1
python3 tools/prof_summary.py gpurun_out/r3i/final gpurun_out/r3i/final/summary.json --meta workload=bench_default spp=8 > gpurun_out/r3i/final.summary.txt 2>&1
cp $(find gpurun_out/r3i/final/kt -name "*kernel_stats.csv" | head -1) gpurun_out/r3i/final/kernel_stats.csv
for d in rdreq write sq1 sq2 tcc fetch tcp kt; do rm -rf gpurun_out/r3i/final/$d; done
grep "trace_kernel<3\|wf_record\|film_add" gpurun_out/r3i/final.summary.txt | cut -c1-500
for c in primary bounce bounce2 shadow; do
  tools/profile_bench.sh gpurun_out/r3i/$c --passes "sq1 tcc rdreq write" --script tools/profile_workloads.py crown_$c > gpurun_out/r3i/$c.log 2>&1
  python3 tools/prof_summary.py gpurun_out/r3i/$c gpurun_out/r3i/$c/summary.json --filter "trace_kernel<3" --meta workload=crown_$c spp=8 > gpurun_out/r3i/$c.summary.txt 2>&1
  cp $(find gpurun_out/r3i/$c/kt -name "*kernel_stats.csv" | head -1) gpurun_out/r3i/$c/kernel_stats.csv
  for d in rdreq write sq1 sq2 tcc fetch tcp kt; do rm -rf gpurun_out/r3i/$c/$d; done
  cat gpurun_out/r3i/$c.summary.txt | cut -c1-500
done
timeout -k 10 300 python bench.py > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err; tail -2 gpurun_out/r3i/bench.err; cut -c1-200 gpurun_out/r3i/bench.json
du -sh gpurun_out
