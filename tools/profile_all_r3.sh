set -u
mkdir -p gpurun_out/r3f
(timeout -k 10 300 python -m pytest tests/test_interaction.py tests/test_gpu_parity.py tests/test_alpha.py -m gpu -x -q > gpurun_out/r3f/intr_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3f/intr_tests.log; grep -v "^Extension" gpurun_out/r3f/intr_tests.log | tail -4)
timeout -k 10 120 python tools/overlap_copy_probe.py > gpurun_out/r3f/overlap_copy.txt 2>&1; grep -v amdgpu gpurun_out/r3f/overlap_copy.txt
HSA_ENABLE_SDMA=0 timeout -k 10 120 python tools/overlap_copy_probe.py > gpurun_out/r3f/overlap_copy_nosdma.txt 2>&1; grep -v amdgpu gpurun_out/r3f/overlap_copy_nosdma.txt
timeout -k 10 200 python tools/host_buffer_probe.py > gpurun_out/r3f/host_buffer.txt 2>&1; grep -v amdgpu gpurun_out/r3f/host_buffer.txt
timeout -k 10 200 python tools/batch_order_probe.py > gpurun_out/r3f/batch_order.txt 2>&1; grep -v amdgpu gpurun_out/r3f/batch_order.txt
for w in patches alpha inst anim tr intr; do
  tools/profile_bench.sh gpurun_out/r3f/$w --light --script tools/profile_workloads.py $w > gpurun_out/r3f/$w.log 2>&1
  python3 tools/prof_summary.py gpurun_out/r3f/$w gpurun_out/r3f/$w/summary.json --meta workload=$w > gpurun_out/r3f/$w.summary.txt 2>&1
  rm -rf gpurun_out/r3f/$w/rdreq gpurun_out/r3f/$w/write gpurun_out/r3f/$w/sq1 gpurun_out/r3f/$w/sq2 gpurun_out/r3f/$w/tcc
done
tools/profile_bench.sh gpurun_out/r3f/kd_crown --light --tree kd --sample-sets 1 > gpurun_out/r3f/kd_crown.log 2>&1
python3 tools/prof_summary.py gpurun_out/r3f/kd_crown gpurun_out/r3f/kd_crown/summary.json --filter "kd_trace|wf_|film_" --meta workload=kd_crown spp=8 > gpurun_out/r3f/kd_crown.summary.txt 2>&1
rm -rf gpurun_out/r3f/kd_crown/rdreq gpurun_out/r3f/kd_crown/write gpurun_out/r3f/kd_crown/sq1 gpurun_out/r3f/kd_crown/sq2 gpurun_out/r3f/kd_crown/tcc
du -sh gpurun_out
