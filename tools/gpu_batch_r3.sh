set -u
mkdir -p gpurun_out/r3s
for v in "" _prio1 _prio2 _prio3 ""; do
  NNBVH_LIB=libnnbvh_hip$v.so timeout -k 10 150 python tools/batch_order_probe.py > gpurun_out/r3s/prio$v.txt 2>&1; echo "== lib$v"; grep "S B2 B1 P\|P B1 B2 S" gpurun_out/r3s/prio$v.txt
done
