#!/bin/bash
# A/B of library builds on the bench step, legs interleaved on one box: tools/ab_bench.sh <variant> [rounds]
# (variant = nn_bvh_amd/libnnbvh_hip_<variant>.so, tools: python -m nn_bvh_amd.build <variant> DEFINE...)
v=$1; rounds=${2:-3}
for r in $(seq $rounds); do
  for lib in libnnbvh_hip.so libnnbvh_hip_$v.so; do
    NNBVH_LIB=$lib python bench.py --steps 20 --warmup 3 --no-wavefront --no-order-probe --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$lib', d['value'], d['ms_per_step'])"
  done
done
