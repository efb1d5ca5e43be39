#!/bin/bash
# Scheduling-knob sweep of the kd-tree kernels on the bathroom kd-tree (run on the GPU box from the repo root).
# Needs the tuning build:  python -c "from nn_bvh_amd import build as nb; nb.build_variant('kdtune', ['NNBVH_KD_TUNE'])"
# (NNBVH_KD_TUNE makes kd_launch read NNBVH_KD_PRIMW / NNBVH_KD_NODEREP / NNBVH_KD_REFILLW from the environment).
for pw in 12 24 48 96; do for nr in 2 4 8; do for rw in 8 16; do
NNBVH_LIB=libnnbvh_hip_kdtune.so NNBVH_KD_PRIMW=$pw NNBVH_KD_NODEREP=$nr NNBVH_KD_REFILLW=$rw timeout -k 10 200 python bench.py --tree kd --scene bathroom --no-cpu-baseline --steps 5 --warmup 1 --sample-sets 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pw $pw nr $nr rw $rw', d['value'], d['per_class_mrays'])"
done; done; done
