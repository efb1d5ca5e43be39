#!/bin/bash
# A/B of the fat-record experiment (NNBVH_FAT builds + NNBVH_LAYOUT=32) against the product on the bench step.
# Each leg is one bench.py run (its cpu_baseline leg checks the GPU records against the oracle: matches_gpu).
set -e
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 3 > gpurun_out/fat_base.json 2> gpurun_out/fat_base.err
for v in "$@"; do
  NNBVH_LIB=libnnbvh_hip_$v.so NNBVH_LAYOUT=32 python bench.py --steps 20 --warmup 3 > gpurun_out/fat_$v.json 2> gpurun_out/fat_$v.err
done
python bench.py --steps 20 --warmup 3 > gpurun_out/fat_base2.json 2> gpurun_out/fat_base2.err
python - "$@" <<'PY'
import json, sys
for v in ["base"] + sys.argv[1:] + ["base2"]:
    d = json.load(open(f"gpurun_out/fat_{v}.json"))
    print(v, d["value"], "Mray/s", d["ms_per_step"], "ms", d["per_class_mrays"], "matches", d["cpu_baseline"].get("matches_gpu"),
          "dep", d.get("dependent_step", {}).get("value"))
PY
