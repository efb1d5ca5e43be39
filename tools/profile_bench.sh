#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the bench command (per-kernel average duration)
#   2. separate --pmc passes for the HBM-side traffic of the traversal kernels
#      (TCC_EA0 read requests by size, FETCH_SIZE, WRITE_SIZE) — never combined with traces
# usage: tools/profile_bench.sh <outdir> [bench.py args...]
set -u
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
BENCH=(python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@")
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- "${BENCH[@]}" > "$out/kt_bench.json" 2> "$out/kt.log"
echo "kernel-trace done"
pass() { local name=$1; shift
  timeout -k 5 300 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- "${BENCH[@]}" > "$out/${name}_bench.json" 2> "$out/$name.log" || echo "pass $name failed"
  echo "pass $name done"; }
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
