#!/bin/bash
# rocprofv3 PMC passes over tools/tune.py (one launch per ray class after a warm-up launch).
# usage: tools/prof_passes.sh <outdir> [tune.py args...]   (run on the GPU box from the repo root)
set -u
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
pass() { # name counters...
  local name=$1; shift
  timeout -k 5 150 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 tools/tune.py --rounds 1 "${ARGS[@]}" > "$out/$name.log" 2>&1 || echo "pass $name failed"
}
ARGS=("$@")
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 tools/tune.py --rounds 3 "${ARGS[@]}" > "$out/kt.log" 2>&1
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcp2 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
ls "$out"
