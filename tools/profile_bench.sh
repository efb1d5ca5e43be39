#!/bin/bash
# rocprofv3 evidence for bench.py or another driver script (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the command (per-kernel average duration)
#   2. separate --pmc passes (HBM-side traffic: TCC_EA0 read requests by size, FETCH_SIZE, WRITE_SIZE;
#      SQ issue / wait / lanes; TCP / TCC) — never combined with traces
# usage: tools/profile_bench.sh <outdir> [--script tools/x.py] [--light] [--passes "sq1 tcc"] [args...]
#   default command: python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-wavefront --no-order-probe args...
#   --script S: python3 S args...        --light: kernel trace + the three passes the derived figures need
set -u
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
light=0
only=""
CMD=(python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-wavefront --no-order-probe)
while [ $# -gt 0 ]; do
  case "$1" in
    --script) CMD=(python3 "$2"); shift 2;;
    --light) light=1; shift;;
    --passes) only=" $2 "; shift 2;;
    *) break;;
  esac
done
CMD+=("$@")
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- "${CMD[@]}" > "$out/kt_bench.json" 2> "$out/kt.log"
echo "kernel-trace done"
pass() { local name=$1; shift
  if [ -n "$only" ] && [[ "$only" != *" $name "* ]]; then return; fi
  timeout -k 5 300 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- "${CMD[@]}" > "$out/${name}_bench.json" 2> "$out/$name.log" || echo "pass $name failed"
  echo "pass $name done"; }
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
if [ $light -eq 0 ]; then
  pass fetch FETCH_SIZE
  pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
fi
