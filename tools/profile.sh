#!/bin/bash
# Profiles one command with rocprofv3: one kernel-trace pass and separate counter passes (counters never
# share a run with tracing).  Run on the GPU box:
#   bash tools/profile.sh <tag> [program args...]     -> gpurun_out/prof_<tag>/
# The default program is the headline bench with its default step counts: python3 bench.py --no-cpu-baseline
set -u
TAG=${1:-r1}
shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- python3 bench.py --no-cpu-baseline; fi
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- "$@" > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
pass() { # name counters...
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- "${CMD[@]}" > $OUT/$name.log 2>&1 || { echo "pmc pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
}
CMD=("$@")
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_LDS
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
find $OUT -name "*.csv" | head -30
