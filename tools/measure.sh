#!/bin/bash
# One measurement series on the GPU box: the bench line of every workload, then rocprofv3 passes (kernel trace +
# separate counter passes, tools/profile.sh) for c2, c3 and c4.   bash tools/measure.sh <series>   -> gpurun_out/<series>/
set -u
S=${1:-r2_a}
O=gpurun_out/$S
mkdir -p $O
for wl in c2 c3 c4 c5; do
  timeout -k 10 400 python3 bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err || { echo "bench $wl failed"; tail -5 $O/bench_$wl.err; exit 1; }
  echo "bench $wl done"
done
for wl in c2 c3 c4; do
  bash tools/profile.sh ${S}_$wl python3 bench.py --workload $wl --no-cpu-baseline || exit 1
  echo "profile $wl done"
done
