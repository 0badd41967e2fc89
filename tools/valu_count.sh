#!/bin/bash
# VALU / SALU / LDS instruction counts per wave of the render kernel for one kbench command line
# usage (on the GPU box): bash tools/valu_count.sh <tag> <kbench args...>
TAG=$1; shift
export TMPDIR=/tmp
OUT=/tmp/valu_$TAG
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT -o c -- python3 tools/kbench.py --steps 5 "$@" > /dev/null 2>&1 || { echo "pmc failed"; exit 1; }
python3 - "$OUT" "$TAG" <<'PY'
import csv,glob,sys,collections
# per kernel name: dispatches, and per dispatch the counters
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob(sys.argv[1]+"/*counter_collection.csv")[0])):
    k=r["Kernel_Name"]
    if "renderKernel" in k or "renderSoloKernel" in k or "renderWaveKernel" in k:
        agg[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    m={n:sum(v)/len(v) for n,v in c.items()}
    w=m["SQ_WAVES"]
    print(sys.argv[2], k, "dispatches", len(c["SQ_WAVES"]), "waves",int(w), "VALU/wave",round(m["SQ_INSTS_VALU"]/w,1),"SALU/wave",round(m["SQ_INSTS_SALU"]/w,1),"LDS/wave",round(m["SQ_INSTS_LDS"]/w,1),
          "VMEM rd/wave",round(m["SQ_INSTS_VMEM_RD"]/w,1),"wr/wave",round(m["SQ_INSTS_VMEM_WR"]/w,1),"lane util",round(m["SQ_THREAD_CYCLES_VALU"]/(64*m["SQ_ACTIVE_INST_VALU"]),3),
          "VALU insts per launch (M)", round(m["SQ_INSTS_VALU"]/1e6,1))
PY
