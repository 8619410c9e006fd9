#!/bin/bash
# Stamp profile and two SQ counter passes of the 12-head / 35-slot shape (tools/h12bench.py).
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/h12pmc
mkdir -p $OUT
export TMPDIR=/tmp
CET_LIB=$REPO/channelestimationtransformer_amd/csrc/libcet_prof.so timeout -k 10 100 python3 tools/h12bench.py 2>&1 | grep "cet prof" | tail -1
cd /tmp
PASS_A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE"
PASS_B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SALU"
for p in A B; do
  eval "CNT=\$PASS_$p"
  timeout -s KILL 100 rocprofv3 --pmc $CNT --output-format csv -d $OUT/$p -- python3 $REPO/tools/h12bench.py > $OUT/$p.log 2>&1 || { tail -5 $OUT/$p.log; exit 1; }
  python3 - "$OUT/$p" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "informer_kernel" in k:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"{c:28s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
done
